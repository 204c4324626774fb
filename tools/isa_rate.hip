// ISA issue-rate microbenchmark for gfx950: measures wave-instruction throughput of the integer
// and f64 ops that 256-bit modular arithmetic can be built from. Not part of the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;
constexpr int UNROLL = 16;  // independent chains

template <int OP>
__global__ void k(uint32_t* out, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  uint64_t acc[UNROLL];
  double d[UNROLL];
  uint32_t w[UNROLL];
#pragma unroll
  for (int i = 0; i < UNROLL; i++) { acc[i] = a + i; w[i] = b + i; d[i] = (double)(a + i); }
  double da = (double)a, db = (double)b;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < UNROLL; i++) {
      if (OP == 0) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc"); }
      if (OP == 1) { asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(w[i]) : "v"(a)); }
      if (OP == 2) { asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(w[i]) : "v"(a)); }
      if (OP == 3) { asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(w[i]) : "v"(a), "v"(b)); }
      if (OP == 4) { asm volatile("v_mul_hi_u32_u24 %0, %1, %0" : "+v"(w[i]) : "v"(a)); }
      if (OP == 5) { asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(da), "v"(db)); }
      if (OP == 6) { asm volatile("v_add_co_u32 %0, vcc, %1, %0" : "+v"(w[i]) : "v"(a) : "vcc"); }
      if (OP == 7) { asm volatile("v_add_co_u32 %0, vcc, %1, %0\n\tv_addc_co_u32 %0, vcc, %2, %0, vcc" : "+v"(w[i]) : "v"(a), "v"(b) : "vcc"); }
      if (OP == 8) { asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(w[i]) : "v"(a), "v"(b)); }
      if (OP == 9) { asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(acc[i]) : "v"(acc[(i + 1) % UNROLL])); }
      if (OP == 10) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(w[i]) : "v"(a), "v"(b)); }
      if (OP == 11) { asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b) : "vcc"); }
      if (OP == 12) { asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(w[i]) : "v"(a)); }
      if (OP == 13) { asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(acc[i])); }
      if (OP == 14) { asm volatile("v_and_b32 %0, %1, %0" : "+v"(w[i]) : "v"(a)); }
      if (OP == 15) { asm volatile("v_mul_f64 %0, %1, %0" : "+v"(d[i]) : "v"(da)); }
      if (OP == 16) { asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[i]) : "v"(da)); }
      if (OP == 17) { asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(w[i]) : "v"(a), "v"(b)); }
      if (OP == 18) { asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(w[i]) : "v"(a), "v"(b)); }
      if (OP == 19) { asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(w[i])); }
      if (OP == 20) { asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(w[i]) : "v"(d[i])); }
      // one dependent chain (the pinned product column of fe_mac<true>), without and with the s_nop the hazard recogniser puts after
      // the pin's inline asm; counted as one multiply-add each
      if (OP == 21) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc"); }
      if (OP == 22) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\ts_nop 0" : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc"); }
    }
  }
  uint64_t s = 0; double ds = 0; uint32_t ws = 0;
#pragma unroll
  for (int i = 0; i < UNROLL; i++) { s += acc[i]; ds += d[i]; ws += w[i]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32) ^ ws ^ (uint32_t)ds;
}

typedef void (*kern_t)(uint32_t*, uint32_t);

// ------------------------------------------------------------------------------------------------------------------------------------
// Round 4 probe (review item 8): the constant-operand half of the Montgomery product (m * p, 81 v_mad_u64_u32 per multiplication on 9 x 29-bit
// limbs) on the matrix pipe.  For the 64 field elements of a wavefront m * p is a [64 x K] x [K x N] i8 contraction against the Toeplitz matrix
// of p's digits: 7-bit unsigned digits (the i8 MFMA is signed) give K = 38 digits of m (padded to 64), N = 75 product columns (5 tiles of 16),
// M = 64 lanes (4 tiles of 16) = 20 v_mfma_i32_16x16x64_i8 per wavefront-multiplication, column sums < 38 * 127^2 < 2^20.
// What it costs around the MFMAs, counted from the data path (none of it exists today, all of it is VALU / LDS-crossbar work of the same wave):
//   split      fe_pack (17) + 38 digits by v_bfe_u32 / v_alignbit_b32 (46) + 4 digits per operand register (30)             =  93 VALU
//   A layout   the MFMA wants lane l to hold row l % 16, k-slice l / 16; a lane owns its element's whole row: 16 ds_bpermute_b32
//   C layout   20 tiles x 4 registers come back column-major over the lanes; an element's 75 columns go to its lane:            80 ds_bpermute_b32
//   recombine  75 columns at bit 7 c into 18 limbs of 29 bits: one v_lshl_add_u64 per column + 18 carry steps                   =  93 VALU
// Streams measured (one "multiplication" per iteration, ITERS_MIX iterations, 1 / 2 / 4 waves per SIMD):
//   mode 0  162 v_mad_u64_u32                         today's product (a * b and m * p)
//   mode 1   81 v_mad_u64_u32 + 20 MFMA               m * p on the matrix pipe, NOTHING else: the upper bound of the idea
//   mode 2   mode 1 + the 93 split operations
//   mode 3   mode 2 + the 93 recombination operations
//   mode 4   mode 3 + the 96 ds_bpermute_b32 of the two layout changes
typedef int v4i_t __attribute__((ext_vector_type(4)));
constexpr int ITERS_MIX = 512;

template <int MODE>
__global__ void __launch_bounds__(256) kmix(uint32_t* out, uint32_t seed) {
  uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9e3779b9u;
  uint64_t col[17];
  uint32_t w[16];
#pragma unroll
  for (int i = 0; i < 17; i++) col[i] = a + i;
#pragma unroll
  for (int i = 0; i < 16; i++) w[i] = b + i;
  v4i_t acc[5], A[4], B[5];
#pragma unroll
  for (int i = 0; i < 5; i++) { acc[i] = v4i_t{(int)a, (int)b, (int)i, 0}; B[i] = v4i_t{(int)(b + i), (int)a, 1, 2}; }
#pragma unroll
  for (int i = 0; i < 4; i++) A[i] = v4i_t{(int)(a + i), (int)b, 3, 4};
  for (int it = 0; it < ITERS_MIX; it++) {
    // a * b: 81 multiply-adds on 17 independent columns (as the product-scanning code issues them)
#pragma unroll
    for (int i = 0; i < 81; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(col[i % 17]) : "v"(a), "v"(b) : "vcc");
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 81; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(col[i % 17]) : "v"(a), "v"(w[i % 16]) : "vcc");
    } else {
      if (MODE >= 2) {   // split: 17 pack + 46 extract + 30 merge
#pragma unroll
        for (int i = 0; i < 17; i++) asm volatile("v_lshl_or_b32 %0, %1, 3, %0" : "+v"(w[i % 16]) : "v"(a));
#pragma unroll
        for (int i = 0; i < 38; i++) asm volatile("v_bfe_u32 %0, %1, 7, 7" : "=v"(w[i % 16]) : "v"(w[(i + 5) % 16]));
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_alignbit_b32 %0, %1, %2, 25" : "=v"(w[i % 16]) : "v"(w[(i + 3) % 16]), "v"(w[(i + 7) % 16]));
#pragma unroll
        for (int i = 0; i < 30; i++) asm volatile("v_lshl_or_b32 %0, %1, 8, %0" : "+v"(w[i % 16]) : "v"(w[(i + 9) % 16]));
      }
      if (MODE >= 4) {   // A layout: 16 crossbar reads
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(w[i % 16]) : "v"(a), "v"(w[(i + 1) % 16]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; i++) A[i] = v4i_t{(int)w[4 * i], (int)w[4 * i + 1], (int)w[4 * i + 2], (int)w[4 * i + 3]};
      }
      // m * p: 4 row tiles x 5 column tiles
#pragma unroll
      for (int t = 0; t < 4; t++)
#pragma unroll
        for (int u = 0; u < 5; u++) acc[u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[t], B[u], acc[u], 0, 0, 0);
      if (MODE >= 4) {   // C layout: 80 crossbar reads
#pragma unroll
        for (int i = 0; i < 80; i++) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(w[i % 16]) : "v"(a), "v"(acc[i % 5][i % 4]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if (MODE >= 3) {   // recombine: 75 columns into 64-bit limb accumulators + 18 carry steps
#pragma unroll
        for (int i = 0; i < 75; i++) asm volatile("v_lshl_add_u64 %0, %1, 7, %0" : "+v"(col[i % 17]) : "v"(col[(i + 1) % 17]));
#pragma unroll
        for (int i = 0; i < 18; i++) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(col[i % 17]));
      }
    }
  }
  uint64_t s = 0; uint32_t ws = 0;
#pragma unroll
  for (int i = 0; i < 17; i++) s += col[i];
#pragma unroll
  for (int i = 0; i < 16; i++) ws += w[i];
#pragma unroll
  for (int i = 0; i < 5; i++) ws += (uint32_t)(acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)s ^ (uint32_t)(s >> 32) ^ ws;
}

static int mfma_probe(uint32_t* out, int cus, hipEvent_t e0, hipEvent_t e1) {
  const char* names[] = {"162 mad (today)", "81 mad + 20 MFMA i8 16x16x64", "... + split (93 VALU)", "... + recombine (93 VALU)", "... + 96 ds_bpermute (layouts)"};
  kern_t ks[] = {kmix<0>, kmix<1>, kmix<2>, kmix<3>, kmix<4>};
  printf("\n# MFMA probe: Montgomery product with m * p on the matrix pipe (see the comment in tools/isa_rate.hip); ns per wave-multiplication per SIMD\n");
  double base[4] = {0, 0, 0, 0};
  for (int mode = 0; mode < 5; mode++) {
    int wi = 0;
    for (int wps : {1, 2, 4}) {
      const int blocks = cus * wps;
      hipLaunchKernelGGL(ks[mode], dim3(blocks), dim3(256), 0, 0, out, 1u);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(ks[mode], dim3(blocks), dim3(256), 0, 0, out, 2u);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double ns = ms * 1e6 / ((double)ITERS_MIX * wps);
      if (mode == 0) base[wi] = ns;
      printf("%-34s waves/SIMD %d  %.3f ms  %.1f ns per wave-multiplication per SIMD (%.0f cyc @2.4GHz)  %.2f x today\n", names[mode], wps, ms, ns, ns * 2.4, ns / base[wi]);
      wi++;
    }
  }
  return 0;
}

int main() {
  const char* names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mul_hi_u32_u24",
                         "v_fma_f64", "v_add_co_u32", "v_add_co+v_addc(2)", "v_add3_u32", "v_lshl_add_u64",
                         "v_fma_f32", "v_mad_i64_i32", "v_mul_u32_u24", "v_lshrrev_b64", "v_and_b32",
                         "v_mul_f64", "v_add_f64", "v_dot4_u32_u8", "v_mad_u32_u16", "v_cvt_f64_u32", "v_cvt_u32_f64",
                         "v_mad_u64_u32 dependent chain", "v_mad_u64_u32 dependent + s_nop 0"};
  kern_t ks[] = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>, k<13>, k<14>,
                 k<15>, k<16>, k<17>, k<18>, k<19>, k<20>, k<21>, k<22>};
  const int nops = sizeof(ks) / sizeof(ks[0]);
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s CUs %d clock %d kHz\n", prop.name, cus, prop.clockRate);
  uint32_t* out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 1024 * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  // waves per SIMD: 1, 2, 4 (block 256 = one wave per SIMD; blocks per CU 1, 2, 4)
  for (int op = 0; op < nops; op++) {
    for (int wps : {1, 2, 4, 8}) {
      int blocks = cus * wps;
      hipLaunchKernelGGL(ks[op], dim3(blocks), dim3(256), 0, 0, out, 1u);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(ks[op], dim3(blocks), dim3(256), 0, 0, out, 2u);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      double winstr = (double)ITERS * UNROLL * wps;  // wave-instructions per SIMD
      if (op == 7) winstr *= 2;
      double ns_per = ms * 1e6 / winstr;
      printf("%-20s waves/SIMD %d  %.3f ms  %.2f ns/wave-instr/SIMD (= %.2f cyc @2.4GHz)  chip %.2f Tlane-op/s\n",
             names[op], wps, ms, ns_per, ns_per * 2.4, 64.0 * cus * 4 / ns_per / 1e3);
    }
  }
  return mfma_probe(out, cus, e0, e1);
}
