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
  return 0;
}
