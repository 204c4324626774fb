// Round 5 probe (review item 1): a COMPLETE BN254 base-field Montgomery multiplication on 5 x 52-bit limbs held as doubles, built from
// v_fma_f64 high / low product pairs, checked bit-exact against a host big-integer reference and timed against the product path's
// 9 x 29-bit v_mad_u64_u32 multiplication (csrc/fp29.hpp) in the same harness.  Not part of the product path.
//
// What it would replace: halo2curves bn256::Fq::mul [DEP] under best_multiexp (reached from /root/reference/aggregator/src/wrapper.rs:129).
//
// Formulation (the published double-precision technique of Emmart, Zheng, Weems, "Faster modular exponentiation using double precision
// floating point arithmetic on the GPU", ARITH 2018, restated for gfx950).  The f64 rounding mode is set to round-toward-zero once per
// kernel (s_setreg MODE[3:2] = 3); then for integers 0 <= x, y < 2^52 held exactly in doubles
//     hi = fma(x, y, 2^104)                  = 2^104 + 2^52 h,  h = floor(x y / 2^52)          (the ulp of [2^104, 2^105) is 2^52)
//     lo = fma(x, y, (2^104 + 2^52) - hi)    = 2^52 + l,        l = x y mod 2^52               (exact)
//   so x y = 2^52 h + l, and the BIT PATTERNS of hi and lo are affine in h and l: raw(hi) = (0x467 << 52) + h, raw(lo) = (0x433 << 52) + l.
//   (Round-to-nearest would make l signed, and its bias 1.5 * 2^52 is half an ulp of 2^104: a fourth floating-point operation per product.)
//   Column sums are 64-bit INTEGER additions of the raw patterns (a sum of ten 52-bit values needs 56 bits: not a double), with the
//   accumulated biases folded into the columns' start values.
//   Per 52 x 52 product: 2 v_fma_f64 + 1 v_add_f64 + 2 64-bit integer additions = 5 instructions for 2704 product bits
//   (9 x 29-bit limbs: 1 v_mad_u64_u32 for 841 product bits, the column sum included).
//   Montgomery, radix 2^52, R = 2^260, operand scanning over the quotient digits: q_k = (col_k mod 2^52) * (-p^-1) mod 2^52 by the same
//   hi / lo pair (6 instructions), 25 + 25 products, 9 carries (shift + add), 5 result limbs back to doubles (2 each).
//   Result (a b + q p) / R < a b / R + p.
//
// Build: hipcc -O3 --offload-arch=gfx950 -I zksnap_circuits_halo2_amd/csrc tools/fp64_field.hip -o tools/fp64_field
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "fp29.hpp"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

namespace f52 {

constexpr int N = 5;
constexpr uint64_t M52 = (1ull << 52) - 1;
// BN254 Fq in 52-bit limbs; INV = -p^-1 mod 2^52 (tools/gen_constants.py's integers re-cut)
__device__ __constant__ const double P[N] = {(double)0x8c16d87cfd47ull, (double)0x916871ca8d3c2ull, (double)0x181585d97816aull, (double)0xa029b85045b68ull, (double)0x30644e72e131ull};
constexpr uint64_t P_INT[N] = {0x8c16d87cfd47ull, 0x916871ca8d3c2ull, 0x181585d97816aull, 0xa029b85045b68ull, 0x30644e72e131ull};
constexpr uint64_t INV_INT = 0x20782e4866389ull;

constexpr uint64_t LO_B = 0x433ull << 52;          // raw(2^52)
constexpr uint64_t HI_B = 0x467ull << 52;          // raw(2^104)

struct fe { double l[N]; };

// f64 rounding mode = toward zero: MODE[3:2] <- 3.  As inline asm: after __builtin_amdgcn_s_setreg the compiler's mode-register pass
// puts the mode back to round-to-nearest in front of the first f64 operation.
__device__ __forceinline__ void round_toward_zero() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3" ::: "memory"); }

// number of index pairs (i, j), 0 <= i, j < 5, with i + j = k
constexpr int pairs(int k) { return k < 0 ? 0 : (k < N ? k + 1 : (k <= 2 * N - 2 ? 2 * N - 1 - k : 0)); }
// start value of column k: minus every bias that will be added into it (a b and q p: twice the same counts)
constexpr uint64_t col_start(int k) { return 0 - 2 * ((uint64_t)pairs(k) * LO_B + (uint64_t)pairs(k - 1) * HI_B); }

__device__ __forceinline__ uint64_t raw(double x) { return (uint64_t)__double_as_longlong(x); }

// x, y integers in [0, 2^52): clo += x y mod 2^52, chi += x y >> 52 (plus the biases)
__device__ __forceinline__ void prod(double x, double y, uint64_t& clo, uint64_t& chi) {
  const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
  const double hi = __builtin_fma(x, y, C1);
  const double lo = __builtin_fma(x, y, C2 - hi);
  clo += raw(lo);
  chi += raw(hi);
}

// Montgomery product a b 2^-260 mod p, limbs in [0, 2^52); inputs a, b < 2^256 give a result < 1.4 p.  Needs round_toward_zero().
__device__ __forceinline__ fe mul(const fe& a, const fe& b) {
  uint64_t col[2 * N + 1];
#pragma unroll
  for (int k = 0; k <= 2 * N; k++) col[k] = col_start(k);
#pragma unroll
  for (int i = 0; i < N; i++)
#pragma unroll
    for (int j = 0; j < N; j++) prod(a.l[i], b.l[j], col[i + j], col[i + j + 1]);
#pragma unroll
  for (int k = 0; k < N; k++) {
    // t = col[k] mod 2^52 as a double: splice the low 52 bits under the exponent of 2^52 (every bias is a multiple of 2^52)
    const double t = __longlong_as_double((long long)((col[k] & M52) | LO_B)) - 0x1p52;
    const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
    const double hi = __builtin_fma(t, (double)INV_INT, C1);
    const double q = __builtin_fma(t, (double)INV_INT, C2 - hi) - 0x1p52;       // t * INV mod 2^52
#pragma unroll
    for (int j = 0; j < N; j++) prod(q, P[j], col[k + j], col[k + j + 1]);
    col[k + 1] += col[k] >> 52;                                                  // col[k] is now a multiple of 2^52
  }
  fe r;
#pragma unroll
  for (int k = N; k < 2 * N; k++) {
    r.l[k - N] = __longlong_as_double((long long)((col[k] & M52) | LO_B)) - 0x1p52;
    col[k + 1] += col[k] >> 52;
  }
  return r;
}

// 4 x u64 little-endian words <-> 5 doubles (what a load / store in the external format would add)
__device__ __forceinline__ fe from_words(const uint64_t (&w)[4]) {
  fe r;
#pragma unroll
  for (int k = 0; k < N; k++) {
    const int bit = 52 * k, wi = bit >> 6, sh = bit & 63;
    uint64_t v = w[wi] >> sh;
    if (sh > 12 && wi + 1 < 4) v |= w[wi + 1] << (64 - sh);
    r.l[k] = __longlong_as_double((long long)((v & M52) | LO_B)) - 0x1p52;
  }
  return r;
}
__device__ __forceinline__ void to_words(const fe& a, uint64_t (&w)[4]) {
  uint64_t v[N];
#pragma unroll
  for (int k = 0; k < N; k++) v[k] = raw(a.l[k] + 0x1p52) & M52;
  w[0] = v[0] | (v[1] << 52);
  w[1] = (v[1] >> 12) | (v[2] << 40);
  w[2] = (v[2] >> 24) | (v[3] << 28);
  w[3] = (v[3] >> 36) | (v[4] << 16);
}

}  // namespace f52

// ---- correctness: out[i] = a[i] * b[i] * 2^-260 (external words in, external words out) ---------------------------------------------
__global__ void k_check(const uint64_t* a, const uint64_t* b, uint64_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f52::round_toward_zero();
  uint64_t wa[4], wb[4], wr[4];
#pragma unroll
  for (int j = 0; j < 4; j++) { wa[j] = a[4 * i + j]; wb[j] = b[4 * i + j]; }
  f52::to_words(f52::mul(f52::from_words(wa), f52::from_words(wb)), wr);
#pragma unroll
  for (int j = 0; j < 4; j++) out[4 * i + j] = wr[j];
}

// ---- timing: CH independent dependent chains x <- x * y per lane ---------------------------------------------------------------------
constexpr int ITERS = 256;

template <int CH>
__global__ void __launch_bounds__(256) k_time_f52(uint64_t* out, uint32_t seed) {
  f52::fe x[CH], y;
  f52::round_toward_zero();
#pragma unroll
  for (int c = 0; c < CH; c++)
#pragma unroll
    for (int k = 0; k < f52::N; k++) x[c].l[k] = (double)((threadIdx.x * 2654435761u + seed + 977u * k + 31u * c) & 0xfffffu) * 4294967296.0 + (double)(seed * 7u + k);
#pragma unroll
  for (int k = 0; k < f52::N; k++) y.l[k] = (double)((threadIdx.x * 40503u + seed + 13u * k) & 0xfffffu) * 4294967296.0 + (double)(seed * 3u + k);
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int c = 0; c < CH; c++) x[c] = f52::mul(x[c], y);
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; c++)
#pragma unroll
    for (int k = 0; k < f52::N; k++) s += x[c].l[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)__double_as_longlong(s);
}

template <int CH, bool CHAIN>
__global__ void __launch_bounds__(256) k_time_f29(uint64_t* out, uint32_t seed) {
  using namespace zkhip;
  fe x[CH], y;
#pragma unroll
  for (int c = 0; c < CH; c++)
#pragma unroll
    for (int k = 0; k < NL; k++) x[c].l[k] = (threadIdx.x * 2654435761u + seed + 977u * k + 31u * c) & LMASK;
#pragma unroll
  for (int k = 0; k < NL; k++) y.l[k] = (threadIdx.x * 40503u + seed + 13u * k) & LMASK;
  x[0].l[NL - 1] &= 0xfffff; y.l[NL - 1] &= 0xfffff;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int c = 0; c < CH; c++) x[c] = fe_mul<FqParams, CHAIN>(x[c], y);
  }
  uint32_t s = 0;
#pragma unroll
  for (int c = 0; c < CH; c++)
#pragma unroll
    for (int k = 0; k < NL; k++) s += x[c].l[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- host big-integer reference ------------------------------------------------------------------------------------------------------
typedef unsigned __int128 u128;
struct u256 { uint64_t w[4]; };
static const u256 Q = {{0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull}};
static const u256 RINV260 = {{0xf736cbda0b09dd16ull, 0x5cbb5610a2123cc3ull, 0xb9c17ff814dd8c05ull, 0x30447ae2c8bce28aull}};   // 2^-260 mod q
static bool geq(const u256& a, const u256& b) { for (int i = 3; i >= 0; i--) if (a.w[i] != b.w[i]) return a.w[i] > b.w[i]; return true; }
static u256 sub(const u256& a, const u256& b) { u256 r; u128 br = 0; for (int i = 0; i < 4; i++) { u128 t = (u128)a.w[i] - b.w[i] - br; r.w[i] = (uint64_t)t; br = (t >> 64) & 1; } return r; }
static u256 addmod(const u256& a, const u256& b) {   // a, b < q
  u256 r; u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)a.w[i] + b.w[i]; r.w[i] = (uint64_t)c; c >>= 64; }
  if (c || geq(r, Q)) r = sub(r, Q);
  return r;
}
static u256 reduce(u256 a) { while (geq(a, Q)) a = sub(a, Q); return a; }   // a < 2^256 < 6 q
static u256 mulmod(const u256& a, const u256& b) {     // double-and-add, a, b < q
  u256 r = {{0, 0, 0, 0}};
  for (int i = 255; i >= 0; i--) { r = addmod(r, r); if ((b.w[i >> 6] >> (i & 63)) & 1) r = addmod(r, a); }
  return r;
}
static uint64_t splitmix(uint64_t& s) { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }

template <class K>
static int time_kernel(const char* name, K kern, int ch, uint64_t* out, int cus, hipEvent_t e0, hipEvent_t e1) {
  for (int wps : {1, 2, 4}) {
    const int blocks = cus * wps;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 2u);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ns = ms * 1e6 / ((double)ITERS * ch * wps);
    printf("%-46s chains/lane %d  waves/SIMD %d  %.3f ms  %.1f ns per wave-multiplication per SIMD (%.0f cyc @2.4GHz)\n", name, ch, wps, ms, ns, ns * 2.4);
  }
  return 0;
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s CUs %d\n", prop.name, cus);
  // correctness: 65536 random pairs below 2q + the edge list (0, 1, q - 1, q, q + 1, 2q - 1, 2^256 - 1, limb boundaries)
  const int NR = 65536;
  std::vector<u256> A, B;
  uint64_t s = 0x5A4B534E41500005ull;
  const u256 edges[] = {{{0, 0, 0, 0}}, {{1, 0, 0, 0}}, sub(Q, u256{{1, 0, 0, 0}}), Q, {{Q.w[0] + 1, Q.w[1], Q.w[2], Q.w[3]}},
                        {{~0ull, ~0ull, ~0ull, ~0ull}}, {{0, ~0ull, 0, ~0ull}}, {{~0ull, 0, ~0ull, 0}},
                        {{(1ull << 52) - 1, 0, 0, 0}}, {{1ull << 52, 0, 0, 0}}, {{0xfffffffffffff000ull, 0xfffull, 0, 0}}, {{0, 0, 0, 1ull << 63}}};
  const int NE = sizeof(edges) / sizeof(edges[0]);
  for (int i = 0; i < NE; i++) for (int j = 0; j < NE; j++) { A.push_back(edges[i]); B.push_back(edges[j]); }
  for (int i = 0; i < NR; i++) {
    u256 a, b;
    for (int j = 0; j < 4; j++) { a.w[j] = splitmix(s); b.w[j] = splitmix(s); }
    if (i & 1) { a.w[3] &= 0x3fffffffffffffffull; b.w[3] &= 0x3fffffffffffffffull; }   // half of them below 2^254, half anywhere below 2^256
    A.push_back(a); B.push_back(b);
  }
  const int n = (int)A.size();
  uint64_t *da, *db, *dout;
  CHECK(hipMalloc(&da, n * 32)); CHECK(hipMalloc(&db, n * 32)); CHECK(hipMalloc(&dout, (size_t)cus * 8 * 256 * 8 > (size_t)n * 32 ? (size_t)cus * 8 * 256 * 8 : (size_t)n * 32));
  CHECK(hipMemcpy(da, A.data(), n * 32, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, B.data(), n * 32, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_check, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, dout, n);
  CHECK(hipDeviceSynchronize());
  std::vector<u256> Rr(n);
  CHECK(hipMemcpy(Rr.data(), dout, n * 32, hipMemcpyDeviceToHost));
  int bad = 0, over = 0;
  for (int i = 0; i < n; i++) {
    const u256 exp = mulmod(mulmod(reduce(A[i]), reduce(B[i])), RINV260);
    u256 two_q; { u128 c = 0; for (int j = 0; j < 4; j++) { c += (u128)Q.w[j] * 2; two_q.w[j] = (uint64_t)c; c >>= 64; } }
    if (geq(Rr[i], two_q)) over++;      // a b / 2^260 + p < 2^252 + p < 1.4 p: never
    const u256 got = reduce(Rr[i]);
    if (memcmp(&got, &exp, 32) != 0) { if (bad < 5) printf("MISMATCH at %d\n", i); bad++; }
  }
  printf("fp64 Montgomery product (5 x 52-bit limbs): %d cases (%d edge pairs + %d random), mismatches %d, results >= 2q: %d\n", n, NE * NE, NR, bad, over);

  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("\n# dependent chains x <- x * y, %d multiplications per chain; ns per wave-multiplication per SIMD\n", ITERS);
  if (time_kernel("9 x 29-bit v_mad_u64_u32, compiler's columns", k_time_f29<1, false>, 1, dout, cus, e0, e1)) return 1;
  if (time_kernel("9 x 29-bit v_mad_u64_u32, compiler's columns", k_time_f29<2, false>, 2, dout, cus, e0, e1)) return 1;
  if (time_kernel("9 x 29-bit v_mad_u64_u32, single-chain columns", k_time_f29<1, true>, 1, dout, cus, e0, e1)) return 1;
  if (time_kernel("9 x 29-bit v_mad_u64_u32, single-chain columns", k_time_f29<2, true>, 2, dout, cus, e0, e1)) return 1;
  if (time_kernel("5 x 52-bit v_fma_f64 hi/lo pairs", k_time_f52<1>, 1, dout, cus, e0, e1)) return 1;
  if (time_kernel("5 x 52-bit v_fma_f64 hi/lo pairs", k_time_f52<2>, 2, dout, cus, e0, e1)) return 1;
  return bad != 0;
}
