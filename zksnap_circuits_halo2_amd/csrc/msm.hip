// BN254 G1 multi-scalar multiplication for MI355X (gfx950).
//
// Replaces halo2-axiom `best_multiexp` / `multiexp_serial` [DEP] (halo2_proofs/src/arithmetic.rs; reached
// from /root/reference/aggregator/src/wrapper.rs:129 `create_proof`, :108 `keygen_pk` and
// /root/reference/aggregator/benches/wrapper_circuit.rs:140 `gen_proof`).  Same mathematical result
// (sum_i s_i * P_i), different algorithm: the reference chunks points over CPU threads and runs an
// unsigned c = ceil(ln n) Pippenger per chunk; here one GPU runs a signed-digit Pippenger over all points:
//
//   1. digits      scalar -> canonical integer -> W signed c-bit digits (int16), window-major
//   2. count       per (window, chunk) workgroup: histogram of |digit|-1 in LDS, merged into the global
//                  per-bucket counts with *contiguous* atomics (one wave-instruction = 256 contiguous bytes)
//   3. scan        exclusive scan of the counts -> bucket offsets
//   4. scatter     same tiling: LDS histogram again, one ranged global atomicAdd per non-empty bucket reserves
//                  the slot range, LDS cursors place (point index | sign<<31) -> counting sort by bucket
//   5. tasks       buckets are cut into tasks of <= 2^task_shift entries (load balance independent of the
//                  scalar distribution: a bucket holding 100k points becomes 100k / 2^task_shift tasks)
//   6. accumulate  one thread per task, XYZZ accumulator in registers, mixed additions
//   7. combine     per bucket: in-place radix-4 tree over its task partials (any skew)
//   8. reduce      log-depth pyramid: sum_k (k+1) B_k = Tot + sum_l 2^l T_l,  T_l = sum of buckets with bit l
//   9. fold        Horner over the T_l and over the windows
//
// HBM layout: digits int16 [W][n]; sorted refs u32 [W*n]; counts/offsets u32 [W*B+1]; XYZZ points are
// 36 x u32 (9 limbs x 4 coordinates, 144 B) arrays-of-structs.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include "ec_quad.hpp"
#include "fe_inverse.hpp"
#include "zkhip_internal.hpp"

namespace zkhip {

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

constexpr int FINE_BITS = 12;           // wide windows: a bucket id = (coarse group, 12-bit fine id); MAX_GROUPS = 2^(20 - 1 - 12) groups at c = 20
constexpr int MAX_GROUPS = 128;
constexpr int MAX_WIDE_W = 16;          // windows of the wide path (c >= 17: at most 16): the coarse groups are counted and reserved per (window, group)
// Balanced windows of the wide prepared path.  W = ceil(256 / c) windows of c bits cover W c >= 256 bits, and with uniform widths the surplus sits
// in the top window: at c = 20 it holds 15 bits, so its 2^20 digits fall into the lowest 2^14 buckets, which then hold ~90 entries = two tasks
// (a combining addition each, and the longest chains of the accumulation) while every other bucket holds ~26.  Here the last `narrow` = W c - 256
// windows are c - 1 bits wide instead (c = 20: nine windows of 20 bits, four of 19): every window fills at least half of the bucket range.  The top
// window still ends at bit 256, two bits above any scalar, so the signed recoding never carries out of it.  Offsets: win_off(w).
struct win_layout { int c, W, narrow; };
__host__ __device__ __forceinline__ int win_bits(const win_layout& L, int w) { return w < L.W - L.narrow ? L.c : L.c - 1; }
__host__ __device__ __forceinline__ int win_off(const win_layout& L, int w) { const int full = L.W - L.narrow; return w <= full ? w * L.c : full * L.c + (w - full) * (L.c - 1); }
static int msm_narrow_windows(int c, int W) {
  static const bool uniform = getenv("ZKHIP_UNIFORM_WINDOWS") != nullptr;      // A/B knob (read once: tables and digits of a process agree)
  if (c <= 16 || uniform) return 0;
  const int narrow = W * c - 256;
  return narrow > 0 && narrow <= W ? narrow : 0;
}
constexpr int MAX_TASK_LEN = 128;       // tasks are 2^task_shift entries, task_shift <= 7 (length histograms hold MAX_TASK_LEN + 1 counters)
constexpr int TASK_SHIFT = 6;           // 64 entries per accumulate task: short tasks keep the tail of the launch balanced
                                        // (measured at 2^22: 5.9 ms with 64-entry tasks, 6.9 ms with 256, 8.1 ms with 512)

struct task_t {
  uint32_t bucket, start, len;
};

// ------------------------------------------------------------------------------------------------
// XYZZ <-> global memory (array of 36-word structs)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ xyzz load_xyzz(const uint32_t* base, size_t idx) {
  const uint4* q = reinterpret_cast<const uint4*>(base + idx * 36);
  uint32_t w[36];
#pragma unroll
  for (int i = 0; i < 9; i++) {
    uint4 v = q[i];
    w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
  }
  xyzz r;
#pragma unroll
  for (int i = 0; i < 9; i++) { r.X.l[i] = w[i]; r.Y.l[i] = w[9 + i]; r.ZZ.l[i] = w[18 + i]; r.ZZZ.l[i] = w[27 + i]; }
  return r;
}

__device__ __forceinline__ void store_xyzz(uint32_t* base, size_t idx, const xyzz& a) {
  uint32_t w[36];
#pragma unroll
  for (int i = 0; i < 9; i++) { w[i] = a.X.l[i]; w[9 + i] = a.Y.l[i]; w[18 + i] = a.ZZ.l[i]; w[27 + i] = a.ZZZ.l[i]; }
  uint4* q = reinterpret_cast<uint4*>(base + idx * 36);
#pragma unroll
  for (int i = 0; i < 9; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

__device__ __forceinline__ xyzz xyzz_shfl_xor(const xyzz& a, int mask) {
  xyzz r;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    r.X.l[i] = (uint32_t)__shfl_xor((int)a.X.l[i], mask, 64);
    r.Y.l[i] = (uint32_t)__shfl_xor((int)a.Y.l[i], mask, 64);
    r.ZZ.l[i] = (uint32_t)__shfl_xor((int)a.ZZ.l[i], mask, 64);
    r.ZZZ.l[i] = (uint32_t)__shfl_xor((int)a.ZZZ.l[i], mask, 64);
  }
  return r;
}

__device__ __forceinline__ fe load_fe9_generic(const uint32_t* p, size_t idx) {
  fe r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = p[idx * 9 + i];
  return r;
}
__device__ __forceinline__ void store_fe9_generic(uint32_t* p, size_t idx, const fe& a) {
#pragma unroll
  for (int i = 0; i < 9; i++) p[idx * 9 + i] = a.l[i];
}

// ------------------------------------------------------------------------------------------------
// 1. digits
// ------------------------------------------------------------------------------------------------
template <typename DIGIT>   // int16_t for c <= 16, int32_t for the wide windows of the prepared path
__global__ void __launch_bounds__(256) k_digits(const uint32_t* __restrict__ scalars, DIGIT* __restrict__ digits,
                                                uint32_t n, uint32_t n_pad, int c, int W, uint32_t K, size_t scalar_stride, int narrow) {
  // batch: K scalar vectors (vector k at scalars + k * scalar_stride elements); digits are laid out [W][K][n_pad]
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)K * n_pad) return;
  const uint32_t kb = (uint32_t)(idx / n_pad), i = (uint32_t)(idx % n_pad);
  digits += (size_t)kb * n_pad;
  const size_t row = (size_t)K * n_pad;
  if (i >= n) {                                   // padding entries: digit 0 = "no entry"
    for (int win = 0; win < W; win++) digits[(size_t)win * row + i] = 0;
    return;
  }
  uint32_t w[8];
  load_words(scalars + ((size_t)kb * scalar_stride + i) * 8, w);
  // Montgomery-256 -> plain integer: mont261(a*2^256, 2^5) = a
  fe c32;
#pragma unroll
  for (int k = 0; k < NL; k++) c32.l[k] = FrParams::FROM_EXT_CANON[k];
  fe s = fe_canon_lt2p<FrParams>(fe_mul<FrParams>(c32, fe_unpack<0>(w)));
  fe_pack(s, w);
  const win_layout lay{c, W, narrow};
  uint32_t carry = 0;
  for (int win = 0; win < W; win++) {
    const int bit = win_off(lay, win), cw = win_bits(lay, win);
    const uint32_t half = 1u << (cw - 1), mask = (1u << cw) - 1;
    uint32_t v = 0;
    if (bit < 256) {
      int wi = bit >> 5, sh = bit & 31;
      uint64_t two = w[wi];
      if (wi + 1 < 8) two |= (uint64_t)w[wi + 1] << 32;
      v = (uint32_t)(two >> sh) & mask;
    }
    v += carry;
    int32_t d;
    if (v >= half) { d = (int32_t)v - (int32_t)(1u << cw); carry = 1; } else { d = (int32_t)v; carry = 0; }
    digits[(size_t)win * row + i] = (DIGIT)d;
  }
}

// Wide windows (one vector, int32 digits): the same digits, and the coarse groups' entry counts on the way -- the two-level sort needs the
// group sizes before it can place anything, and counting them used to be a pass of its own over the 52 bytes of digits per scalar
// (k_coarse_count).  A workgroup of 1024 threads walks its scalars with a grid stride, counts in LDS and merges its 2^(c-13) counters
// with one global atomic each at the end: 512 - 1024 workgroups, so at most 128 K atomics on the 128 words (the naive form -- one
// 256-thread workgroup per 256 scalars, 0.5 M atomics -- was measured in round 1: +39 us in this kernel for the -33 us it saved).
__global__ void __launch_bounds__(1024) k_digits_wide(const uint32_t* __restrict__ scalars, int32_t* __restrict__ digits, uint32_t n, uint32_t n_pad,
                                                       int c, int W, uint32_t* __restrict__ gcount, int narrow) {
  __shared__ uint32_t cnt[MAX_WIDE_W * MAX_GROUPS];          // [window][group]
  for (uint32_t t = threadIdx.x; t < (uint32_t)W * MAX_GROUPS; t += blockDim.x) cnt[t] = 0;
  __syncthreads();
  const win_layout lay{c, W, narrow};
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += gridDim.x * blockDim.x) {
    if (i >= n) {                                   // padding entries: digit 0 = "no entry"
      for (int win = 0; win < W; win++) digits[(size_t)win * n_pad + i] = 0;
      continue;
    }
    uint32_t w[8];
    load_words(scalars + (size_t)i * 8, w);
    fe c32;
#pragma unroll
    for (int k = 0; k < NL; k++) c32.l[k] = FrParams::FROM_EXT_CANON[k];
    fe sv = fe_canon_lt2p<FrParams>(fe_mul<FrParams>(c32, fe_unpack<0>(w)));
    fe_pack(sv, w);
    uint32_t carry = 0;
    for (int win = 0; win < W; win++) {
      const int bit = win_off(lay, win), cw = win_bits(lay, win);
      const uint32_t half = 1u << (cw - 1), mask = (1u << cw) - 1;
      uint32_t v = 0;
      if (bit < 256) {
        const int wi = bit >> 5, sh = bit & 31;
        uint64_t two = w[wi];
        if (wi + 1 < 8) two |= (uint64_t)w[wi + 1] << 32;
        v = (uint32_t)(two >> sh) & mask;
      }
      v += carry;
      int32_t d;
      if (v >= half) { d = (int32_t)v - (int32_t)(1u << cw); carry = 1; } else { d = (int32_t)v; carry = 0; }
      digits[(size_t)win * n_pad + i] = d;
      if (d != 0) atomicAdd(&cnt[win * MAX_GROUPS + (((uint32_t)(d < 0 ? -d : d) - 1) >> FINE_BITS)], 1u);
    }
  }
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < (uint32_t)W * MAX_GROUPS; t += blockDim.x)
    if (cnt[t]) atomicAdd(&gcount[t], cnt[t]);
}

// ------------------------------------------------------------------------------------------------
// 1'. GLV digits (general path: arbitrary, unprepared bases).  BN254 has the endomorphism phi(x, y) = (beta x, y) = lambda (x, y) with
//     lambda^3 = 1 in Fr, beta^3 = 1 in Fq, so k P = k1 P + k2 phi(P) with |k1|, |k2| < 2^127: the MSM over n points and 254-bit scalars
//     becomes one over 2n points (P_i, phi(P_i)) and 127-bit scalars -- the same number of digit entries (W' = ceil(128 / c) windows x 2n),
//     half the bucket sets, and a window fold of 128 - c dependent doublings instead of 256 - c (the fold is one quad's latency chain:
//     0.60 of 2.3 ms at 2^20, 0.70 of 1.25 ms at 2^17).  Decomposition (the lattice basis of the published GLV parameters of the curve,
//     re-derived here with the extended Euclid on (r, lambda) and checked in tests/test_oracle.py):
//         v1 = (a1, b1) = (0x89d3256894d213e3, -0x6f4d8248eeb859fc8211bbeb7d4f1128),  v2 = (a2, b2) = (0x6f4d8248eeb859fd0be4e1541221250b, 0x89d3256894d213e3)
//         c1 = floor(k g1 / 2^256), g1 = round(2^256 b2 / r);   c2 = floor(k g2 / 2^256), g2 = round(2^256 (-b1) / r)
//         k1 = k - c1 a1 - c2 a2,   k2 = c1 (-b1) - c2 b2          (k1 + lambda k2 = k mod r for ANY c1, c2: the rounding only bounds the size)
//     With floors of the approximate quotients (c = floor(x + e), |e| <= k / 2^257 < 1/8) |k_i| < 9/8 (|a1| + |a2|) < 0.979 * 2^127, so the
//     signed recoding never carries out of the top window (msm_windows adds a window for the two window sizes where it would).
//     Differences are taken mod 2^160 (sign = bit 159).
// ------------------------------------------------------------------------------------------------
namespace glv {
__device__ constexpr uint32_t G1[3] = {0xc7e0b3d7u, 0xd91d232eu, 0x2u};
__device__ constexpr uint32_t G2[5] = {0x391eb18eu, 0x7a7bd9d4u, 0xa773d2cfu, 0x4ccef014u, 0x2u};
__device__ constexpr uint32_t A1[2] = {0x94d213e3u, 0x89d32568u};                               // = b2
__device__ constexpr uint32_t A2[4] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u};
__device__ constexpr uint32_t NB1[4] = {0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u};    // -b1
// beta (the cube root of unity in Fq with phi = lambda for this lambda), Montgomery-256 words
__device__ constexpr uint32_t BETA_EXT[8] = {0xd782e155u, 0x71930c11u, 0xffbe3323u, 0xa6bb947cu, 0xd4741444u, 0xaa303344u, 0x26594943u, 0x2c3b3f0du};

// out[0 .. NO) = low NO words of a (NA words) * b (NB words)
template <int NA, int NB, int NO>
__device__ __forceinline__ void mul_low(const uint32_t (&a)[NA], const uint32_t* __restrict__ b, uint32_t (&out)[NO]) {
#pragma unroll
  for (int i = 0; i < NO; i++) out[i] = 0;
#pragma unroll
  for (int i = 0; i < NA; i++) {
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < NB; j++) {
      if (i + j < NO) {
        const uint64_t t = (uint64_t)a[i] * b[j] + out[i + j] + carry;
        out[i + j] = (uint32_t)t;
        carry = (uint32_t)(t >> 32);
      }
    }
    if (i + NB < NO) out[i + NB] = carry;      // (this word has not been written by an earlier row: rows ascend)
  }
}

// magnitude (4 words) and sign of a two's-complement 160-bit value whose magnitude is below 2^127
__device__ __forceinline__ bool abs160(uint32_t (&v)[5], uint32_t (&mag)[4]) {
  const bool neg = (v[4] >> 31) != 0;
  uint32_t carry = neg ? 1u : 0u;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t w = neg ? ~v[i] : v[i];
    const uint32_t t = w + carry;
    carry = (t < w) ? 1u : 0u;
    mag[i] = t;
  }
  return neg;
}

// k (canonical, 8 words) -> |k1|, sign(k1), |k2|, sign(k2)
__device__ __forceinline__ void decompose(const uint32_t (&k)[8], uint32_t (&m1)[4], bool& s1, uint32_t (&m2)[4], bool& s2) {
  uint32_t t1[11], t2[13];
  mul_low<8, 3, 11>(k, G1, t1);
  mul_low<8, 5, 13>(k, G2, t2);
  const uint32_t c1[2] = {t1[8], t1[9]};                           // < 2^64
  const uint32_t c2[4] = {t2[8], t2[9], t2[10], t2[11]};           // < 2^127
  uint32_t p[5], q[5], v[5];
  mul_low<2, 2, 5>(c1, A1, p);                                     // c1 a1 (< 2^128: exact in 5 words)
  mul_low<4, 4, 5>(c2, A2, q);                                     // c2 a2 mod 2^160
  uint64_t borrow = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) {                                    // v = k - p - q mod 2^160
    const uint64_t d = (uint64_t)k[i] - p[i] - q[i] - borrow;
    v[i] = (uint32_t)d;
    borrow = (d >> 32) ? ((~(d >> 32) + 1) & 3) : 0;               // 0, 1 or 2 borrowed
  }
  s1 = abs160(v, m1);
  mul_low<2, 4, 5>(c1, NB1, p);                                    // c1 (-b1) mod 2^160
  mul_low<4, 2, 5>(c2, A1, q);                                     // c2 b2 (b2 = a1)
  borrow = 0;
#pragma unroll
  for (int i = 0; i < 5; i++) {                                    // v = p - q mod 2^160
    const uint64_t d = (uint64_t)p[i] - q[i] - borrow;
    v[i] = (uint32_t)d;
    borrow = (d >> 32) ? 1 : 0;
  }
  s2 = abs160(v, m2);
}
}  // namespace glv

// digits of both halves: point i (k1, base P_i) in column i, point n + i (k2, base phi(P_i)) in column n + i of the [W][n2_pad] array
template <typename DIGIT>   // int16_t; int32_t when the two-level sort takes over (msm_two_level)
__global__ void __launch_bounds__(256) k_digits_glv(const uint32_t* __restrict__ scalars, DIGIT* __restrict__ digits, uint32_t n, uint32_t n2_pad, int c, int W) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n2_pad - 2 * n) {                         // padding columns: digit 0 = "no entry"
    for (int win = 0; win < W; win++) digits[(size_t)win * n2_pad + 2 * n + i] = 0;
  }
  if (i >= n) return;
  uint32_t w[8];
  load_words(scalars + (size_t)i * 8, w);
  fe c32;
#pragma unroll
  for (int k = 0; k < NL; k++) c32.l[k] = FrParams::FROM_EXT_CANON[k];
  fe s = fe_canon_lt2p<FrParams>(fe_mul<FrParams>(c32, fe_unpack<0>(w)));      // Montgomery-256 -> the integer k < r
  fe_pack(s, w);
  uint32_t m[2][4];
  bool neg[2];
  glv::decompose(w, m[0], neg[0], m[1], neg[1]);
  const uint32_t half = 1u << (c - 1), mask = (1u << c) - 1;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    uint32_t carry = 0;
    DIGIT* col = digits + (size_t)h * n + i;
    for (int win = 0; win < W; win++) {
      const int bit = win * c;
      uint32_t v = 0;
      if (bit < 128) {
        const int wi = bit >> 5, sh = bit & 31;
        uint64_t two = m[h][wi];
        if (wi + 1 < 4) two |= (uint64_t)m[h][wi + 1] << 32;
        v = (uint32_t)(two >> sh) & mask;
      }
      v += carry;
      // digits of +m lie in [-2^(c-1), 2^(c-1)); a negative half stores -d, so its recoding takes the mirrored range (-2^(c-1), 2^(c-1)]
      // (at c = 16 the digit +2^15 does not exist in int16: -(-2^15) would wrap)
      int32_t d;
      if (neg[h] ? (v > half) : (v >= half)) { d = (int32_t)v - (int32_t)(1u << c); carry = 1; } else { d = (int32_t)v; carry = 0; }
      col[(size_t)win * n2_pad] = (DIGIT)(neg[h] ? -d : d);
    }
  }
}

// endo[i] = phi(bases[i]) = (beta x, y) in the G1Affine memory format; the identity (0, 0) stays the identity
__global__ void __launch_bounds__(256) k_endo_bases(const uint32_t* __restrict__ bases, uint32_t n, uint32_t* __restrict__ endo) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  affine_words pt = load_affine(bases, i);
  uint32_t bw[8], xw[8];
#pragma unroll
  for (int k = 0; k < 8; k++) bw[k] = glv::BETA_EXT[k];
  const fe beta = fe_mul<Fq>(fe_one<Fq>(), fe_from_ext_lazy(bw));            // internal form
  fe_to_ext<Fq>(fe_mul<Fq>(beta, fe_from_ext_lazy(pt.x)), xw);
  uint4* o = reinterpret_cast<uint4*>(endo + (size_t)i * 16);
  o[0] = make_uint4(xw[0], xw[1], xw[2], xw[3]);
  o[1] = make_uint4(xw[4], xw[5], xw[6], xw[7]);
  o[2] = make_uint4(pt.y[0], pt.y[1], pt.y[2], pt.y[3]);
  o[3] = make_uint4(pt.y[4], pt.y[5], pt.y[6], pt.y[7]);
}

// ------------------------------------------------------------------------------------------------
// 2. count / 4. scatter: grid (chunks, W), LDS histogram of B = 2^(c-1) u32 counters
// ------------------------------------------------------------------------------------------------
// Both passes are latency-bound if written naively (one 2-byte load per thread per iteration): digits are read 8 at a
// time (16-byte loads; rows and chunks are padded to multiples of 8) and the returning global atomics of the reservation
// step are issued in independent batches.
template <bool SCATTER>
__global__ void __launch_bounds__(1024) k_sort_pass(const int16_t* __restrict__ digits, uint32_t n_pad, uint32_t chunk,
                                                    int c, uint32_t* __restrict__ count_or_cursor,
                                                    uint32_t* __restrict__ sorted, uint32_t win_bucket_stride,
                                                    uint32_t ref_base, uint32_t ref_stride) {
  extern __shared__ uint32_t hist[];
  const uint32_t B = 1u << (c - 1);
  const int win = blockIdx.y;
  const uint32_t kb = blockIdx.z, K = gridDim.z;          // batch: MSM kb of K (own bucket set, same bases)
  const uint32_t lo = blockIdx.x * chunk;                 // multiple of 8
  const uint32_t hi = min(n_pad, lo + chunk);             // multiple of 8
  for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) hist[b] = 0;
  __syncthreads();
  const uint4* dv = reinterpret_cast<const uint4*>(digits + ((size_t)win * K + kb) * n_pad);
  const uint32_t v_lo = lo >> 3, v_hi = hi >> 3;
  for (uint32_t vi = v_lo + threadIdx.x; vi < v_hi; vi += blockDim.x) {
    const uint4 q = dv[vi];
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int d = (int16_t)(w[k >> 1] >> ((k & 1) * 16));
      if (d != 0) atomicAdd(&hist[(d < 0 ? -d : d) - 1], 1u);
    }
  }
  __syncthreads();
  // bucket set: general path (K = 1): one per window; prepared path: one per MSM of the batch, shared by its windows
  uint32_t* g = count_or_cursor + (size_t)win * win_bucket_stride + (size_t)kb * B;
  if (!SCATTER) {
    for (uint32_t b = threadIdx.x; b < B; b += blockDim.x) {
      uint32_t v = hist[b];
      if (v) atomicAdd(&g[b], v);
    }
  } else {
    // reserve [pos, pos + v) in each non-empty bucket's slot range: 8 independent returning atomics in flight per thread
    for (uint32_t b0 = threadIdx.x; b0 < B; b0 += 8 * blockDim.x) {
      uint32_t v[8], r[8];
#pragma unroll
      for (int k = 0; k < 8; k++) { const uint32_t b = b0 + k * blockDim.x; v[k] = b < B ? hist[b] : 0u; }
#pragma unroll
      for (int k = 0; k < 8; k++) r[k] = v[k] ? atomicAdd(&g[b0 + k * blockDim.x], v[k]) : 0u;
#pragma unroll
      for (int k = 0; k < 8; k++) { const uint32_t b = b0 + k * blockDim.x; if (b < B) hist[b] = r[k]; }
    }
    __syncthreads();
    const uint32_t rbase = ref_base + (uint32_t)win * ref_stride;       // prepared: window w reads table slice w
    for (uint32_t vi = v_lo + threadIdx.x; vi < v_hi; vi += blockDim.x) {
      const uint4 q = dv[vi];
      const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int d = (int16_t)(w[k >> 1] >> ((k & 1) * 16));
        if (d != 0) {
          const uint32_t pos = atomicAdd(&hist[(d < 0 ? -d : d) - 1], 1u);
          sorted[pos] = (rbase + vi * 8 + k) | (d < 0 ? 0x80000000u : 0u);
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// 2'/4'. the two-level sort: wide windows (16 < c <= 21, prepared path only) and, since round 3, 16-bit windows with enough entries per
//     group (msm_two_level: the general path's GLV MSMs, prepared tables of 2^16 ... 2^19 points).  Fewer windows = fewer additions
//     (c = 20: 13 instead of 16), but 2^(c-1) buckets no longer fit an LDS histogram -- and at any window size the one-level scatter of
//     step 4 stores every reference with its own 4-byte write.  Two levels:
//       coarse  bucket >> 12 selects one of G = 2^(c-13) groups (128 at c = 20; with a bucket set per window: window * 8 + (bucket >> 12));
//               a (window, chunk) workgroup counts / places its entries per group (G LDS counters, one ranged global atomic per group on
//               the cursor of ITS window) into a staging array (ref u32 + fine u16) -- a group's region holds its windows one after the other;
//       fine    per group, the 2^12 "fine" buckets: LDS histogram, atomics into the global counts (k_fine_count), scan, then
//               k_fine_sorted: reservation and a counting sort of the chunk inside LDS, so that bucket runs leave as runs.
// ------------------------------------------------------------------------------------------------

// (Recomputing the digits from the scalars in both passes instead of storing them as int32 was tried and measured slower:
// 0.139 vs 0.117 ms at 2^20, 1.75 vs 1.3 ms at 2^24.)
// The group counts as a pass of their own over the stored digits: what the pipeline ran until round 3, when the counts moved into the digit
// kernel (k_digits_wide); kept as the A/B reference (ZKHIP_NO_FUSED_COUNT=1).
__global__ void __launch_bounds__(1024) k_coarse_count(const int32_t* __restrict__ digits, uint32_t n_pad, uint32_t chunk, uint32_t* __restrict__ gcount, uint32_t gstride) {
  __shared__ uint32_t cnt[MAX_GROUPS];
  const int win = blockIdx.y;
  const uint32_t lo = blockIdx.x * chunk, hi = min(n_pad, lo + chunk);   // multiples of 8
  if (threadIdx.x < MAX_GROUPS) cnt[threadIdx.x] = 0;
  __syncthreads();
  const uint4* dv = reinterpret_cast<const uint4*>(digits + (size_t)win * n_pad);
  const uint32_t v_lo = lo >> 2, v_hi = hi >> 2;
  for (uint32_t vi = v_lo + threadIdx.x; vi < v_hi; vi += 4 * blockDim.x) {        // four vector loads in flight per lane
    uint4 q[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const uint32_t vj = vi + u * blockDim.x; q[u] = vj < v_hi ? dv[vj] : make_uint4(0, 0, 0, 0); }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int32_t d4[4] = {(int32_t)q[u].x, (int32_t)q[u].y, (int32_t)q[u].z, (int32_t)q[u].w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int32_t d = d4[k];
        if (d != 0) atomicAdd(&cnt[(uint32_t)win * gstride + (((uint32_t)(d < 0 ? -d : d) - 1) >> FINE_BITS)], 1u);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < MAX_GROUPS && cnt[threadIdx.x]) atomicAdd(&gcount[win * MAX_GROUPS + threadIdx.x], cnt[threadIdx.x]);
}

// Coarse placement with the chunk sorted by group in LDS first (same reasoning as k_fine_sorted below): a workgroup takes
// COARSE_CHUNK digits of one window, counting-sorts (reference, fine bucket) by group inside LDS and writes each group's run --
// COARSE_CHUNK / G = 64 entries on average -- as neighbouring words.  LDS: cur[G] | delta[G] | refs[CH] | fines[CH] | mask, wpre [CH / 32].
constexpr uint32_t COARSE_CHUNK = 8192;
__global__ void __launch_bounds__(1024) k_coarse_sorted(const int32_t* __restrict__ digits, uint32_t n_pad, uint32_t* __restrict__ gcursor,
                                                        uint32_t* __restrict__ stage_ref, uint16_t* __restrict__ stage_fine,
                                                        uint32_t ref_base, uint32_t ref_stride, uint32_t gstride) {
  __shared__ uint32_t cur[MAX_GROUPS], delta[MAX_GROUPS];
  __shared__ uint32_t refs[COARSE_CHUNK];
  __shared__ uint16_t fines[COARSE_CHUNK];
  __shared__ uint32_t mask[COARSE_CHUNK / 32], wpre[COARSE_CHUNK / 32];      // run-start bits of the sorted chunk and their word prefixes
  const int win = blockIdx.y;
  // gstride = 0: one bucket set shared by all windows (prepared tables), group = bucket >> 12.  gstride = B >> 12: a bucket set per window
  // (general path), group = window * gstride + (bucket >> 12) -- the groups of a window are contiguous, as its buckets are
  const uint32_t gbase = (uint32_t)win * gstride;
  const uint32_t lo = blockIdx.x * COARSE_CHUNK, hi = min(n_pad, lo + COARSE_CHUNK);   // multiples of 8
  if (lo >= hi) return;
  if (threadIdx.x < MAX_GROUPS) cur[threadIdx.x] = 0;
  if (threadIdx.x < COARSE_CHUNK / 32) mask[threadIdx.x] = 0;
  __syncthreads();
  const uint4* dv = reinterpret_cast<const uint4*>(digits + (size_t)win * n_pad);
  const uint32_t v_lo = lo >> 2, v_hi = hi >> 2;
  // COARSE_CHUNK / 4 = 2048 vectors for 1024 lanes: a lane's two vectors are loaded together (independent loads), counted, and kept in
  // registers for the placement below -- the digits are read once
  static_assert(COARSE_CHUNK == 8192, "a lane owns two 4-digit vectors of the chunk");
  const uint32_t vi = v_lo + threadIdx.x, vj = vi + 1024u;
  const uint4 q0 = vi < v_hi ? dv[vi] : make_uint4(0, 0, 0, 0), q1 = vj < v_hi ? dv[vj] : make_uint4(0, 0, 0, 0);
  const int32_t d8[8] = {(int32_t)q0.x, (int32_t)q0.y, (int32_t)q0.z, (int32_t)q0.w, (int32_t)q1.x, (int32_t)q1.y, (int32_t)q1.z, (int32_t)q1.w};
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int32_t d = d8[k];
    if (d != 0) atomicAdd(&cur[gbase + (((uint32_t)(d < 0 ? -d : d) - 1) >> FINE_BITS)], 1u);
  }
  __syncthreads();
  if (threadIdx.x < 64) {            // one wavefront: exclusive scan of the MAX_GROUPS = 128 counts (two per lane), then the reservations
    const uint32_t c0 = cur[2 * threadIdx.x], c1 = cur[2 * threadIdx.x + 1], s = c0 + c1;
    uint32_t x = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t y = __shfl_up(x, off, 64);
      if ((int)threadIdx.x >= off) x += y;
    }
    const uint32_t l0 = x - s, l1 = l0 + c0;
    // The reservation cursors are per (window, group): a group's staging region is laid out window after window (k_group_offsets).  With one
    // cursor per group every workgroup of every window hit the same 128 words -- 6656 workgroups at 2^22 -- and the kernel's time WAS those
    // atomics: it scaled with the number of workgroups, not with the bytes (0.27 / 0.50 / 0.97 ms at 2^22 with 8 Ki / 4 Ki / 2 Ki-digit chunks:
    // ~40 ns per same-address atomic, profiles/r03_msm_experiments_ab.txt J).
    uint32_t* const wcur = gcursor + (uint32_t)win * MAX_GROUPS;
    const uint32_t g0 = c0 ? atomicAdd(&wcur[2 * threadIdx.x], c0) : 0u, g1 = c1 ? atomicAdd(&wcur[2 * threadIdx.x + 1], c1) : 0u;
    cur[2 * threadIdx.x] = l0; cur[2 * threadIdx.x + 1] = l1;
    // the way out without a search (as in k_fine_sorted): run-start bits, word prefixes, and delta indexed by a run's rank among the
    // non-empty groups
    const uint32_t nz = (c0 != 0) + (c1 != 0);
    uint32_t y = nz;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(y, off, 64);
      if ((int)threadIdx.x >= off) y += t;
    }
    uint32_t rank = y - nz;
    if (c0) { delta[rank++] = g0 - l0; atomicOr(&mask[l0 >> 5], 1u << (l0 & 31)); }
    if (c1) { delta[rank] = g1 - l1; atomicOr(&mask[l1 >> 5], 1u << (l1 & 31)); }
    // word prefixes of the 256 mask words, four per lane (LDS operations of one wavefront complete in order: the bits are set; the
    // fence keeps the compiler from moving the reads above the atomics)
    __threadfence_block();
    uint32_t pc[4], ps = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { pc[k] = (uint32_t)__popc(mask[4 * threadIdx.x + k]); ps += pc[k]; }
    uint32_t z = ps;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(z, off, 64);
      if ((int)threadIdx.x >= off) z += t;
    }
    uint32_t zb = z - ps;
#pragma unroll
    for (int k = 0; k < 4; k++) { wpre[4 * threadIdx.x + k] = zb; zb += pc[k]; }
  }
  __syncthreads();
  const uint32_t rbase = ref_base + (uint32_t)win * ref_stride;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int32_t d = d8[k];
    if (d != 0) {
      const uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1;
      const uint32_t pos = atomicAdd(&cur[gbase + (b >> FINE_BITS)], 1u);
      refs[pos] = (rbase + (k < 4 ? vi : vj) * 4 + (k & 3)) | (d < 0 ? 0x80000000u : 0u);
      fines[pos] = (uint16_t)(b & ((1u << FINE_BITS) - 1));
    }
  }
  __syncthreads();
  const uint32_t total = cur[MAX_GROUPS - 1];                  // cur[g] = end of group g's run
  for (uint32_t i = threadIdx.x; i < total; i += blockDim.x) {
    const uint32_t r = wpre[i >> 5] + (uint32_t)__popc(mask[i >> 5] & ((2u << (i & 31)) - 1u)) - 1u;      // rank of the run position i lies in
    const uint32_t pos = delta[r] + i;
    stage_ref[pos] = refs[i];
    stage_fine[pos] = fines[i];
  }
}

constexpr uint32_t SORT_CHUNK = 28672;   // entries per workgroup of the sorted fine scatter: 112 KiB of references in LDS

// goff[g] = start of group g in the staging array (exclusive scan of the group counts), gcursor = copy; cstart[g] = index of the
// group's first SORT_CHUNK-sized chunk in the numbering of k_fine_sorted's workgroups
__global__ void __launch_bounds__(64) k_group_offsets(const uint32_t* __restrict__ wcount, int W, int G, uint32_t* __restrict__ goff, uint32_t* __restrict__ wcursor,
                                                       uint32_t* __restrict__ cstart, uint32_t chunk) {
  // one wavefront, groups 2 lane and 2 lane + 1 per lane (G <= MAX_GROUPS = 128): two exclusive scans by shuffles (the serial loop
  // this replaces took 12 us: 128 dependent global loads).  wcount[w][g] = entries of window w in group g; a group's region holds its
  // windows one after the other and wcursor[w][g] is where window w's workgroups start to reserve.
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  uint32_t c0 = 0, c1 = 0;
  for (int w = 0; w < W; w++) {
    if (2 * lane < G) c0 += wcount[w * MAX_GROUPS + 2 * lane];
    if (2 * lane + 1 < G) c1 += wcount[w * MAX_GROUPS + 2 * lane + 1];
  }
  const uint32_t k0 = (c0 + chunk - 1) / chunk, k1 = (c1 + chunk - 1) / chunk;
  uint32_t x = c0 + c1, y = k0 + k1;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t xs = __shfl_up(x, off, 64), ys = __shfl_up(y, off, 64);
    if (lane >= off) { x += xs; y += ys; }
  }
  const uint32_t ex = x - (c0 + c1), ey = y - (k0 + k1);
  if (2 * lane < G) { goff[2 * lane] = ex; cstart[2 * lane] = ey; }
  if (2 * lane + 1 < G) { goff[2 * lane + 1] = ex + c0; cstart[2 * lane + 1] = ey + k0; }
  uint32_t r0 = ex, r1 = ex + c0;
  for (int w = 0; w < W; w++) {
    if (2 * lane < G) { wcursor[w * MAX_GROUPS + 2 * lane] = r0; r0 += wcount[w * MAX_GROUPS + 2 * lane]; }
    if (2 * lane + 1 < G) { wcursor[w * MAX_GROUPS + 2 * lane + 1] = r1; r1 += wcount[w * MAX_GROUPS + 2 * lane + 1]; }
  }
  if (lane == 63) { goff[G] = x; cstart[G] = y; }
}

// Fine scatter with the chunk sorted in LDS first.  A plain placement (as in k_sort_pass<true>) stores every reference with its own
// 4-byte write, and PMC shows what that costs: 8.7 bytes reach memory per byte of payload (32-byte sectors).  Here a workgroup
// owns SORT_CHUNK staged entries of one group of 2^FINE_BITS buckets, counting-sorts their references by bucket inside LDS, and
// writes bucket runs: neighbouring lanes store neighbouring words (~7 entries per bucket and chunk).
//   LDS: cur[FB] (histogram -> run starts -> run ends) | delta[FB] (global slot of a run minus its LDS position) | refs[SORT_CHUNK]
//        | run-start bit mask and its word prefixes (SORT_CHUNK / 16 words)
// (A second shape that kept every entry's 16-bit bucket id in LDS -- 20 Ki entries per chunk -- served the smallest wide size until the
// rank structure below made the way out three reads for any chunk: 2^20 scatter 0.110 -> 0.101 ms, and the variant was removed.)
// LDS histogram of the fine bucket ids of staged entries [start, end).  The sort kernels spent 77 - 84 % of their cycles parked
// (profiles/r02_rocprofv3_pmc_sq_issue.txt): with one load per lane and iteration, each feeding an LDS atomic, every trip exposes a full
// memory latency.  Four independent loads are issued before the first atomic.
#ifndef ZKHIP_FINE_BATCH
#define ZKHIP_FINE_BATCH 7
#endif
constexpr int FINE_BATCH = ZKHIP_FINE_BATCH;      // measured (2 / 4 / 7 / 14 / 28 loads in flight per lane): see (profiles/r03_msm_sort_pipelining_ab.txt)
__device__ __forceinline__ void fine_histogram(const uint16_t* __restrict__ stage_fine, uint32_t start, uint32_t end, uint32_t* __restrict__ cur) {
  for (uint32_t p = start + threadIdx.x; p < end; p += FINE_BATCH * blockDim.x) {
    uint32_t f[FINE_BATCH];
#pragma unroll
    for (int k = 0; k < FINE_BATCH; k++) { const uint32_t q = p + k * blockDim.x; f[k] = q < end ? stage_fine[q] : 0xffffffffu; }
#pragma unroll
    for (int k = 0; k < FINE_BATCH; k++) if (f[k] != 0xffffffffu) atomicAdd(&cur[f[k]], 1u);
  }
}

__global__ void __launch_bounds__(1024) k_fine_sorted(const uint16_t* __restrict__ stage_fine, const uint32_t* __restrict__ stage_ref,
                                                      const uint32_t* __restrict__ goff, const uint32_t* __restrict__ cstart, int G,
                                                      uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted) {
  extern __shared__ uint32_t hist[];
  constexpr uint32_t FB = 1u << FINE_BITS;
  constexpr uint32_t CHUNK = SORT_CHUNK;
  static_assert(FB == 4096, "one thread owns four buckets in the scan");
  __shared__ uint32_t wsum[16];
  uint32_t* cur = hist;
  uint32_t* delta = hist + FB;
  uint32_t* refs = hist + 2 * FB;
  const uint32_t w = blockIdx.x;
  if (w >= cstart[G]) return;
  uint32_t glo = 0, ghi = (uint32_t)G;                      // group of chunk w: last g with cstart[g] <= w
  while (ghi - glo > 1) { const uint32_t mid = (glo + ghi) >> 1; if (cstart[mid] <= w) glo = mid; else ghi = mid; }
  const uint32_t g = glo;
  const uint32_t start = goff[g] + (w - cstart[g]) * CHUNK, end = min(goff[g + 1], start + CHUNK), cnt_n = end - start;
  for (uint32_t b = threadIdx.x; b < FB; b += blockDim.x) cur[b] = 0;
  __syncthreads();
  fine_histogram(stage_fine, start, end, cur);
  __syncthreads();
  // exclusive scan over the 4096 counts: thread t owns buckets 4t .. 4t + 3
  uint32_t c[4], l[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) { c[k] = cur[4 * threadIdx.x + k]; s += c[k]; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t x = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  uint32_t before = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) if (i < wave) before += wsum[i];
  uint32_t run = before + x - s;
  uint32_t* gl = cursor + (size_t)g * FB;
  uint32_t gb[4];
#pragma unroll
  for (int k = 0; k < 4; k++) { l[k] = run; run += c[k]; }
#pragma unroll
  for (int k = 0; k < 4; k++) gb[k] = c[k] ? atomicAdd(&gl[4 * threadIdx.x + k], c[k]) : 0u;      // reserve the runs' slots
  {
    // Way out without a search: a bit per sorted position marks the run starts, wpre[w] counts the starts before 32-position word w, and
    // delta is indexed by a run's RANK among the non-empty buckets -- entry i then finds its slot with three LDS reads (mask word and
    // word prefix, then delta) where a binary search over the 4096 run ends took twelve dependent ones (30 % of this kernel at 2^22).
    uint32_t* mask = refs + CHUNK;                             // CHUNK / 32 words
    uint32_t* wpre = mask + CHUNK / 32;                        // CHUNK / 32 words
    const uint32_t nz = (c[0] != 0) + (c[1] != 0) + (c[2] != 0) + (c[3] != 0);
    uint32_t y = nz;                                           // rank of this thread's first non-empty bucket: a second scan, same shape
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(y, off, 64);
      if (lane >= off) y += t;
    }
    __syncthreads();                                           // wsum is read by everyone above
    if (lane == 63) wsum[wave] = y;
    for (uint32_t w2 = threadIdx.x; w2 < CHUNK / 32; w2 += blockDim.x) mask[w2] = 0;
    __syncthreads();
    uint32_t rank = y - nz;
#pragma unroll
    for (int i = 0; i < 16; i++) if (i < wave) rank += wsum[i];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      cur[4 * threadIdx.x + k] = l[k];
      if (c[k]) {
        delta[rank++] = gb[k] - l[k];
        atomicOr(&mask[l[k] >> 5], 1u << (l[k] & 31));
      }
    }
    __syncthreads();
    // word prefixes: CHUNK / 32 = 896 <= 1024 words, one per thread
    const uint32_t pc = threadIdx.x < CHUNK / 32 ? (uint32_t)__popc(mask[threadIdx.x]) : 0u;
    uint32_t z = pc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(z, off, 64);
      if (lane >= off) z += t;
    }
    __syncthreads();
    if (lane == 63) wsum[wave] = z;
    __syncthreads();
    uint32_t zb = z - pc;
#pragma unroll
    for (int i = 0; i < 16; i++) if (i < wave) zb += wsum[i];
    if (threadIdx.x < CHUNK / 32) wpre[threadIdx.x] = zb;
  }
  __syncthreads();
  {
    // (FINE_BATCH entries per lane in flight, as in fine_histogram)
    for (uint32_t p = start + threadIdx.x; p < end; p += FINE_BATCH * blockDim.x) {
      uint32_t f[FINE_BATCH], r[FINE_BATCH];
#pragma unroll
      for (int k = 0; k < FINE_BATCH; k++) { const uint32_t q = p + k * blockDim.x; f[k] = q < end ? stage_fine[q] : 0xffffffffu; r[k] = q < end ? stage_ref[q] : 0u; }
#pragma unroll
      for (int k = 0; k < FINE_BATCH; k++) if (f[k] != 0xffffffffu) refs[atomicAdd(&cur[f[k]], 1u)] = r[k];
    }
    __syncthreads();
    const uint32_t* mask = refs + CHUNK;
    const uint32_t* wpre = mask + CHUNK / 32;
    for (uint32_t i = threadIdx.x; i < cnt_n; i += blockDim.x) {
      const uint32_t r = wpre[i >> 5] + (uint32_t)__popc(mask[i >> 5] & ((2u << (i & 31)) - 1u)) - 1u;     // rank of the run position i lies in
      sorted[delta[r] + i] = refs[i];
    }
  }
}

// count pass of the same tiling (one workgroup per SORT_CHUNK entries of a group): LDS histogram, non-empty counts merged into the
// global per-bucket counts
__global__ void __launch_bounds__(1024) k_fine_count(const uint16_t* __restrict__ stage_fine, const uint32_t* __restrict__ goff,
                                                     const uint32_t* __restrict__ cstart, int G, uint32_t* __restrict__ count, uint32_t chunk) {
  constexpr uint32_t FB = 1u << FINE_BITS;
  __shared__ uint32_t cur[FB];
  const uint32_t w = blockIdx.x;
  if (w >= cstart[G]) return;
  uint32_t glo = 0, ghi = (uint32_t)G;
  while (ghi - glo > 1) { const uint32_t mid = (glo + ghi) >> 1; if (cstart[mid] <= w) glo = mid; else ghi = mid; }
  const uint32_t g = glo;
  const uint32_t start = goff[g] + (w - cstart[g]) * chunk, end = min(goff[g + 1], start + chunk);
  for (uint32_t b = threadIdx.x; b < FB; b += blockDim.x) cur[b] = 0;
  __syncthreads();
  fine_histogram(stage_fine, start, end, cur);
  __syncthreads();
  uint32_t* gl = count + (size_t)g * FB;
  for (uint32_t b = threadIdx.x; b < FB; b += blockDim.x) {
    const uint32_t v = cur[b];
    if (v) atomicAdd(&gl[b], v);
  }
}

// ------------------------------------------------------------------------------------------------
// 3. exclusive scans of the bucket counts (three small kernels).  MODE 0: identity, MODE 1: ceil(x / 2^task_shift)
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_BLOCK = 256, SCAN_PER_THREAD = 4,     // (16 -> 4: 4 x the workgroups and 16-byte lane strides: scan phase 0.024 -> 0.0155 ms at 2^20)
          SCAN_TILE = SCAN_BLOCK * SCAN_PER_THREAD;

template <int MODE>
__device__ __forceinline__ uint32_t scan_xform(uint32_t v, uint32_t task_shift) {
  return MODE == 0 ? v : (v + (1u << task_shift) - 1) >> task_shift;
}

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds, uint32_t& total) {
  // 256 threads; wave scan via shuffles then 4 wave totals through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  if (lane == 63) lds[wave] = x;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < SCAN_BLOCK / 64; i++) {
    uint32_t t = lds[i];
    if (i < wave) base += t;
    tot += t;
  }
  __syncthreads();
  total = tot;
  return base + x - v;
}

// Both scans of the bucket counts in one pass (bucket sets above SCAN_SINGLE_MAX): the offsets (MODE 0) and the task offsets (MODE 1:
// ceil(count / 2^task_shift)) read the same array, and the second scan's three launches (~5 us each, 1 % of a 2^20 MSM) disappear.
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_sums2(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ sums0,
                                                           uint32_t* __restrict__ sums1, uint32_t task_shift) {
  __shared__ uint32_t lds[4];
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
  uint32_t s0 = 0, s1 = 0;
#pragma unroll
  for (int i = 0; i < SCAN_PER_THREAD; i++) if (base + i < n) { const uint32_t v = in[base + i]; s0 += v; s1 += scan_xform<1>(v, task_shift); }
  uint32_t t0, t1;
  block_exclusive_scan(s0, lds, t0);
  block_exclusive_scan(s1, lds, t1);
  if (threadIdx.x == 0) { sums0[blockIdx.x] = t0; sums1[blockIdx.x] = t1; }
}

__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_top2(uint32_t* __restrict__ sums0, uint32_t* __restrict__ sums1, uint32_t nb,
                                                          uint32_t* __restrict__ total0, uint32_t* __restrict__ total1) {
  __shared__ uint32_t lds[4];
  uint32_t run0 = 0, run1 = 0;
  for (uint32_t base = 0; base < nb; base += SCAN_BLOCK) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v0 = i < nb ? sums0[i] : 0, v1 = i < nb ? sums1[i] : 0;
    uint32_t t0, t1;
    const uint32_t e0 = block_exclusive_scan(v0, lds, t0), e1 = block_exclusive_scan(v1, lds, t1);
    if (i < nb) { sums0[i] = run0 + e0; sums1[i] = run1 + e1; }
    run0 += t0; run1 += t1;
  }
  if (threadIdx.x == 0) { *total0 = run0; *total1 = run1; }
}

__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_apply2(const uint32_t* __restrict__ in, uint32_t n, const uint32_t* __restrict__ sums0,
                                                            const uint32_t* __restrict__ sums1, const uint32_t* __restrict__ total0,
                                                            const uint32_t* __restrict__ total1, uint32_t* __restrict__ offset,
                                                            uint32_t* __restrict__ cursor, uint32_t* __restrict__ task_off, uint32_t task_shift) {
  __shared__ uint32_t lds[4];
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
  uint32_t v[SCAN_PER_THREAD], s0 = 0, s1 = 0;
#pragma unroll
  for (int i = 0; i < SCAN_PER_THREAD; i++) { v[i] = base + i < n ? in[base + i] : 0; s0 += v[i]; s1 += scan_xform<1>(v[i], task_shift); }
  uint32_t t0, t1;
  uint32_t e0 = block_exclusive_scan(s0, lds, t0) + sums0[blockIdx.x];
  uint32_t e1 = block_exclusive_scan(s1, lds, t1) + sums1[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_PER_THREAD; i++) {
    if (base + i < n) { offset[base + i] = e0; cursor[base + i] = e0; task_off[base + i] = e1; }
    e0 += v[i];
    e1 += scan_xform<1>(v[i], task_shift);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { offset[n] = *total0; task_off[n] = *total1; }
}

// The same scan as ONE workgroup, for up to SCAN_SINGLE_MAX values = one tile of 1024 threads x 8 (bucket sets of c <= 14: the
// small, latency-bound MSMs): three dependent launches become one, and with MODE 1 the task-length histogram and its cursors
// (k_make_tasks' atomics and k_order_offsets) come out of the same pass -- five launches of ~5 us less per MSM (2^13: 0.254 ->
// 0.245 ms).  More tiles in one workgroup lose: 2^15 values take 47 us this way against 17 us with the three kernels.
constexpr uint32_t SCAN_SINGLE_MAX = 8192;
template <int MODE>
__global__ void __launch_bounds__(1024) k_scan_single(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ total_out,
                                                      uint32_t* __restrict__ out, uint32_t* __restrict__ out2, uint32_t task_shift,
                                                      uint32_t* __restrict__ len_cursor) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t lh[MAX_TASK_LEN + 1];
  if (MODE == 1 && threadIdx.x <= MAX_TASK_LEN) lh[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t full = 1u << task_shift;
  uint32_t running = 0, n_full = 0;
  for (uint32_t tile = 0; tile < n; tile += 8192) {
    const uint32_t base = tile + threadIdx.x * 8;
    uint32_t v[8], s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const uint32_t raw = base + i < n ? in[base + i] : 0u;
      v[i] = scan_xform<MODE>(raw, task_shift);
      s += v[i];
      if (MODE == 1 && raw) {                                  // v[i] - 1 full tasks and one of the remaining length
        n_full += v[i] - 1;
        atomicAdd(&lh[raw - ((v[i] - 1) << task_shift)], 1u);
      }
    }
    uint32_t x = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t y = __shfl_up(x, off, 64);
      if (lane >= off) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const uint32_t t = wsum[i];
      if (i < wave) before += t;
      tot += t;
    }
    __syncthreads();
    uint32_t ex = running + before + x - s;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (base + i < n) { out[base + i] = ex; if (out2) out2[base + i] = ex; }
      ex += v[i];
    }
    running += tot;
  }
  if (threadIdx.x == 0) { out[n] = running; *total_out = running; }
  if (MODE == 1) {
    if (n_full) atomicAdd(&lh[full], n_full);
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t run = 0;
      for (int L = MAX_TASK_LEN; L >= 1; L--) { len_cursor[L] = run; run += lh[L]; }
    }
  }
}

constexpr uint32_t HEAVY_SLICE = 2048;   // partials of a heavy bucket one workgroup of the combine step sums (16 per thread)
// ------------------------------------------------------------------------------------------------
// 5. tasks: bucket k with cnt entries -> ceil(cnt / 2^task_shift) tasks; max_parts = largest task count of any bucket
// ------------------------------------------------------------------------------------------------
// One thread per bucket; a bucket with more than COOP_TASKS tasks is written by the whole wavefront instead (its owner's values
// are broadcast, lane l writes records l, l + 64, ...).  A column of selector bits or of many equal witness values puts a large share
// of all entries into one bucket: with the owner writing alone, 5 % equal scalars cost 0.9 ms at 2^22 and an all-ones column 11 ms
// (65536 records from one thread, here and again in k_make_order).  (A thread-per-task variant with a binary search for the bucket was
// measured 2-5x slower in the common case: 0.28 vs 0.05 ms at 2^22.)
constexpr uint32_t COOP_TASKS = 32;
__global__ void __launch_bounds__(256) k_make_tasks(const uint32_t* __restrict__ offset, const uint32_t* __restrict__ task_off,
                                                    uint32_t nbuckets, task_t* __restrict__ tasks, uint32_t task_shift,
                                                    uint32_t* __restrict__ max_parts, uint32_t* __restrict__ len_count, uint32_t seq_parts,
                                                    uint32_t* __restrict__ heavy_count, uint2* __restrict__ heavy, uint32_t heavy_cap) {
  __shared__ uint32_t lh[MAX_TASK_LEN + 1];
  if (threadIdx.x <= MAX_TASK_LEN) lh[threadIdx.x] = 0;
  __syncthreads();
  // a capped grid walks the buckets: the length histogram costs one global atomic per (workgroup, length) on the same 65 words,
  // and with one workgroup per 256 buckets those were 2048 x 65 contended atomics at 2^19 buckets (99 us of a 6 ms MSM)
  uint32_t nt = 0;
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t base = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); base < nbuckets; base += gridDim.x * blockDim.x) {   // uniform per wavefront
    const uint32_t k = base + lane;
    uint32_t s = 0, e = 0, t = 0, m = 0;
    if (k < nbuckets) { s = offset[k]; e = offset[k + 1]; t = task_off[k]; m = task_off[k + 1] - t; }
    const bool coop = m > COOP_TASKS;
    if (!coop) {
      for (uint32_t j = 0; j < m; j++) {
        task_t tk;
        tk.bucket = k;
        tk.start = s + (j << task_shift);
        tk.len = min(1u << task_shift, e - tk.start);
        tasks[t + j] = tk;
      }
    }
    if (m) {   // m - 1 full tasks and one of the remaining length
      if (m > 1) atomicAdd(&lh[1u << task_shift], m - 1);
      atomicAdd(&lh[(e - s) - ((m - 1) << task_shift)], 1u);
    }
    if (m > seq_parts) {      // more partials than the per-bucket step of the combine kernel sums: listed for its heavy-bucket workgroups,
      const uint32_t P = (m + HEAVY_SLICE - 1) / HEAVY_SLICE;     // one entry (bucket, slice) per HEAVY_SLICE partials, consecutive in the list
      const uint32_t idx = atomicAdd(heavy_count, P);
      for (uint32_t i = 0; i < P; i++)
        if (idx + i < heavy_cap) heavy[idx + i] = make_uint2(k, i);   // (the cap cannot be exceeded: msm_lay_out)
    }
    nt = max(nt, m);
    unsigned long long big = __ballot(coop);
    while (big) {
      const int src = __ffsll((long long)big) - 1;
      big &= big - 1;
      const uint32_t bk = (uint32_t)__shfl((int)k, src, 64), bs = (uint32_t)__shfl((int)s, src, 64), be = (uint32_t)__shfl((int)e, src, 64),
                     bt = (uint32_t)__shfl((int)t, src, 64), bm = (uint32_t)__shfl((int)m, src, 64);
      for (uint32_t j = lane; j < bm; j += 64) {
        task_t tk;
        tk.bucket = bk;
        tk.start = bs + (j << task_shift);
        tk.len = min(1u << task_shift, be - tk.start);
        tasks[bt + j] = tk;
      }
    }
  }
  __syncthreads();
  if (len_count && threadIdx.x <= MAX_TASK_LEN && lh[threadIdx.x]) atomicAdd(&len_count[threadIdx.x], lh[threadIdx.x]);   // nullptr: k_scan_single<1> made the histogram
  // block max -> one atomic per wave
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) nt = max(nt, (uint32_t)__shfl_xor((int)nt, off, 64));
  if ((threadIdx.x & 63) == 0 && nt > 1) atomicMax(max_parts, nt);
}

// Execution order of the tasks: longest first, so that the 64 tasks of a wave have equal length (a wave runs as long as its
// longest task) and the launch tail consists of short tasks.  len_cursor[L] = first slot of length L.
__global__ void k_order_offsets(const uint32_t* __restrict__ len_count, uint32_t* __restrict__ len_cursor) {
  // one wavefront: lengths MAX_TASK_LEN - 2 lane and MAX_TASK_LEN - 2 lane - 1 per lane, an exclusive scan by shuffles (longest first).  (The
  // single-thread loop this replaces was 128 dependent loads and stores: 5 us.)
  static_assert(MAX_TASK_LEN == 128, "two lengths per lane of one wavefront");
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  const int lane = threadIdx.x, L0 = MAX_TASK_LEN - 2 * lane, L1 = L0 - 1;         // L0 = 128 ... 2, L1 = 127 ... 1
  const uint32_t c0 = len_count[L0], c1 = len_count[L1];
  uint32_t x = c0 + c1;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  const uint32_t ex = x - (c0 + c1);
  len_cursor[L0] = ex;
  len_cursor[L1] = ex + c0;
}

// Within the class of full-length tasks the order is j-major per workgroup (j = the task's index inside its
// bucket): a wavefront of k_accumulate then runs 64 tasks that sit at the same depth of 64 neighbouring buckets.  The sort fills
// every bucket roughly window by window, so those 64 tasks gather their points from the same one or two window slices of the
// prepared table instead of from all of it -- the table of a 2^24-point MSM is 13 GiB, and the gathers' address-translation
// footprint, not their bandwidth, is what costs (an emulated dense 16 GiB footprint slows the 2^20 accumulation by 11 %).
constexpr int ORDER_JMAX = 64;
__global__ void __launch_bounds__(256) k_make_order(const uint32_t* __restrict__ offset, const uint32_t* __restrict__ task_off,
                                                    uint32_t nbuckets, uint32_t task_shift, uint32_t* __restrict__ len_cursor,
                                                    const uint32_t* __restrict__ sorted, uint4* __restrict__ order) {
  __shared__ uint32_t lh[MAX_TASK_LEN + 1];
  __shared__ uint32_t jcnt[ORDER_JMAX + 1], joff[ORDER_JMAX + 1];
  if (threadIdx.x <= MAX_TASK_LEN) lh[threadIdx.x] = 0;
  if (threadIdx.x <= ORDER_JMAX) jcnt[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t full = 1u << task_shift, step = gridDim.x * blockDim.x;     // capped grid, as k_make_tasks
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nbuckets; k += step) {
    const uint32_t nt = task_off[k + 1] - task_off[k];
    if (nt) {
      const uint32_t rem = (offset[k + 1] - offset[k]) - ((nt - 1) << task_shift);
      atomicAdd(&lh[rem], 1u);
      const uint32_t nfull = nt - 1, head = min(nfull, (uint32_t)ORDER_JMAX);
      for (uint32_t j = 0; j < head; j++) atomicAdd(&jcnt[j], 1u);
      if (nfull > head) atomicAdd(&jcnt[ORDER_JMAX], nfull - head);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {                      // offsets of the depth classes inside this workgroup's range of full tasks
    uint32_t run = 0;
    for (int j = 0; j <= ORDER_JMAX; j++) { joff[j] = run; run += jcnt[j]; }
    const uint32_t base = run ? atomicAdd(&len_cursor[full], run) : 0u;
    for (int j = 0; j <= ORDER_JMAX; j++) joff[j] += base;
  }
  if (threadIdx.x <= MAX_TASK_LEN && threadIdx.x != full) { const uint32_t v = lh[threadIdx.x]; lh[threadIdx.x] = v ? atomicAdd(&len_cursor[threadIdx.x], v) : 0u; }
  __syncthreads();
  if (threadIdx.x == 0 && lh[full]) lh[full] = atomicAdd(&len_cursor[full], lh[full]);     // remainder tasks of full length, after the depth classes
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  for (uint32_t base = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); base < nbuckets; base += step) {     // uniform per wavefront
    const uint32_t k = base + lane;
    uint32_t t = 0, nt = 0, s0 = 0, pos = 0, nfull = 0, head = 0;
    if (k < nbuckets) { t = task_off[k]; nt = task_off[k + 1] - t; }
    if (nt) {
      // the record a task starts from, in execution order: (task, first entry, length, first point reference) -- one coalesced
      // 16-byte load in k_accumulate instead of the chain order -> task -> sorted before the first point can be fetched
      s0 = offset[k];
      const uint32_t rem = (offset[k + 1] - s0) - ((nt - 1) << task_shift);
      nfull = nt - 1;
      head = min(nfull, (uint32_t)ORDER_JMAX);
      for (uint32_t j = 0; j < head; j++) {
        const uint32_t st = s0 + (j << task_shift);
        order[atomicAdd(&joff[j], 1u)] = make_uint4(t + j, st, full, sorted[st]);
      }
      if (nfull > head) pos = atomicAdd(&joff[ORDER_JMAX], nfull - head);
      const uint32_t st = s0 + ((nt - 1) << task_shift);
      // a bucket that is a single task needs no combining: its sum goes straight to the bucket array (bit 31 = "x is the bucket")
      order[atomicAdd(&lh[rem], 1u)] = make_uint4(nt == 1 ? (0x80000000u | k) : t + nt - 1, st, rem, sorted[st]);
    }
    // full tasks beyond depth ORDER_JMAX: a short run is written by its owner, a long one by the wavefront (see k_make_tasks)
    const bool coop = nfull - head > COOP_TASKS;
    if (!coop) {
      for (uint32_t j = head; j < nfull; j++) {
        const uint32_t st = s0 + (j << task_shift);
        order[pos + j - head] = make_uint4(t + j, st, full, sorted[st]);
      }
    }
    unsigned long long big = __ballot(coop);
    while (big) {
      const int src = __ffsll((long long)big) - 1;
      big &= big - 1;
      const uint32_t bt = (uint32_t)__shfl((int)t, src, 64), bs0 = (uint32_t)__shfl((int)s0, src, 64), bpos = (uint32_t)__shfl((int)pos, src, 64),
                     bnfull = (uint32_t)__shfl((int)nfull, src, 64), bhead = (uint32_t)__shfl((int)head, src, 64);
      for (uint32_t j = bhead + lane; j < bnfull; j += 64) {
        const uint32_t st = bs0 + (j << task_shift);
        order[bpos + j - bhead] = make_uint4(bt + j, st, full, sorted[st]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 6. accumulate: one thread per task.  TABLE: the points come from a prepared table (k_build_table), whose coordinates are
// stored reduced, in the internal Montgomery form: the first point of a bucket is then a copy, where a caller's point (radix
// 2^256, unpacked lazily to < 32p) costs two multiplications by one to reduce.  With the ~2 points per bucket and window of
// the wide-window path that is one multiplication in seven.
// ------------------------------------------------------------------------------------------------
template <bool TABLE>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) k_accumulate(const task_t* __restrict__ tasks, const uint32_t* __restrict__ ntasks_p,
                                                    const uint4* __restrict__ order, const uint32_t* __restrict__ sorted,
                                                    const uint32_t* __restrict__ bases, uint32_t* __restrict__ partials,
                                                    uint32_t* __restrict__ buckets, const uint32_t* __restrict__ bases2, uint32_t split) {
  // general path with GLV digits: point references >= split are the endomorphism images, a second array (bases2[ref - split]); the
  // prepared path (TABLE) has one table and ignores both
  auto point = [&](uint32_t idx) -> affine_words {
    if constexpr (!TABLE) { if (idx >= split) return load_affine(bases2, idx - split); }
    return load_affine(bases, idx);
  };
  const uint32_t ntasks = *ntasks_p;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ntasks; i += gridDim.x * blockDim.x) {
    const uint4 rec = order[i];            // tasks run longest-first; partial t stays in bucket order
    const uint32_t t = rec.x;
    struct { uint32_t start, len; } tk = {rec.y, rec.z};
    xyzz acc = xyzz_identity();
    const uint32_t* refs = sorted + tk.start;
    uint32_t ref = rec.w;
    affine_words pt = point(ref & 0x7fffffffu);
    uint32_t j = 0;
    if constexpr (TABLE) {             // the first point is a copy (peeled: merging it with the addition in one loop body costs 30 VGPRs)
      if (!affine_is_identity(pt)) {
        acc.X = fe_unpack<0>(pt.x);     // < p, N form
        acc.Y = fe_unpack<0>(pt.y);
        if (ref >> 31) acc.Y = fe_norm(fe_neg_red(acc.Y, Fq::P2_S1));   // -y: 2p - y
        acc.ZZ = acc.ZZZ = fe_one<Fq>();
      }
      if (tk.len > 1) {
        ref = refs[1];
        pt = point(ref & 0x7fffffffu);
      }
      j = 1;
    }
    // The reference of point j + 1 is loaded during iteration j - 1, so that the gather of point j + 1 (issued at the top of iteration j, under the
    // current addition) does not start with a dependent load: with `ref = refs[j + 1]` inside the iteration every trip began with that load and an
    // s_waitcnt vmcnt(0) before the gather's address existed.
    uint32_t nref = j + 1 < tk.len ? refs[j + 1] : 0u;
    for (; j < tk.len; j++) {
      affine_words cur = pt;
      uint32_t cref = ref;
      if (j + 1 < tk.len) {            // prefetch the next point under the current addition
        ref = nref;
        pt = point(ref & 0x7fffffffu);
      }
      if (j + 2 < tk.len) nref = refs[j + 2];
      if (affine_is_identity(cur)) continue;
      fe x2, y2;
      if constexpr (TABLE) {
        x2 = fe_unpack<0>(cur.x);
        y2 = fe_unpack<0>(cur.y);
        if (cref >> 31) y2 = fe_neg_red(y2, Fq::P2_S1);   // limbs < 2^30, < 2p
      } else {
        x2 = fe_from_ext_lazy(cur.x);   // < 32p
        y2 = fe_from_ext_lazy(cur.y);
        if (cref >> 31) y2 = fe_neg_red(y2, Fq::P64_S1);   // -y: 64p - y < 64p, limbs < 2^30
      }
      xyzz_madd<true>(acc, x2, y2);
    }
    if (t >> 31) store_xyzz(buckets, t & 0x7fffffffu, acc);     // the bucket's only task (k_make_order)
    else store_xyzz(partials, t, acc);
  }
}

// ------------------------------------------------------------------------------------------------
// 7. combine task partials per bucket.  A bucket with m partials needs m - 1 additions; total work is ~1/64 of the
//    accumulation, so what matters is robustness to skew and few launches.  ONE launch:
//      - workgroups [0, blocks): LANES adjacent lanes per bucket sum its m <= seq_parts partials (lane q takes q, q + LANES, ...),
//        log2(LANES) xor-shuffle steps add the lane sums, and the dense bucket array the pyramid reads is written.  The host
//        picks LANES from the expected partials per bucket (1 lane when buckets hold ~1-2 partials: the general path).
//      - the last HEAVY_WGS workgroups walk the list of heavy buckets (m > seq_parts, collected by k_make_tasks: a selector column
//        puts half of all points into the bucket of digit 1), one entry per HEAVY_SLICE partials: the 128 threads sum the slice's
//        partials 128 apart, an LDS tree adds the 128 sums, and for a bucket of several slices the workgroup that finishes last adds the
//        slice sums.  Depth <= 16 + 7 (+ slices / 128 + 7) additions; with uniform scalars the list is empty and these workgroups exit at once.
//        (Until round 2 the heavy buckets were pre-reduced by up to eight radix-4 tree launches over all tasks, each of which cost
//        ~4.7 us even when no bucket needed it.)
// ------------------------------------------------------------------------------------------------
constexpr uint32_t HEAVY_WGS = 256;
constexpr uint32_t HEAVY_QUAD_MAX = 256;     // heavy buckets of at most this many partials are summed by 32 quads instead of 128 lanes

// sum of the 128 threads' points: LDS tree, result in thread 0 (xch: 64 x 36 words)
__device__ __forceinline__ xyzz workgroup_sum_128(xyzz acc, uint32_t* __restrict__ xch) {
#pragma unroll 1
  for (uint32_t half = 64; half >= 1; half >>= 1) {
    __syncthreads();
    if (threadIdx.x >= half && threadIdx.x < 2 * half) store_xyzz(xch, threadIdx.x - half, acc);
    __syncthreads();
    if (threadIdx.x < half) acc = xyzz_add(acc, load_xyzz(xch, threadIdx.x));
  }
  return acc;
}

// One list entry = HEAVY_SLICE partials of one heavy bucket.  A bucket of one slice is finished by its workgroup.  Otherwise the
// slice sum replaces the slice's first partial, and the workgroup that finishes LAST (a per-bucket counter; release / acquire through
// __threadfence, the usual last-block reduction) adds the slice sums: no workgroup ever waits for another.
__device__ __forceinline__ void combine_heavy_buckets(uint32_t wg, const uint32_t* __restrict__ task_off, uint32_t* __restrict__ partials,
                                                      uint32_t* __restrict__ buckets, const uint32_t* __restrict__ heavy_count,
                                                      const uint2* __restrict__ heavy, uint32_t heavy_cap, uint32_t* __restrict__ heavy_done,
                                                      uint32_t* __restrict__ xch /* 64 x 36 words of LDS */, uint32_t quad_max) {
  __shared__ uint32_t is_last;
  const uint32_t nh = min(*heavy_count, heavy_cap);
  for (uint32_t i = wg; i < nh; i += HEAVY_WGS) {          // uniform over the workgroup
    const uint2 e = heavy[i];
    const uint32_t k = e.x, sl = e.y, t = task_off[k], m = task_off[k + 1] - t;
    const uint32_t P = (m + HEAVY_SLICE - 1) / HEAVY_SLICE, lo = sl * HEAVY_SLICE, hi = min(m, lo + HEAVY_SLICE);
    if (m <= quad_max) {
      // a mildly heavy bucket (round 4): up to 256 partials -- e.g. the 2^8 buckets that a column of 88-bit limbs fills in window 4 of the
      // c = 20 layout, ~19 partials each.  The 128-lane slice sum below would spend one single-lane addition plus a 7-level LDS tree of
      // single-lane additions on it (~13 us each at a lone wavefront's issue rate); here the workgroup is 32 quads: quad t adds partials t,
      // t + 32, ... on the quad formulas, a shuffle tree adds the 16 quads of each wavefront, one LDS step the two wavefronts: <= 8 + 4 + 1
      // quad additions of ~5 us (combine 0.118 -> see profiles/r04_heavy_quad_ab.txt on the witness-like mix at 2^20)
      const uint32_t q = threadIdx.x & 3, quad = threadIdx.x >> 2;
      xyzz acc = xyzz_identity();
#pragma unroll 1
      for (uint32_t j = quad; j < m; j += 32) acc = xyzz_add_quad(acc, load_xyzz(partials, t + j), q);
#pragma unroll 1
      for (int mask = 4; mask < 64; mask <<= 1) acc = xyzz_add_quad(acc, xyzz_shfl_xor(acc, mask), q);
      __syncthreads();                                     // (xch may still be read by the previous list entry's tree)
      if (threadIdx.x == 64) store_xyzz(xch, 0, acc);
      __syncthreads();
      if (threadIdx.x < 4) {
        acc = xyzz_add_quad(acc, load_xyzz(xch, 0), q);
        if (q == 0) store_xyzz(buckets, k, acc);
      }
      continue;
    }
    xyzz acc = xyzz_identity();
#pragma unroll 1
    for (uint32_t j = lo + threadIdx.x; j < hi; j += 128) acc = xyzz_add(acc, load_xyzz(partials, t + j));
    acc = workgroup_sum_128(acc, xch);
    if (P == 1) {
      if (threadIdx.x == 0) store_xyzz(buckets, k, acc);
      continue;
    }
    if (threadIdx.x == 0) {
      store_xyzz(partials, t + lo, acc);
      __threadfence();                                     // the slice sum is visible device-wide before the count moves
      is_last = atomicAdd(&heavy_done[i - sl], 1u) == P - 1;
    }
    __syncthreads();
    if (!is_last) continue;                                // (uniform: is_last is shared)
    __threadfence();                                       // the other slices' sums, written by other workgroups
    acc = xyzz_identity();
#pragma unroll 1
    for (uint32_t j = threadIdx.x; j < P; j += 128) acc = xyzz_add(acc, load_xyzz(partials, t + j * HEAVY_SLICE));
    acc = workgroup_sum_128(acc, xch);
    if (threadIdx.x == 0) store_xyzz(buckets, k, acc);
  }
}

template <int COMBINE_LANES>
__global__ void __launch_bounds__(128) k_combine_seq(const uint32_t* __restrict__ task_off, uint32_t nbuckets,
                                                     uint32_t* __restrict__ partials, uint32_t* __restrict__ buckets,
                                                     uint32_t seq_parts, uint32_t blocks, const uint32_t* __restrict__ heavy_count,
                                                     const uint2* __restrict__ heavy, uint32_t heavy_cap, uint32_t* __restrict__ heavy_done, uint32_t quad_max) {
  __shared__ __attribute__((aligned(16))) uint32_t xch[64 * 36];
  if (blockIdx.x >= blocks) { combine_heavy_buckets(blockIdx.x - blocks, task_off, partials, buckets, heavy_count, heavy, heavy_cap, heavy_done, xch, quad_max); return; }
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t k = gid / COMBINE_LANES, q = gid % COMBINE_LANES;
  const bool live = k < nbuckets;                  // whole lane groups are live or dead together (128 % 8 == 0)
  xyzz acc = xyzz_identity();
  uint32_t m = 0;
  if (live) {
    const uint32_t t = task_off[k];
    m = task_off[k + 1] - t;
    const uint32_t left = (m == 1 || m > seq_parts) ? 0u : m;   // a single task wrote its bucket itself; a heavy bucket has its own workgroup
#pragma unroll 1
    for (uint32_t j = q; j < left; j += COMBINE_LANES) acc = xyzz_add(acc, load_xyzz(partials, t + j));
  }
#pragma unroll 1
  for (int mask = 1; mask < COMBINE_LANES; mask <<= 1) acc = xyzz_add(acc, xyzz_shfl_xor(acc, mask));
  if (live && q == 0 && m != 1 && m <= seq_parts) store_xyzz(buckets, k, acc);
}

// the same with a quad per lane of the above (small MSMs: the bucket count is far below the chip's lane count and the step is
// the latency of its additions)
template <int COMBINE_LANES>
__global__ void __launch_bounds__(128) k_combine_seq_quad(const uint32_t* __restrict__ task_off, uint32_t nbuckets,
                                                          uint32_t* __restrict__ partials, uint32_t* __restrict__ buckets,
                                                          uint32_t seq_parts, uint32_t blocks, const uint32_t* __restrict__ heavy_count,
                                                          const uint2* __restrict__ heavy, uint32_t heavy_cap, uint32_t* __restrict__ heavy_done, uint32_t quad_max) {
  __shared__ __attribute__((aligned(16))) uint32_t xch[64 * 36];
  if (blockIdx.x >= blocks) { combine_heavy_buckets(blockIdx.x - blocks, task_off, partials, buckets, heavy_count, heavy, heavy_cap, heavy_done, xch, quad_max); return; }
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t q = gid & 3, lane = (gid >> 2) % COMBINE_LANES, k = (gid >> 2) / COMBINE_LANES;
  const bool live = k < nbuckets;                  // whole groups of 4 * COMBINE_LANES lanes are live or dead together
  xyzz acc = xyzz_identity();
  uint32_t m = 0;
  if (live) {
    const uint32_t t = task_off[k];
    m = task_off[k + 1] - t;
    const uint32_t left = (m == 1 || m > seq_parts) ? 0u : m;
#pragma unroll 1
    for (uint32_t j = lane; j < left; j += COMBINE_LANES) acc = xyzz_add_quad(acc, load_xyzz(partials, t + j), q);
  }
#pragma unroll 1
  for (int mask = 1; mask < COMBINE_LANES; mask <<= 1) acc = xyzz_add_quad(acc, xyzz_shfl_xor(acc, 4 * mask), q);
  if (live && lane == 0 && q == 0 && m != 1 && m <= seq_parts) store_xyzz(buckets, k, acc);
}

// ------------------------------------------------------------------------------------------------
// 8. pyramid reduction.  Per window, state at the start of step s (1-based):
//      X   : N elements            (N = B >> (s-1))
//      Z^l : N/2 elements each, l = 0 .. s-2
//    laid out contiguously [X | Z^0 | ... | Z^(s-2)], window stride = `in_stride` elements.
//    Step s writes [X' (N/2) | Z'^0 .. Z'^(s-1) (N/4 each)]:
//      X'[t]        = X[2t] + X[2t+1]
//      Z'^l[u]      = Z^l[2u] + Z^l[2u+1]          (l <= s-2)
//      Z'^(s-1)[u]  = X[4u+1] + X[4u+3]            (odd elements of X start their own tree)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(128) k_pyramid_step(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                      uint32_t N, int s, uint32_t in_stride, uint32_t out_stride) {
  const uint32_t per_win = N / 2 + (uint32_t)s * (N / 4);
  const int win = blockIdx.y;
  const uint32_t* wi = in + (size_t)win * in_stride * 36;
  uint32_t* wo = out + (size_t)win * out_stride * 36;
  // grid stride: the host may launch fewer threads than additions (ZKHIP_PYR_PERSIST: one resident round of waves that loop, instead of
  // several rounds of one-addition waves that all start with their loads at the same moment)
#pragma unroll 1
  for (uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x; tid < per_win; tid += gridDim.x * blockDim.x) {
    uint32_t ia, ib;
    if (tid < N / 2) {
      ia = 2 * tid; ib = 2 * tid + 1;
    } else {
      uint32_t r = tid - N / 2;
      uint32_t l = r / (N / 4), u = r % (N / 4);
      if ((int)l == s - 1) { ia = 4 * u + 1; ib = 4 * u + 3; }
      else { ia = N + l * (N / 2) + 2 * u; ib = ia + 1; }
    }
    store_xyzz(wo, tid, xyzz_add(load_xyzz(wi, ia), load_xyzz(wi, ib)));
  }
}

// the same step with the 4 lanes of a quad per addition (ec_quad.hpp): every pyramid level is far smaller than the chip, so the
// step time is the latency of one addition -- 4 stages of one multiplication instead of 14 dependent ones
__global__ void __launch_bounds__(128) k_pyramid_step_quad(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                           uint32_t N, int s, uint32_t in_stride, uint32_t out_stride) {
  const uint32_t per_win = N / 2 + (uint32_t)s * (N / 4);
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t tid = gid >> 2, q = gid & 3;
  if (tid >= per_win) return;                          // whole quads leave together
  const int win = blockIdx.y;
  const uint32_t* wi = in + (size_t)win * in_stride * 36;
  uint32_t* wo = out + (size_t)win * out_stride * 36;
  uint32_t ia, ib;
  if (tid < N / 2) {
    ia = 2 * tid; ib = 2 * tid + 1;
  } else {
    uint32_t r = tid - N / 2;
    uint32_t l = r / (N / 4), u = r % (N / 4);
    if ((int)l == s - 1) { ia = 4 * u + 1; ib = 4 * u + 3; }
    else { ia = N + l * (N / 2) + 2 * u; ib = ia + 1; }
  }
  const xyzz r = xyzz_add_quad(load_xyzz(wi, ia), load_xyzz(wi, ib), q);
  if (q == 0) store_xyzz(wo, tid, r);
}

// 9a. per-window weighted sum.  Input state after the last pyramid step: X has 2 elements, Z^0..Z^(nz-1) one each.
//     window sum = X0 + X1 + sum_l 2^l Z^l + 2^nz X1.  One 32-lane group per window: lane l < nz computes 2^l Z^l by l
//     doublings, lane nz computes 2^nz X1, lane nz + 1 holds X0 + X1; a 4-step shuffle tree adds the terms.
//     Depth nz doublings + 5 additions instead of nz (doubling + addition).  Needs nz + 2 <= 32 (c <= 21: nz = c - 2).
__global__ void __launch_bounds__(64) k_window_horner(const uint32_t* __restrict__ in, uint32_t in_stride, int nz,
                                                      uint32_t* __restrict__ winsum, int W) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int win = gid >> 5, lane = gid & 31;
  xyzz acc = xyzz_identity();
  if (win < W) {
    const uint32_t* wi = in + (size_t)win * in_stride * 36;
    int dbl = 0;
    if (lane < nz) { acc = load_xyzz(wi, 2 + lane); dbl = lane; }
    else if (lane == nz) { acc = load_xyzz(wi, 1); dbl = nz; }
    else if (lane == nz + 1) acc = xyzz_add(load_xyzz(wi, 0), load_xyzz(wi, 1));
#pragma unroll 1
    for (int i = 0; i < dbl; i++) acc = xyzz_dbl(acc);
  }
#pragma unroll 1
  for (int mask = 1; mask < 32; mask <<= 1) acc = xyzz_add(acc, xyzz_shfl_xor(acc, mask));
  if (win < W && lane == 0) store_xyzz(winsum, win, acc);
}

// The same with one quad per term: a 128-thread workgroup per window, term t = threadIdx / 4 (t < 32), doubling chains and the
// 5-step addition tree on the quad formulas; the two wavefronts meet through LDS for the last addition.
__global__ void __launch_bounds__(128) k_window_horner_quad(const uint32_t* __restrict__ in, uint32_t in_stride, int nz,
                                                            uint32_t* __restrict__ winsum, uint32_t* __restrict__ jac_out) {
  __shared__ __attribute__((aligned(16))) uint32_t xch[36];
  const int win = blockIdx.x, term = threadIdx.x >> 2;
  const uint32_t q = threadIdx.x & 3;
  const uint32_t* wi = in + (size_t)win * in_stride * 36;
  xyzz acc = xyzz_identity();
  int dbl = 0;
  if (term < nz) { acc = load_xyzz(wi, 2 + term); dbl = term; }
  else if (term == nz) { acc = load_xyzz(wi, 1); dbl = nz; }
  else if (term == nz + 1) acc = xyzz_add_quad(load_xyzz(wi, 0), load_xyzz(wi, 1), q);
#pragma unroll 1
  for (int i = 0; i < dbl; i++) acc = xyzz_dbl_quad(acc, q);
#pragma unroll 1
  for (int mask = 4; mask < 64; mask <<= 1) acc = xyzz_add_quad(acc, xyzz_shfl_xor(acc, mask), q);   // quads 4 lanes apart
  if (threadIdx.x == 64) store_xyzz(xch, 0, acc);       // sum of terms 16..31
  __syncthreads();
  if (threadIdx.x < 4) {
    acc = xyzz_add_quad(acc, load_xyzz(xch, 0), q);
    if (q == 0) {
      if (jac_out) store_jacobian(acc, jac_out + (size_t)win * 24);     // prepared path: a bucket set's weighted sum is a final result
      else store_xyzz(winsum, win, acc);
    }
  }
}

// 9b. fold windows: result = sum_w 2^(c w) winsum[w]; writes the Jacobian result (24 words).  One quad: c (W - 1) dependent
// doublings are the whole cost (the general path only; prepared bases need no fold).
__global__ void __launch_bounds__(64) k_fold(const uint32_t* __restrict__ winsum, int W, int c, uint32_t* __restrict__ out) {
  if (threadIdx.x >= 4 || blockIdx.x != 0) return;
  const uint32_t q = threadIdx.x;
  xyzz acc = load_xyzz(winsum, W - 1);
  for (int w = W - 2; w >= 0; w--) {
#pragma unroll 1
    for (int i = 0; i < c; i++) acc = xyzz_dbl_quad(acc, q);
    acc = xyzz_add_quad(acc, load_xyzz(winsum, w), q);
  }
  if (q == 0) store_jacobian(acc, out);
}

// batched prepared MSM: every bucket set's weighted sum is a final result
__global__ void __launch_bounds__(64) k_store_results(const uint32_t* __restrict__ winsum, uint32_t K, uint32_t* __restrict__ out) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K) store_jacobian(load_xyzz(winsum, k), out + (size_t)k * 24);
}

// sum of `m` Jacobian points (multi-GPU partial fold): out = sum in[i].  One wave of 16 quads on the quad formulas, then a
// 4-step shuffle tree (8 GPUs: depth 3 additions instead of 8 sequential ones).
// Workgroup b: out[b] = sum over i < m of in[i * stride + b] (96-byte Jacobian points).  One workgroup with stride 1 is the plain fold of
// m gathered partial sums; `count` workgroups with stride = count fold the [shard][vector] partials of a batched sharded commit.
__global__ void __launch_bounds__(64) k_sum_jacobian(const uint32_t* __restrict__ in, int m, uint32_t* __restrict__ out, uint32_t stride) {
  const uint32_t q = threadIdx.x & 3;
  const int grp = threadIdx.x >> 2;                 // 16 quads: quad g sums points g, g + 16, ...
  in += (size_t)blockIdx.x * 24;
  out += (size_t)blockIdx.x * 24;
  xyzz acc = xyzz_identity();
  for (int i = grp; i < m; i += 16) acc = xyzz_add_quad(acc, load_jacobian(in + (size_t)i * stride * 24), q);
#pragma unroll 1
  for (int mask = 4; mask < 64; mask <<= 1) {
    if ((mask >> 2) >= m && mask > 4) break;        // quads >= m hold the identity: higher steps add nothing
    acc = xyzz_add_quad(acc, xyzz_shfl_xor(acc, mask), q);
  }
  if (threadIdx.x == 0) store_jacobian(acc, out);
}

// ------------------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------------------
// A scalar has 254 bits, so the top window holds 254 - (W - 1) c of them.  With 1..5 bits there every point lands in one of at most 16
// buckets of that window: a few enormous buckets whose partial sums are combined as long chains (measured, general path at 2^18:
// c = 14 -> 2 top bits 2.29 ms, c = 13 -> 7 bits 1.59 ms, c = 15 -> carries only 1.58 ms).  Such window sizes are never chosen.
static bool msm_top_window_is_degenerate(int c) {
  const int W = (256 + c - 1) / c, top_bits = 254 - (W - 1) * c;
  return top_bits > 0 && top_bits < 6;
}

// general path: GLV halves the scalars (k_digits_glv): 2n points, 127-bit magnitudes, W = ceil(128 / c) windows.  ZKHIP_NO_GLV=1: the
// plain 254-bit digits (A/B knob).
static bool msm_glv_enabled() {
  static const bool on = getenv("ZKHIP_NO_GLV") == nullptr;
  return on;
}
// windows of c bits: ceil(256 / c) for 254-bit scalars (the signed recoding's last carry lands in the spare top bits); for the GLV halves
// (magnitudes below 0.979 * 2^127) ceil(128 / c), plus one when the top window would otherwise overflow: with c W = 128 the top window
// holds up to 0.49 * 2^c and a carry of 1 must still leave it below 2^(c-1) -- true for c = 8 (125 + 1 < 128) and c = 16, not for c = 2
// and c = 4; every other window size has spare bits above bit 127
static inline int msm_windows(int c, bool glv) {
  if (!glv) return (256 + c - 1) / c;
  return (128 + c - 1) / c + ((c == 2 || c == 4) ? 1 : 0);
}

int msm_pick_window(size_t n) {
  // cost model in field multiplications: W * (10 n' + 2 * 14 * 2^(c-1)); c <= 16 (int16 digits, 128 KiB LDS).  Without GLV n' = n points
  // and W = ceil(256 / c); with it n' = 2 n and W = ceil(128 / c).  Window sizes whose top window would hold 1..5 bits are skipped.
  const bool glv = msm_glv_enabled();
  int best = 16;
  double best_cost = 1e300;
  for (int c = 2; c <= 16; c++) {
    const int Wi = msm_windows(c, glv), top_bits = (glv ? 127 : 254) - (Wi - 1) * c;
    if (top_bits > 0 && top_bits < 6) continue;
    const double cost = (double)Wi * (10.0 * (double)(glv ? 2 * n : n) + 28.0 * (double)(1u << (c - 1)));
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}


constexpr int MAX_WINDOW_PREPARED = 20;   // 2^19 buckets = 128 groups of 2^12

int msm_pick_window_prepared(size_t n) {
  // one shared bucket set: W * 10 n multiplications of accumulation + the bucket reduction.
  // Below 2^20 points (windows <= 16 bits): the reduction priced at 2 * 14 * 2^(c-1) multiplications; the choice was measured end to end under every
  // window size (tools/small_window_probe.py) and is the fastest or within 1 % of it, with the one exception handled below.
  // From 2^20 points (round 4): the reduction tail priced as MEASURED (tools/window_sweep.py, profiles/r04_window_sweep.txt; same box, alternating,
  // c = 16 / 18 / 19 / 20 at 2^20 .. 2^23), in units of one field multiplication of the accumulation kernel (6.1 ps: 61 ps per mixed addition):
  //   pyramid  = 0.24 ns per bucket (39 units; the single-lane levels) + 6.7 us per level (1.1 M units; c - 2 dependent quad steps)
  //              measured 0.094 / 0.141 / 0.173 / 0.247 ms at c = 16 / 18 / 19 / 20, the fit gives 0.094 / 0.139 / 0.177 / 0.247
  //   weighted sum + combine ~ 0.08 ms (13 M units), nearly independent of c
  //   sort     ~ 1 unit per entry (W n entries)
  //   accumulation per addition relative to c = 20: 1.07 (c = 16), 1.06 (c = 18), 1.13 (c = 19) -- the balanced layout of W = 14 / 15 windows
  //              fills the lower half of the bucket range with all windows and the upper half with 4 / 1 of them (96 against 16 entries per
  //              bucket at 2^20), which costs the task list its uniformity; measured 64.9 / 64.3 / 68.4 / 60.6 ps per addition at 2^20
  // The tail's latency part is why wider windows lose below 2^20 points and why c = 19 never wins on uniform scalars (2^20: 1.521 vs 1.385 ms,
  // 2^23: 9.78 vs 9.08): halving the buckets returns 0.07 ms of pyramid and costs 7.7 % + 13 % of the accumulation.
  const bool large = n >= ((size_t)1 << 20);
  int best = 2;
  double best_cost = 1e300;
  for (int c = 2; c <= MAX_WINDOW_PREPARED; c++) {
    const double W = (256 + c - 1) / c, buckets = (double)(1u << (c - 1));
    if (msm_top_window_is_degenerate(c) && c <= 16) continue;     // (windows above 16 bits are balanced: no short top window)
    if (c > 16 && !large) continue;   // measured (median of 60, c = 16 / c = 20): 2^20 1.83 / 1.67 ms, 2^21 3.32 / 2.95 on one box; the crossover was at 2^22 before the wide sort wrote bucket runs
    if (c == 17) continue;            // 16 balanced windows of 16 bits in a 2^16-bucket range: c = 16 with half the buckets empty
    double cost;
    if (!large) cost = W * 10.0 * (double)n + 28.0 * buckets;
    else {
      const double eff = c == 20 ? 1.0 : (c == 19 ? 1.13 : (c == 18 ? 1.06 : 1.07));
      cost = W * (10.0 * eff + 1.0) * (double)n + 39.0 * buckets + 1.1e6 * (double)(c - 2) + 13.0e6;
    }
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  // small MSMs are latency-bound (their cost is the depth of the reduction tail and the task chains, which the multiplication count above
  // does not see): measured end to end under every window size (tools/small_window_probe.py), the model's choice is the fastest or within
  // 1 % of it except at 2^14 .. 2^15 points, where 15 bits beat its 13 / 16 (2^14: 0.257 vs 0.270 ms, 2^15: 0.310 vs 0.336 ms)
  if (n >= ((size_t)1 << 14) && n < ((size_t)3 << 14)) best = 15;
  if (const char* e = getenv("ZKHIP_MAX_WINDOW")) { int m = atoi(e); if (m >= 2 && best > m) best = m; }   // A/B knob
  return best;
}

// Workspace layout.  ONE routine lays the regions out, and it serves both the size query (base == nullptr) and the launcher, so
// the two can never disagree (they once did: the estimator padded n before applying the task-count switch below and the
// launcher did not, which undersized the workspace for n just below 2^19).
struct msm_layout {
  uint32_t task_shift;          // log2 of the task length
  size_t max_tasks;             // upper bound of the task count for that length
  void* digits;
  uint32_t* stage_ref; uint16_t* stage_fine;
  char* zero_lo; size_t zero_bytes;   // everything that starts at zero sits together: one fill instead of three
  uint32_t *gcounters, *counters, *count, *sorted, *offset, *cursor, *task_off, *bsum1, *bsum2;
  task_t* tasks; uint4* order;
  uint32_t *partials, *pyrA, *pyrB, *winsum;
  uint32_t* endo;               // GLV (general path): phi(bases), n points
  int combine_lanes;            // lanes that sum one bucket's partials in the combine step
  uint32_t seq_parts;           // partials per bucket that step sums; buckets with more are "heavy"
  uint2* heavy; uint32_t heavy_cap;      // list of (heavy bucket, slice) entries (count: counters[3])
  uint32_t* heavy_done;                  // per list index of a bucket's first entry: slices finished
  size_t total;                 // bytes from the base to the end of the last region
};

// Task length.  With >= 2^17 tasks of 64 entries the chip is full and 64 is best (throughput-bound, see TASK_SHIFT).  Below
// that the MSM is latency-bound: a task of L entries is L sequential mixed adds (~5 us each at low occupancy) and a bucket
// of s entries leaves s/L partials to sum (~10 us each): pick L = 2^shift minimising 5 L + 10 (s/L - 1).
static uint32_t msm_task_shift(size_t entries /* W n K */, size_t NB, size_t xyzz_bytes = 144) {
  uint32_t task_shift = TASK_SHIFT;
  // 128-entry tasks halve the partials the combine step has to add, and pay once the launch is many rounds of waves deep even
  // so: measured (6 / 7) 2^24 21.40 / 21.00 ms (combine 0.71 -> 0.27), but 2^22 6.00 / 6.07 and 2^20 1.72 / 1.85 (too few tasks).
  // Round 3 (W = 13): 2^22 4.78 / 4.89, 2^23 9.24 / 9.02, 2^24 17.99 / 17.70 ms -- the switch sits between 2^22 and 2^23 points.
  static const int knob = [] { const char* e = getenv("ZKHIP_TASK_SHIFT"); return e ? atoi(e) : 0; }();
  if (knob >= 2 && knob <= 7) task_shift = (uint32_t)knob;
  else if ((entries >> 7) >= ((size_t)3 << 16)) task_shift = 7;      // (with the balanced windows: 2^21 2.397 / 2.380 ms, 2^22 4.812 / 4.815 -- from 2^21 points on)
  // G2 (xyzz_bytes 288): a general addition costs ~1.4 mixed ones in instructions but runs in the combine step's thinly occupied lanes, and the
  // accumulation holds 2 waves per SIMD: the chip is full from 2^16 tasks, and a partial is priced at 3 mixed additions
  const bool g2 = xyzz_bytes == 288;
  if ((entries >> TASK_SHIFT) < ((size_t)1 << (g2 ? 14 : 17))) {
    const double occ = (double)entries / (double)NB;
    double best = 1e300;
    for (uint32_t sh = 2; sh <= (uint32_t)TASK_SHIFT; sh++) {
      const double L = (double)(1u << sh), parts = occ / L;
      const double est = 5.0 * L + (g2 ? 15.0 : 10.0) * (parts > 1.0 ? parts - 1.0 : 0.0);
      if (est < best) { best = est; task_shift = sh; }
    }
  }
  return task_shift;
}

// The two-level sort (coarse groups of 4096 buckets, then bucket runs sorted inside LDS) serves the wide windows of the prepared path and, since
// round 3, the general path's GLV MSMs at c = 16: there a window has its own 2^15 buckets = 8 groups, 8 windows = 64 groups.  The one-level
// scatter it replaces stores every reference with its own 4-byte write (PMC: 8.7 bytes reach memory per byte of payload).
// The same holds for prepared tables with 16-bit windows (2^16 <= n < 2^20: one shared set of 2^15 buckets = 8 groups).
static bool msm_two_level(int c, bool glv, size_t n_in, bool shared_buckets, size_t K) {
  if (c > 16) return true;
  static const long thr_glv = [] { const char* e = getenv("ZKHIP_GLV_TWO_LEVEL_MIN"); return e ? atol(e) : (1L << 18); }();      // A/B knobs (0 = never)
  static const long thr_prep = [] { const char* e = getenv("ZKHIP_PREP_TWO_LEVEL_MIN"); return e ? atol(e) : (1L << 16); }();
  if (c != 16 || K != 1) return false;
  if (glv) return thr_glv > 0 && n_in >= (size_t)thr_glv;
  return shared_buckets && thr_prep > 0 && n_in >= (size_t)thr_prep;
}

// n_in: scalars per vector.  glv: the sort sees 2 n_in points (a point and its endomorphism image) and ceil(128 / c) windows.
static msm_layout msm_lay_out(char* base, size_t n_in, size_t K, int c, bool prepared, size_t xyzz_bytes = 144, bool glv = false) {
  msm_layout L;
  const size_t n = glv ? 2 * n_in : n_in;
  const size_t W = (size_t)msm_windows(c, glv), B = (size_t)1 << (c - 1), WB = prepared ? K : W, NB = WB * B;
  const bool wide = msm_two_level(c, glv, n_in, prepared, K);
  const size_t nk = n * K, entries = W * nk;
  const size_t n_pad = (n + 7) & ~(size_t)7;
  L.task_shift = msm_task_shift(entries, NB, xyzz_bytes);
  L.max_tasks = (entries >> L.task_shift) + NB + 1;        // every bucket adds at most one task that is not full
  // expected partials per bucket decide how many lanes sum a bucket in the combine step; a bucket with more than 16 partials per lane
  // is handled by a workgroup of its own, so a skewed bucket (a selector column, a short top window) never becomes a long chain
  const double parts_avg = (double)entries / (double)NB / (double)(1u << L.task_shift);
  L.combine_lanes = parts_avg <= 2.0 ? 1 : (parts_avg <= 4.0 ? 2 : (parts_avg <= 12.0 ? 4 : 8));
  L.seq_parts = 16u * (uint32_t)L.combine_lanes;
  L.heavy_cap = (uint32_t)(std::min<size_t>(L.max_tasks / (L.seq_parts + 1) + 1, NB) + L.max_tasks / HEAVY_SLICE + 1);   // heavy buckets + full slices
  char* p = base;
  auto carve = [&](size_t bytes) { void* r = p; p += align_up(bytes, 256); return r; };
  L.digits = carve(W * K * n_pad * (wide ? sizeof(int32_t) : sizeof(int16_t)));
  L.stage_ref = wide ? (uint32_t*)carve(entries * sizeof(uint32_t)) : nullptr;
  L.stage_fine = wide ? (uint16_t*)carve(entries * sizeof(uint16_t)) : nullptr;
  L.zero_lo = p;
  L.gcounters = wide ? (uint32_t*)carve(32768) : nullptr;  // words: [128..257) group offsets, [512..641) chunk starts, [1024..3072) counts [window][group], [3072..5120) cursors [window][group]
  L.counters = (uint32_t*)carve(2048);                     // [0] total entries, [1] total tasks, [2] max task partials of one bucket, [3] heavy buckets,
                                                           // [32..161) task-length histogram (MAX_TASK_LEN + 1), [192..321) its cursors
  L.count = (uint32_t*)carve((NB + 1) * sizeof(uint32_t));
  L.heavy_done = (uint32_t*)carve((size_t)L.heavy_cap * sizeof(uint32_t));
  L.zero_bytes = (size_t)(p - L.zero_lo);
  L.sorted = (uint32_t*)carve(entries * sizeof(uint32_t));
  L.offset = (uint32_t*)carve((NB + 1) * sizeof(uint32_t));
  L.cursor = (uint32_t*)carve((NB + 1) * sizeof(uint32_t));
  L.task_off = (uint32_t*)carve((NB + 1) * sizeof(uint32_t));
  L.heavy = (uint2*)carve((size_t)L.heavy_cap * sizeof(uint2));
  L.bsum1 = (uint32_t*)carve((NB / SCAN_TILE + 2) * sizeof(uint32_t));
  L.bsum2 = (uint32_t*)carve((NB / SCAN_TILE + 2) * sizeof(uint32_t));
  L.tasks = (task_t*)carve(L.max_tasks * sizeof(task_t));
  L.order = (uint4*)carve(L.max_tasks * sizeof(uint4));    // execution order: one record per task
  L.partials = (uint32_t*)carve(L.max_tasks * xyzz_bytes);
  L.pyrA = (uint32_t*)carve(WB * B * xyzz_bytes);          // pyramid ping-pong: per bucket set N + (s-1) N/2 <= B elements at every step
  L.pyrB = (uint32_t*)carve(WB * B * xyzz_bytes);
  L.winsum = (uint32_t*)carve((W + K) * xyzz_bytes);       // window / bucket-set sums
  L.endo = glv ? (uint32_t*)carve(n_in * (xyzz_bytes == 288 ? 128 : 64)) : nullptr;     // phi(bases): G1Affine 64 B, G2Affine 128 B
  L.total = (size_t)(p - base);
  return L;
}

// GLV applies to the general path of both curves (xyzz_bytes 144 = G1, 288 = G2; round 5: the twist has the same endomorphism with beta^2):
// arbitrary bases, one scalar vector (the prepared path has its doublings in the table already)
bool msm_uses_glv(bool prepared, size_t batch, size_t xyzz_bytes) { return !prepared && batch == 1 && (xyzz_bytes == 144 || xyzz_bytes == 288) && msm_glv_enabled(); }
int msm_glv_windows(int c) { return msm_windows(c, true); }

// Direct tables (below: "Direct tables") serve single MSMs over prepared sets of at most 2^ZKHIP_DIRECT_MAX_LOG points (default 15; 0 = never; at most 18)
int msm_direct_max_log() {
  static const int v = [] { const char* e = getenv("ZKHIP_DIRECT_MAX_LOG"); const int x = e ? atoi(e) : 15; return (x >= 0 && x <= 18) ? x : 15; }();
  return v;
}
static inline bool msm_direct_wanted(size_t n) { const int L = msm_direct_max_log(); return L > 0 && n >= 2 && n <= ((size_t)1 << L); }

size_t msm_workspace_bytes(size_t n_one, int c, bool prepared, size_t batch, size_t xyzz_bytes) {
  if (n_one == 0 || batch == 0) return 0;
  size_t b = msm_lay_out(nullptr, n_one, batch, c, prepared, xyzz_bytes, msm_uses_glv(prepared, batch, xyzz_bytes)).total;
  if (prepared && batch == 1 && xyzz_bytes == 144 && msm_direct_wanted(n_one)) b = std::max(b, msm_direct_workspace_bytes(n_one));   // the same call may take the direct path
  return b;
}

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s failed: %s", #x, hipGetErrorString(e_)); return ZKHIP_EHIP; } } while (0)

// Steps 1 - 5 (digits, bucket sort, tasks, execution order): everything that depends on the scalars only, shared by the curves.
// `shared_buckets`: one bucket set per MSM of the batch for all windows (prepared G1 tables) instead of one per window.
// ref_base / ref_stride: point reference of entry i of window w = ref_base + w * ref_stride + i.
int msm_build_tasks(const uint32_t* d_scalars, size_t n_in, size_t batch, size_t scalar_stride, int c, bool shared_buckets, uint32_t ref_base, uint32_t ref_stride,
                    size_t xyzz_bytes, void* ws, size_t ws_bytes, hipStream_t stream, msm_tasks_view* out, bool glv) {
  const uint32_t K = (uint32_t)batch;
  const bool wide = msm_two_level(c, glv, n_in, shared_buckets, batch);     // two-level bucket sort, int32 digits
  const int W = msm_windows(c, glv);
  if (glv && (K != 1 || c > 16 || shared_buckets)) { set_error("msm: GLV digits are for one vector of the general path"); return ZKHIP_EINVAL; }
  const size_t n = glv ? 2 * n_in : n_in;           // points the sort sees
  const uint32_t B = 1u << (c - 1);
  const int WB = shared_buckets ? (int)K : W;      // number of bucket sets
  const uint32_t NB = (uint32_t)WB * B;
  if ((size_t)W * n * K >= (1ull << 32) || (size_t)WB * B >= (1ull << 31)) { set_error("msm: W*n*batch overflows 32-bit slot index"); return ZKHIP_EINVAL; }
  if (c > 16 && (!shared_buckets || K != 1)) { set_error("msm: windows above 16 bits need one shared bucket set"); return ZKHIP_EINVAL; }
  if (wide && !shared_buckets && (size_t)W * (B >> FINE_BITS) > (size_t)MAX_GROUPS) { set_error("msm: two-level sort: %d windows x %u groups exceed %d", W, B >> FINE_BITS, MAX_GROUPS); return ZKHIP_EINVAL; }
  if (wide && W > MAX_WIDE_W) { set_error("msm: %d windows of %d bits exceed the wide path's %d", W, c, MAX_WIDE_W); return ZKHIP_EINVAL; }
  const msm_layout lay = msm_lay_out((char*)ws, n_in, K, c, shared_buckets, xyzz_bytes, glv);
  if (ws_bytes < lay.total) { set_error("msm: workspace too small (%zu < %zu bytes)", ws_bytes, lay.total); return ZKHIP_EINVAL; }
  const uint32_t task_shift = lay.task_shift;
  const uint32_t n_pad = (uint32_t)((n + 7) & ~(size_t)7);
  void* const digits = lay.digits;
  uint32_t* const stage_ref = lay.stage_ref;
  uint16_t* const stage_fine = lay.stage_fine;
  uint32_t *const gcounters = lay.gcounters, *const counters = lay.counters, *const count = lay.count, *const sorted = lay.sorted,
           *const offset = lay.offset, *const cursor = lay.cursor, *const task_off = lay.task_off, *const bsum1 = lay.bsum1, *const bsum2 = lay.bsum2;
  task_t* const tasks = lay.tasks;
  uint4* const order = lay.order;

  prof_begin(stream);
  HIPCHK(hipMemsetAsync(lay.zero_lo, 0, lay.zero_bytes, stream));
  // 1. digits
  const size_t sstride = K > 1 ? scalar_stride : n;
  const size_t nk = n * K, max_tasks = lay.max_tasks;
  (void)nk; (void)max_tasks;
  const unsigned dblocks = (unsigned)(((size_t)K * n_pad + 255) / 256);
  const int narrow = (shared_buckets && !glv) ? msm_narrow_windows(c, W) : 0;       // balanced windows of the wide prepared tables (k_build_table agrees)
  if (glv && wide) hipLaunchKernelGGL(k_digits_glv<int32_t>, dim3((unsigned)((std::max<size_t>(n_in, 8) + 255) / 256)), dim3(256), 0, stream, d_scalars, (int32_t*)digits, (uint32_t)n_in, n_pad, c, W);
  else if (glv) hipLaunchKernelGGL(k_digits_glv<int16_t>, dim3((unsigned)((std::max<size_t>(n_in, 8) + 255) / 256)), dim3(256), 0, stream, d_scalars, (int16_t*)digits, (uint32_t)n_in, n_pad, c, W);
  else if (wide) {
    static const bool fused_count = getenv("ZKHIP_NO_FUSED_COUNT") == nullptr;      // A/B knob
    if (fused_count) {
      // workgroups: 512 up to 2^21 scalars, 1024 above (measured 256 / 512 / 1024 at 2^20: digits 0.041 / 0.036 / 0.040 ms; 256 / 1024 / 4096 at
      // 2^22: 0.129 / 0.110 / 0.112 -- fewer workgroups leave a lane several scalars with exposed loads, more multiply the contended atomics)
      static const unsigned wg_knob = [] { const char* e = getenv("ZKHIP_DIGIT_WGS"); const int v = e ? atoi(e) : 0; return v >= 64 && v <= 4096 ? (unsigned)v : 0u; }();   // A/B knob
      const unsigned wg_cap = wg_knob ? wg_knob : (n_pad > (1u << 21) ? 1024u : 512u);
      const unsigned wblocks = (unsigned)std::min<size_t>(((size_t)n_pad + 1023) / 1024, wg_cap);
      hipLaunchKernelGGL(k_digits_wide, dim3(wblocks), dim3(1024), 0, stream, d_scalars, (int32_t*)digits, (uint32_t)n, n_pad, c, W, gcounters + 1024, narrow);
    } else {
      hipLaunchKernelGGL(k_digits<int32_t>, dim3(dblocks), dim3(256), 0, stream, d_scalars, (int32_t*)digits, (uint32_t)n, n_pad, c, W, K, sstride, narrow);
    }
  }
  else hipLaunchKernelGGL(k_digits<int16_t>, dim3(dblocks), dim3(256), 0, stream, d_scalars, (int16_t*)digits, (uint32_t)n, n_pad, c, W, K, sstride, 0);
  prof_mark(stream, "digits");
  // 2. count
  uint32_t chunks = (uint32_t)((n + 65535) / 65536);
  while (chunks * (uint32_t)W * K < 256 && chunks < (n + 4095) / 4096) chunks *= 2;   // fill the chip
  if (chunks == 0) chunks = 1;
  const uint32_t chunk = (uint32_t)((((n + chunks - 1) / chunks) + 7) & ~(size_t)7);   // multiple of 8 (vector loads)
  const size_t lds = wide ? ((size_t)4 << FINE_BITS) : (size_t)B * sizeof(uint32_t);
  // function attributes belong to the device's copy of the kernel: once per device.  The enqueue for one device may come from the caller's
  // thread (device-resident fan-out, under capi.hip's g_mu) AND from that device's worker thread (host-buffer fan-out, without it), so the
  // one-time setup is guarded per device rather than left to the callers' serialisation.
  static std::once_flag attr_once[64];
  static hipError_t attr_err[64];
  int cur_dev = 0;
  HIPCHK(hipGetDevice(&cur_dev));
  std::call_once(attr_once[cur_dev & 63], [&] {
    hipError_t e = hipFuncSetAttribute((const void*)k_sort_pass<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_sort_pass<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_fine_sorted, hipFuncAttributeMaxDynamicSharedMemorySize, (2 * (1 << FINE_BITS) + SORT_CHUNK + SORT_CHUNK / 16) * 4);
    attr_err[cur_dev & 63] = e;
  });
  HIPCHK(attr_err[cur_dev & 63]);
  const uint32_t gstride = (wide && !shared_buckets) ? (B >> FINE_BITS) : 0u;       // groups per window when every window has its own bucket set
  const int G = wide ? (int)(gstride ? (uint32_t)W * gstride : (B >> FINE_BITS)) : 1;                      // coarse groups
  const uint32_t sort_chunk = SORT_CHUNK;
  const uint32_t sort_chunks = (uint32_t)(((size_t)W * n + sort_chunk - 1) / sort_chunk) + (uint32_t)G;     // upper bound of the fine passes' chunks; surplus workgroups return at once
  const uint32_t wb_stride = shared_buckets ? 0u : B;
  if (wide) {
    static const bool fused_count2 = getenv("ZKHIP_NO_FUSED_COUNT") == nullptr;
    if (!fused_count2 || glv)        // (otherwise k_digits_wide has counted the groups; the GLV digit kernel -- one scalar per thread, thousands of
                                     // workgroups -- leaves the counting to this pass: its merges would be that many same-address atomics)
      hipLaunchKernelGGL(k_coarse_count, dim3(chunks, W), dim3(1024), 0, stream, (const int32_t*)digits, n_pad, chunk, gcounters + 1024, gstride);
    hipLaunchKernelGGL(k_group_offsets, dim3(1), dim3(64), 0, stream, gcounters + 1024, W, G, gcounters + 128, gcounters + 3072, gcounters + 512, sort_chunk);
    static_assert(MAX_GROUPS == 128, "k_coarse_sorted scans two groups per lane of one wavefront");
    hipLaunchKernelGGL(k_coarse_sorted, dim3((n_pad + COARSE_CHUNK - 1) / COARSE_CHUNK, W), dim3(1024), 0, stream, (const int32_t*)digits, n_pad, gcounters + 3072,
                       stage_ref, stage_fine, ref_base, ref_stride, gstride);
    prof_mark(stream, "coarse");
    hipLaunchKernelGGL(k_fine_count, dim3(sort_chunks), dim3(1024), 0, stream, stage_fine, gcounters + 128, gcounters + 512, G, count, sort_chunk);
  } else {
    hipLaunchKernelGGL(k_sort_pass<false>, dim3(chunks, W, K), dim3(1024), lds, stream, (const int16_t*)digits, n_pad, chunk, c, count, (uint32_t*)nullptr, wb_stride, ref_base, ref_stride);
  }
  prof_mark(stream, "count");
  // 3. scan counts -> offset (+ cursor copy)
  const uint32_t nblk = (NB + SCAN_TILE - 1) / SCAN_TILE;
  const bool scan_single = NB <= SCAN_SINGLE_MAX;
  if (scan_single) {
    hipLaunchKernelGGL(k_scan_single<0>, dim3(1), dim3(1024), 0, stream, count, (uint32_t)NB, counters + 0, offset, cursor, 0u, (uint32_t*)nullptr);
  } else {
    hipLaunchKernelGGL(k_scan_sums2, dim3(nblk), dim3(SCAN_BLOCK), 0, stream, count, NB, bsum1, bsum2, task_shift);
    hipLaunchKernelGGL(k_scan_top2, dim3(1), dim3(SCAN_BLOCK), 0, stream, bsum1, bsum2, nblk, counters + 0, counters + 1);
    hipLaunchKernelGGL(k_scan_apply2, dim3(nblk), dim3(SCAN_BLOCK), 0, stream, count, NB, bsum1, bsum2, counters + 0, counters + 1, offset, cursor, task_off, task_shift);
  }
  prof_mark(stream, "scan");
  // 4. scatter
  if (wide) {
    hipLaunchKernelGGL(k_fine_sorted, dim3(sort_chunks), dim3(1024), (size_t)(2u * (1u << FINE_BITS) + SORT_CHUNK + SORT_CHUNK / 16) * 4, stream, stage_fine, stage_ref,
                       gcounters + 128, gcounters + 512, G, cursor, sorted);
  }
  else hipLaunchKernelGGL(k_sort_pass<true>, dim3(chunks, W, K), dim3(1024), lds, stream, (const int16_t*)digits, n_pad, chunk, c, cursor, sorted, wb_stride, ref_base, ref_stride);
  prof_mark(stream, "scatter");
  // 5. tasks
  static const unsigned task_wg_cap = [] { const char* e = getenv("ZKHIP_TASK_WGS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 4096 ? (unsigned)v : 256u; }();   // A/B knob (anything else: the default)
  const unsigned task_blocks = (unsigned)std::min<size_t>((NB + 255) / 256, task_wg_cap);
  if (scan_single) {
    hipLaunchKernelGGL(k_scan_single<1>, dim3(1), dim3(1024), 0, stream, count, (uint32_t)NB, counters + 1, task_off, (uint32_t*)nullptr, task_shift, counters + 192);
    hipLaunchKernelGGL(k_make_tasks, dim3(task_blocks), dim3(256), 0, stream, offset, task_off, NB, tasks, task_shift, counters + 2, (uint32_t*)nullptr, lay.seq_parts,
                       counters + 3, lay.heavy, lay.heavy_cap);
  } else {
    // (the task offsets came out of the dual scan of step 3)
    hipLaunchKernelGGL(k_make_tasks, dim3(task_blocks), dim3(256), 0, stream, offset, task_off, NB, tasks, task_shift, counters + 2, counters + 32, lay.seq_parts,
                       counters + 3, lay.heavy, lay.heavy_cap);
    hipLaunchKernelGGL(k_order_offsets, dim3(1), dim3(64), 0, stream, counters + 32, counters + 192);
  }
  hipLaunchKernelGGL(k_make_order, dim3(task_blocks), dim3(256), 0, stream, offset, task_off, NB, task_shift, counters + 192, sorted, order);
  prof_mark(stream, "tasks");
  out->c = c; out->W = W; out->WB = WB; out->B = B; out->NB = NB; out->task_shift = task_shift; out->max_tasks = lay.max_tasks; out->nk = n * K;
  out->tasks = tasks; out->ntasks = counters + 1; out->max_parts = counters + 2; out->order = order; out->sorted = sorted; out->task_off = task_off;
  out->partials = lay.partials; out->pyrA = lay.pyrA; out->pyrB = lay.pyrB; out->winsum = lay.winsum;
  out->combine_lanes = lay.combine_lanes; out->seq_parts = lay.seq_parts; out->heavy_count = counters + 3; out->heavy = lay.heavy; out->heavy_cap = lay.heavy_cap; out->heavy_done = lay.heavy_done;
  out->endo = lay.endo;
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}

// Steps 6 - 7 for one task list: bucket accumulation + per-bucket combine; leaves the dense bucket array (NB XYZZ points, empty buckets = identity)
// at `buckets`.  d_bases: the caller's points or the prepared table.
static int msm_accumulate_combine(const msm_tasks_view& tv, const uint32_t* d_bases, bool prepared, bool glv, size_t n, uint32_t* buckets, hipStream_t stream) {
  const uint32_t NB = tv.NB;
  const size_t max_tasks = tv.max_tasks;
  const task_t* const tasks = (const task_t*)tv.tasks;
  uint32_t* const counters = tv.ntasks - 1;
  const uint4* const order = tv.order;
  const uint32_t *const sorted = tv.sorted, *const task_off = tv.task_off;
  uint32_t* const partials = tv.partials;
  // 6. accumulate (grid-stride over the device-side task count)
  {
    uint32_t blocks = (uint32_t)((max_tasks + 127) / 128);
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (prepared) hipLaunchKernelGGL(k_accumulate<true>, dim3(blocks), dim3(128), 0, stream, tasks, counters + 1, order, sorted, d_bases, partials, buckets, (const uint32_t*)nullptr, 0xffffffffu);
    else hipLaunchKernelGGL(k_accumulate<false>, dim3(blocks), dim3(128), 0, stream, tasks, counters + 1, order, sorted, d_bases, partials, buckets, (const uint32_t*)tv.endo,
                            glv ? (uint32_t)n : 0xffffffffu);
  }
  prof_mark(stream, "accumulate");
  // 7. combine: one launch (per-bucket sums + the heavy-bucket workgroups)
  {
    const int combine_lanes = tv.combine_lanes;
    const uint32_t seq_parts = tv.seq_parts;
    const size_t lanes_total = (size_t)NB * combine_lanes;
    const bool quad = lanes_total * 4 <= 131072;        // far below the chip's lane count: the additions' latency is the step time
    static const uint32_t heavy_quad_max = [] { const char* e = getenv("ZKHIP_HEAVY_QUAD_MAX"); const long v = e ? atol(e) : -1; return v >= 0 && v <= 2048 ? (uint32_t)v : HEAVY_QUAD_MAX; }();   // A/B knob (0 = the 128-lane slice sums only)
    const unsigned blocks = (unsigned)((lanes_total * (quad ? 4 : 1) + 127) / 128);
#define ZK_LAUNCH_COMBINE(L)                                                                                                                        \
    do {                                                                                                                                              \
      if (quad) hipLaunchKernelGGL(k_combine_seq_quad<L>, dim3(blocks + HEAVY_WGS), dim3(128), 0, stream, task_off, NB, partials, buckets, seq_parts, blocks,   \
                                   tv.heavy_count, (const uint2*)tv.heavy, tv.heavy_cap, tv.heavy_done, heavy_quad_max);                              \
      else hipLaunchKernelGGL(k_combine_seq<L>, dim3(blocks + HEAVY_WGS), dim3(128), 0, stream, task_off, NB, partials, buckets, seq_parts, blocks,             \
                              tv.heavy_count, (const uint2*)tv.heavy, tv.heavy_cap, tv.heavy_done, heavy_quad_max);                                   \
    } while (0)
    if (combine_lanes == 1) ZK_LAUNCH_COMBINE(1);
    else if (combine_lanes == 2) ZK_LAUNCH_COMBINE(2);
    else if (combine_lanes == 4) ZK_LAUNCH_COMBINE(4);
    else ZK_LAUNCH_COMBINE(8);
#undef ZK_LAUNCH_COMBINE
  }
  prof_mark(stream, "combine");
  return ZKHIP_OK;
}

// Steps 8 - 9: dense buckets (WB sets of B, at `cur`; `nxt` = a second array of the same size, both are overwritten) -> the MSM's result(s) at d_out.
// prepared: every bucket set's weighted sum is a result (K = WB of them); otherwise the WB window sums are folded into one.
static int msm_reduce_buckets(int WB, uint32_t B, int c, uint32_t K, bool prepared, uint32_t* cur, uint32_t* nxt, uint32_t* winsum, uint32_t* d_out, hipStream_t stream) {
  // 8. pyramid: buckets were written with window stride B (dense).  Steps 1 .. c-2 leave X with 2 elements.
  static const size_t quad_max = [] { const char* e = getenv("ZKHIP_PYR_QUAD_MAX"); const long v = e ? atol(e) : 0; return v >= 256 && v <= (1 << 20) ? (size_t)v : (size_t)65536; }();   // A/B knob
  static const long pyr_persist = [] { const char* e = getenv("ZKHIP_PYR_PERSIST"); const long v = e ? atol(e) : 0; return v >= 64 && v <= 65536 ? v : 0L; }();   // A/B knob: workgroup cap of the single-lane steps (0 = one addition per thread)
  uint32_t in_stride = B;
  int nz = 0;
  {
    uint32_t N = B;
    int s = 1;
    while (N > 2) {
      uint32_t per_win = N / 2 + (uint32_t)s * (N / 4);
      uint32_t out_stride = per_win;
      // below ~1/4 of the chip's lanes an addition's latency is the step time: four lanes per addition
      if ((size_t)per_win * WB <= quad_max) hipLaunchKernelGGL(k_pyramid_step_quad, dim3((4 * per_win + 127) / 128, WB), dim3(128), 0, stream, cur, nxt, N, s, in_stride, out_stride);
      else {
        unsigned blocks = (per_win + 127) / 128;
        if (pyr_persist > 0 && WB == 1) blocks = std::min<unsigned>(blocks, (unsigned)pyr_persist);
        hipLaunchKernelGGL(k_pyramid_step, dim3(blocks, WB), dim3(128), 0, stream, cur, nxt, N, s, in_stride, out_stride);
      }
      uint32_t* t = cur; cur = nxt; nxt = t;
      in_stride = out_stride;
      N >>= 1;
      s++;
    }
    nz = s - 1;   // number of Z rows, one element each (B == 2: none)
  }
  prof_mark(stream, "pyramid");
  // 9. Horner + fold
  if (B == 1) { set_error("msm: internal: B == 1"); return ZKHIP_EINVAL; }     // (c >= 2, so B >= 2 always)
  const bool direct = prepared && WB <= 2048;     // the weighted-sum kernel writes the results itself (one launch less)
  if (WB <= 2048) hipLaunchKernelGGL(k_window_horner_quad, dim3(WB), dim3(128), 0, stream, cur, in_stride, nz, winsum, direct ? d_out : (uint32_t*)nullptr);
  else hipLaunchKernelGGL(k_window_horner, dim3((WB * 32 + 63) / 64), dim3(64), 0, stream, cur, in_stride, nz, winsum, WB);
  prof_mark(stream, "horner");
  if (prepared) { if (!direct) hipLaunchKernelGGL(k_store_results, dim3((K + 63) / 64), dim3(64), 0, stream, winsum, K, d_out); }   // one result per bucket set
  else hipLaunchKernelGGL(k_fold, dim3(1), dim3(64), 0, stream, winsum, WB, c, d_out);
  prof_mark(stream, "fold");
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}

// d_scalars: n x 8 words, d_bases: n x 16 words, d_out: 24 words (device).  ws: workspace of msm_workspace_bytes.
// prepared != nullptr: d_bases is ignored, points come from the table (window w of point i at table[w * stride + off + i]).
// batch > 1 (prepared path, c <= 16 only): `batch` scalar vectors of n elements, vector k at d_scalars + k * scalar_stride
// elements, all against the same bases; d_out receives `batch` results (24 words each).
int msm_g1_device(const uint32_t* d_scalars, const uint32_t* d_bases, size_t n, uint32_t* d_out, void* ws, size_t ws_bytes,
                  int c_override, hipStream_t stream, const prepared_bases* prepared, size_t prepared_off, size_t batch,
                  size_t scalar_stride) {
  if (batch == 0) return ZKHIP_OK;
  if (n == 0) {
    for (size_t k = 0; k < batch; k++) hipLaunchKernelGGL(k_sum_jacobian, dim3(1), dim3(64), 0, stream, (const uint32_t*)nullptr, 0, d_out + k * 24, 1u);
    HIPCHK(hipGetLastError());
    return ZKHIP_OK;
  }
  if (batch > 1 && (!prepared || prepared->c > 16 || batch > 65535)) { set_error("msm: batch needs prepared bases with a window <= 16 bits"); return ZKHIP_EINVAL; }
  const uint32_t K = (uint32_t)batch;
  if (n >= (1ull << 31)) { set_error("msm: n = %zu too large", n); return ZKHIP_EINVAL; }
  const int c = prepared ? prepared->c : (c_override > 0 ? c_override : msm_pick_window(n));
  if (c < 2 || c > (prepared ? MAX_WINDOW_PREPARED : 16)) { set_error("msm: window bits %d out of range", c); return ZKHIP_EINVAL; }
  if (prepared) {
    if (prepared_off + n > prepared->n) { set_error("msm: range exceeds the prepared bases"); return ZKHIP_EINVAL; }
    if ((size_t)((256 + c - 1) / c) * prepared->n >= (1ull << 31)) { set_error("msm: prepared table too large for 31-bit point references"); return ZKHIP_EINVAL; }
    d_bases = prepared->table;
    if (prepared->direct && batch == 1) return msm_g1_direct(d_scalars, n, prepared, prepared_off, d_out, ws, ws_bytes, stream);     // small fixed base set: no buckets
  }
  msm_tasks_view tv;
  const bool glv = msm_uses_glv(prepared != nullptr, batch, 144);
  int rc = msm_build_tasks(d_scalars, n, batch, scalar_stride, c, prepared != nullptr, prepared ? (uint32_t)prepared_off : 0u, prepared ? (uint32_t)prepared->n : 0u,
                           144, ws, ws_bytes, stream, &tv, glv);
  if (rc != ZKHIP_OK) return rc;
  if (glv) hipLaunchKernelGGL(k_endo_bases, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_bases, (uint32_t)n, tv.endo);
  if ((rc = msm_accumulate_combine(tv, d_bases, prepared != nullptr, glv, n, tv.pyrA, stream)) != ZKHIP_OK) return rc;
  return msm_reduce_buckets(tv.WB, tv.B, c, K, prepared != nullptr, tv.pyrA, tv.pyrB, tv.winsum, d_out, stream);
}

// ------------------------------------------------------------------------------------------------
// Chunked prepared MSM (round 4): the scalars arrive in pieces -- a host buffer crossing PCIe chunk by chunk (capi.hip: host_msm) -- and every
// piece is sorted and accumulated as soon as it has landed, INTO THE SAME BUCKET SET: the bucket sums of piece j are added, bucket by bucket, to
// those of the pieces before it (a sum over points is a sum over pieces of sums), and the reduction tail runs once at the end.  Against cutting the
// MSM into independent shards this pays one elementwise pass of 2^(c-1) general additions per extra piece (~0.05 ms at c = 20) instead of a whole
// tail (0.34 ms), and the table, the window size and the bucket set are those of the unchunked MSM: the result is the same group element.
//   msm_chunk_workspace_bytes(chunk, c)   workspace for pieces of at most `chunk` scalars
//   msm_chunk_add (once per piece, any sizes <= chunk; the FIRST piece is passed with first = true and must be enqueued before the others: it
//   initialises the running bucket sums the later pieces add into) / msm_chunk_finish
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(128) k_bucket_sets_add(uint32_t* __restrict__ acc, const uint32_t* __restrict__ add, uint32_t nb) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nb) return;
  store_xyzz(acc, k, xyzz_add(load_xyzz(acc, k), load_xyzz(add, k)));
}

// bytes of the per-piece layout: pieces are chunk_cap or chunk_cap - 1 scalars (an even split), and the layout is not monotone in n where the
// task length switches, so both are covered; any other piece size is checked against this bound by msm_build_tasks (an error, never an overrun)
static size_t msm_chunk_lay_bytes(size_t chunk_cap, int c) {
  size_t b = msm_lay_out(nullptr, chunk_cap, 1, c, true).total;
  if (chunk_cap > 1) b = std::max(b, msm_lay_out(nullptr, chunk_cap - 1, 1, c, true).total);
  return align_up(b, 256);
}

size_t msm_chunk_workspace_bytes(size_t chunk_cap, int c) {
  if (chunk_cap == 0) return 0;
  return msm_chunk_lay_bytes(chunk_cap, c) + align_up(((size_t)1 << (c - 1)) * 144, 256);
}

int msm_chunk_add(const uint32_t* d_scalars, size_t n, const prepared_bases* pb, size_t pb_off, size_t chunk_cap, bool first, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!pb || n == 0 || n > chunk_cap || pb_off + n > pb->n) { set_error("msm_chunk_add: bad piece (n = %zu, capacity %zu)", n, chunk_cap); return ZKHIP_EINVAL; }
  const size_t lay_bytes = msm_chunk_lay_bytes(chunk_cap, pb->c);
  if (ws_bytes < msm_chunk_workspace_bytes(chunk_cap, pb->c)) { set_error("msm_chunk_add: workspace too small"); return ZKHIP_EINVAL; }
  uint32_t* const acc = (uint32_t*)((char*)ws + lay_bytes);
  msm_tasks_view tv;
  int rc = msm_build_tasks(d_scalars, n, 1, 0, pb->c, true, (uint32_t)pb_off, (uint32_t)pb->n, 144, ws, lay_bytes, stream, &tv, false);
  if (rc != ZKHIP_OK) return rc;
  if ((rc = msm_accumulate_combine(tv, pb->table, true, false, n, first ? acc : tv.pyrA, stream)) != ZKHIP_OK) return rc;
  if (!first) hipLaunchKernelGGL(k_bucket_sets_add, dim3((tv.NB + 127) / 128), dim3(128), 0, stream, acc, (const uint32_t*)tv.pyrA, tv.NB);
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}

int msm_chunk_finish(const prepared_bases* pb, size_t chunk_cap, uint32_t* d_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!pb || ws_bytes < msm_chunk_workspace_bytes(chunk_cap, pb->c)) { set_error("msm_chunk_finish: workspace too small"); return ZKHIP_EINVAL; }
  const msm_layout lay = msm_lay_out((char*)ws, chunk_cap, 1, pb->c, true);
  uint32_t* const acc = (uint32_t*)((char*)ws + msm_chunk_lay_bytes(chunk_cap, pb->c));
  return msm_reduce_buckets(1, 1u << (pb->c - 1), pb->c, 1, true, acc, lay.pyrA, lay.winsum, d_out, stream);
}


// ------------------------------------------------------------------------------------------------
// synthetic bases: out[i] = (t0 + i d) G.  Thread j owns GEN_CHUNK consecutive points: start = (t0 + j CH d) G by
// double-and-add, then a walk of XYZZ additions of D = d G, then Montgomery-trick normalisation of its chunk.
// ------------------------------------------------------------------------------------------------
constexpr int GEN_CHUNK = 32;

__device__ __forceinline__ xyzz scalar_mul_affine(const uint32_t (&kw)[8], const fe& gx, const fe& gy) {
  xyzz acc = xyzz_identity();
  for (int bit = 255; bit >= 0; bit--) {
    acc = xyzz_dbl(acc);
    if ((kw[bit >> 5] >> (bit & 31)) & 1) xyzz_madd(acc, gx, gy);
  }
  return acc;
}

// a^-1 in Fq: a reduced (< 2p), result < 2p.  Division steps instead of the Fermat chain a^(q-2) (fe_inverse.hpp): ~11 k instead of
// ~82 k instructions of one thread, under every batched affine conversion below.
__device__ __forceinline__ fe fq_inverse(const fe& a) { return fe_inverse<Fq>(a); }

__global__ void __launch_bounds__(64) k_gen_walk(const uint32_t* __restrict__ t0_d /* 16 words: t0, d (Montgomery-256) */,
                                                 uint32_t n, uint32_t* __restrict__ out, uint32_t* __restrict__ tmp_pts,
                                                 uint32_t* __restrict__ tmp_pref) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lo = j * GEN_CHUNK;
  if (lo >= n) return;
  const uint32_t cnt = min((uint32_t)GEN_CHUNK, n - lo);
  // scalars: k = t0 + lo * d in Fr
  uint32_t w0[8], wd[8], kw[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { w0[i] = t0_d[i]; wd[i] = t0_d[8 + i]; }
  fe one_r = fe_one<FrParams>();
  fe t0 = fe_mul<FrParams>(one_r, fe_from_ext_lazy(w0));          // t0 * 2^261
  fe dd = fe_mul<FrParams>(one_r, fe_from_ext_lazy(wd));
  fe lo_f = fe_zero();
  lo_f.l[0] = lo & LMASK; lo_f.l[1] = lo >> LB;                   // plain integer lo
  fe r2; 
#pragma unroll
  for (int i = 0; i < NL; i++) r2.l[i] = FrParams::R2[i];
  fe lo_m = fe_mul<FrParams>(r2, lo_f);                           // lo * 2^261
  fe k = fe_norm(fe_add(t0, fe_mul<FrParams>(lo_m, dd)));         // < 4p
  fe raw1;
#pragma unroll
  for (int i = 0; i < NL; i++) raw1.l[i] = FrParams::RAW_ONE[i];
  fe kc = fe_canon_lt2p<FrParams>(fe_mul<FrParams>(raw1, k));      // canonical integer
  fe_pack(kc, kw);
  fe dc = fe_canon_lt2p<FrParams>(fe_mul<FrParams>(raw1, dd));
  uint32_t dw[8];
  fe_pack(dc, dw);
  // generator (1, 2) in internal Montgomery form
  fe gx = fe_one<Fq>();
  fe gy = fe_norm(fe_dbl(gx));
  xyzz P = scalar_mul_affine(kw, gx, gy);
  xyzz D = scalar_mul_affine(dw, gx, gy);
  // walk + prefix products of w_i = ZZ_i * ZZZ_i (identity points contribute 1)
  fe pref = fe_one<Fq>();
  for (uint32_t i = 0; i < cnt; i++) {
    store_xyzz(tmp_pts, lo + i, P);
    store_fe9_generic(tmp_pref, lo + i, pref);
    if (!xyzz_is_identity(P)) pref = fe_mul<Fq>(pref, fe_mul<Fq>(P.ZZ, P.ZZZ));
    P = xyzz_add(P, D);
  }
  fe inv = fq_inverse(pref);
  for (uint32_t ii = cnt; ii-- > 0;) {
    xyzz Q = load_xyzz(tmp_pts, lo + ii);
    uint32_t* o = out + (size_t)(lo + ii) * 16;
    if (xyzz_is_identity(Q)) {
#pragma unroll
      for (int t = 0; t < 16; t++) o[t] = 0;
      continue;
    }
    fe pre = load_fe9_generic(tmp_pref, lo + ii);
    fe winv = fe_mul<Fq>(inv, pre);                               // 1 / (ZZ * ZZZ)
    inv = fe_mul<Fq>(inv, fe_mul<Fq>(Q.ZZ, Q.ZZZ));
    fe x = fe_mul<Fq>(fe_mul<Fq>(winv, Q.ZZZ), Q.X);              // X / ZZ
    fe y = fe_mul<Fq>(fe_mul<Fq>(winv, Q.ZZ), Q.Y);               // Y / ZZZ
    uint32_t wx[8], wy[8];
    fe_to_ext<Fq>(x, wx);
    fe_to_ext<Fq>(y, wy);
#pragma unroll
    for (int t = 0; t < 8; t++) { o[t] = wx[t]; o[8 + t] = wy[t]; }
  }
}

// Montgomery words of 2^e (e < 256), i.e. 2^e * 2^256 mod r, by e modular doublings of R = 2^256 mod r on the host
static void fr_pow2_montgomery(int e, uint32_t out[8]) {
  static const uint64_t RMOD[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
  uint64_t v[4] = {0xac96341c4ffffffbull, 0x36fc76959f60cd29ull, 0x666ea36f7879462eull, 0x0e0a77c19a07df2full};   // R mod r
  for (int k = 0; k < e; k++) {
    uint64_t carry = 0;                                 // v < r < 2^254: the doubled value fits 256 bits
    for (int i = 0; i < 4; i++) { const uint64_t nc = v[i] >> 63; v[i] = (v[i] << 1) | carry; carry = nc; }
    bool ge = true;
    for (int i = 3; i >= 0; i--) { if (v[i] != RMOD[i]) { ge = v[i] > RMOD[i]; break; } }
    if (ge) {
      unsigned __int128 borrow = 0;
      for (int i = 0; i < 4; i++) {
        const unsigned __int128 d = (unsigned __int128)v[i] - RMOD[i] - borrow;
        v[i] = (uint64_t)d;
        borrow = (d >> 64) ? 1 : 0;
      }
    }
  }
  for (int i = 0; i < 4; i++) { out[2 * i] = (uint32_t)v[i]; out[2 * i + 1] = (uint32_t)(v[i] >> 32); }
}

size_t g1_gen_walk_workspace(size_t n) { return align_up(n * 144, 256) + align_up(n * 36, 256) + 256; }

int g1_gen_walk_device(const uint32_t t0_ext[8], const uint32_t d_ext[8], size_t n, uint32_t* d_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (n == 0) return ZKHIP_OK;
  if (n >= (1ull << 31)) { set_error("gen_walk: n too large"); return ZKHIP_EINVAL; }
  if (ws_bytes < g1_gen_walk_workspace(n)) { set_error("gen_walk: workspace too small"); return ZKHIP_EINVAL; }
  char* p = (char*)ws;
  uint32_t* tmp_pts = (uint32_t*)p; p += align_up(n * 144, 256);
  uint32_t* tmp_pref = (uint32_t*)p; p += align_up(n * 36, 256);
  uint32_t* consts = (uint32_t*)p;
  uint32_t h[16];
  memcpy(h, t0_ext, 32); memcpy(h + 8, d_ext, 32);
  HIPCHK(hipMemcpyAsync(consts, h, 64, hipMemcpyHostToDevice, stream));
  HIPCHK(hipStreamSynchronize(stream));   // `h` is a stack buffer
  const uint32_t threads = (uint32_t)((n + GEN_CHUNK - 1) / GEN_CHUNK);
  hipLaunchKernelGGL(k_gen_walk, dim3((threads + 63) / 64), dim3(64), 0, stream, consts, (uint32_t)n, d_out, tmp_pts, tmp_pref);
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}


// ------------------------------------------------------------------------------------------------
// fixed-base multiplication by the generator: out[i] = k_i G as affine points -- what `ParamsKZG::setup` [DEP halo2-axiom
// poly/kzg/commitment.rs] does n times for g = [s^i] G and n times for g_lagrange = [L_i(s)] G (reached from
// /root/reference/voter/benches/voter_circuit.rs:60, aggregator/benches/state_transition_circuit.rs:64).
// Table T[w][d - 1] = d 2^(16 w) G for d = 1 .. 2^15, w < 16 (32 MiB, built once with the generator walk above); a scalar is 16 signed
// 16-bit digits -> at most 16 mixed additions.  A thread owns FIXED_CHUNK consecutive scalars and normalises its results with one
// inversion, like k_gen_walk.
// ------------------------------------------------------------------------------------------------
constexpr int FIXED_C = 16, FIXED_W = 16, FIXED_CHUNK = 32;
constexpr uint32_t FIXED_B = 1u << (FIXED_C - 1);

__global__ void __launch_bounds__(64) k_fixed_base_mul(const uint32_t* __restrict__ scalars, uint32_t n, const uint32_t* __restrict__ table,
                                                       uint32_t* __restrict__ out, uint32_t* __restrict__ tmp_pts, uint32_t* __restrict__ tmp_pref) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lo = j * FIXED_CHUNK;
  if (lo >= n) return;
  const uint32_t cnt = min((uint32_t)FIXED_CHUNK, n - lo);
  fe c32;
#pragma unroll
  for (int k = 0; k < NL; k++) c32.l[k] = FrParams::FROM_EXT_CANON[k];
  fe pref = fe_one<Fq>();
  for (uint32_t i = 0; i < cnt; i++) {
    uint32_t w[8];
    load_words(scalars + (size_t)(lo + i) * 8, w);
    fe_pack(fe_canon_lt2p<FrParams>(fe_mul<FrParams>(c32, fe_unpack<0>(w))), w);      // the scalar as a plain integer
    xyzz acc = xyzz_identity();
    uint32_t carry = 0;
#pragma unroll 1
    for (int win = 0; win < FIXED_W; win++) {
      const uint32_t v = ((w[win >> 1] >> ((win & 1) * 16)) & 0xffffu) + carry;
      int32_t d;
      if (v >= FIXED_B) { d = (int32_t)v - (int32_t)(1u << FIXED_C); carry = 1; } else { d = (int32_t)v; carry = 0; }
      if (d == 0) continue;
      const affine_words pt = load_affine(table, (size_t)win * FIXED_B + (uint32_t)(d < 0 ? -d : d) - 1);
      fe y2 = fe_from_ext_lazy(pt.y);
      if (d < 0) y2 = fe_neg_red(y2, Fq::P64_S1);
      xyzz_madd(acc, fe_from_ext_lazy(pt.x), y2);
    }
    // carry out of the top window is impossible: scalars are below r < 2^254, so the top digit is below 2^14
    store_xyzz(tmp_pts, lo + i, acc);
    store_fe9_generic(tmp_pref, lo + i, pref);
    if (!xyzz_is_identity(acc)) pref = fe_mul<Fq>(pref, fe_mul<Fq>(acc.ZZ, acc.ZZZ));
  }
  fe inv = fq_inverse(pref);
  for (uint32_t ii = cnt; ii-- > 0;) {
    const xyzz Q = load_xyzz(tmp_pts, lo + ii);
    uint32_t* o = out + (size_t)(lo + ii) * 16;
    if (xyzz_is_identity(Q)) {
#pragma unroll
      for (int t = 0; t < 16; t++) o[t] = 0;
      continue;
    }
    const fe pre = load_fe9_generic(tmp_pref, lo + ii);
    const fe winv = fe_mul<Fq>(inv, pre);                           // 1 / (ZZ * ZZZ)
    inv = fe_mul<Fq>(inv, fe_mul<Fq>(Q.ZZ, Q.ZZZ));
    const fe x = fe_mul<Fq>(fe_mul<Fq>(winv, Q.ZZZ), Q.X);          // X / ZZ
    const fe y = fe_mul<Fq>(fe_mul<Fq>(winv, Q.ZZ), Q.Y);           // Y / ZZZ
    uint32_t wx[8], wy[8];
    fe_to_ext<Fq>(x, wx);
    fe_to_ext<Fq>(y, wy);
#pragma unroll
    for (int t = 0; t < 8; t++) { o[t] = wx[t]; o[8 + t] = wy[t]; }
  }
}

size_t g1_fixed_base_table_bytes() { return (size_t)FIXED_W * FIXED_B * 64; }
size_t g1_fixed_base_workspace(size_t n) {
  const size_t a = g1_gen_walk_workspace(FIXED_B), b = align_up(n * 144, 256) + align_up(n * 36, 256);
  return (a > b ? a : b) + 256;
}

// d_table: g1_fixed_base_table_bytes() of device memory, filled here (window w = the walk 2^(16 w), 2 * 2^(16 w), ...)
int g1_fixed_base_table_build(uint32_t* d_table, void* ws, size_t ws_bytes, hipStream_t stream) {
  for (int w = 0; w < FIXED_W; w++) {
    // step = 2^(16 w) as a Montgomery Fr element: host big-integer shift of R = 2^256 mod r is avoided by building the constant
    // from its plain words with one device multiply inside k_gen_walk -- it takes Montgomery words, so pass 2^(16 w) * R mod r
    uint32_t step[8];
    fr_pow2_montgomery(16 * w, step);
    int rc = g1_gen_walk_device(step, step, FIXED_B, d_table + (size_t)w * FIXED_B * 16, ws, ws_bytes, stream);
    if (rc != ZKHIP_OK) return rc;
  }
  return ZKHIP_OK;
}

int g1_fixed_base_mul_device(const uint32_t* d_scalars, size_t n, const uint32_t* d_table, uint32_t* d_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (n == 0) return ZKHIP_OK;
  if (n >= (1ull << 31)) { set_error("fixed_base_mul: n too large"); return ZKHIP_EINVAL; }
  if (ws_bytes < g1_fixed_base_workspace(n)) { set_error("fixed_base_mul: workspace too small"); return ZKHIP_EINVAL; }
  char* p = (char*)ws;
  uint32_t* tmp_pts = (uint32_t*)p; p += align_up(n * 144, 256);
  uint32_t* tmp_pref = (uint32_t*)p;
  const uint32_t threads = (uint32_t)((n + FIXED_CHUNK - 1) / FIXED_CHUNK);
  hipLaunchKernelGGL(k_fixed_base_mul, dim3((threads + 63) / 64), dim3(64), 0, stream, d_scalars, (uint32_t)n, d_table, d_out, tmp_pts, tmp_pref);
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// prepared bases: table[w * n + i] = 2^(c w) * P_i, w < W, affine, each coordinate the canonical internal-form value
// (x * 2^261 mod p) packed into 8 words -- the library's own format, read only by k_accumulate<true>; (0, 0) = identity.
// One thread per point walks the doubling chain, parks the XYZZ multiples in `tmp`, then normalises its W-1 points with one
// inversion.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void store_table_point(uint32_t* table, size_t idx, const fe& x, const fe& y) {   // x, y < 2p
  uint32_t wx[8], wy[8];
  fe_pack(fe_canon_lt2p<Fq>(x), wx);
  fe_pack(fe_canon_lt2p<Fq>(y), wy);
  uint4* q = reinterpret_cast<uint4*>(table + idx * 16);
  q[0] = make_uint4(wx[0], wx[1], wx[2], wx[3]); q[1] = make_uint4(wx[4], wx[5], wx[6], wx[7]);
  q[2] = make_uint4(wy[0], wy[1], wy[2], wy[3]); q[3] = make_uint4(wy[4], wy[5], wy[6], wy[7]);
}

__global__ void __launch_bounds__(64) k_build_table(const uint32_t* __restrict__ bases, uint32_t n, int c, int W,
                                                    uint32_t* __restrict__ table, uint32_t* __restrict__ tmp_pts,
                                                    uint32_t* __restrict__ tmp_pref, int narrow) {
  const win_layout lay{c, W, narrow};             // table[w] = 2^(win_off(w)) P: window w - 1 is win_bits(w - 1) doublings below window w
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  affine_words pt = load_affine(bases, i);
  if (affine_is_identity(pt)) {
    for (int w = 0; w < W; w++) {
      uint4* q = reinterpret_cast<uint4*>(table + ((size_t)w * n + i) * 16);
      q[0] = q[1] = q[2] = q[3] = make_uint4(0, 0, 0, 0);
    }
    return;
  }
  xyzz P = xyzz_identity();
  xyzz_madd(P, fe_from_ext_lazy(pt.x), fe_from_ext_lazy(pt.y));     // X, Y reduced (< 1.4p), internal form
  store_table_point(table, i, P.X, P.Y);
  fe pref = fe_one<Fq>();
  for (int w = 1; w < W; w++) {
    for (int k = 0, nd = win_bits(lay, w - 1); k < nd; k++) P = xyzz_dbl(P);
    store_xyzz(tmp_pts, (size_t)(w - 1) * n + i, P);
    store_fe9_generic(tmp_pref, (size_t)(w - 1) * n + i, pref);
    pref = fe_mul<Fq>(pref, fe_mul<Fq>(P.ZZ, P.ZZZ));     // never zero: the group has odd prime order
  }
  fe inv = fq_inverse(pref);
  for (int w = W - 1; w >= 1; w--) {
    xyzz Q = load_xyzz(tmp_pts, (size_t)(w - 1) * n + i);
    fe pre = load_fe9_generic(tmp_pref, (size_t)(w - 1) * n + i);
    fe winv = fe_mul<Fq>(inv, pre);
    inv = fe_mul<Fq>(inv, fe_mul<Fq>(Q.ZZ, Q.ZZZ));
    fe x = fe_mul<Fq>(fe_mul<Fq>(winv, Q.ZZZ), Q.X);
    fe y = fe_mul<Fq>(fe_mul<Fq>(winv, Q.ZZ), Q.Y);
    store_table_point(table, (size_t)w * n + i, x, y);
  }
}

// d_bases: n affine points on the device.  Allocates the table (W * n * 64 B) and a temporary (freed before returning).
int prepare_bases_device(const uint32_t* d_bases, size_t n, hipStream_t stream, prepared_bases** out, int c_override, size_t n_whole) {
  if (n == 0 || n >= (1ull << 27)) { set_error("prepare_bases: n = %zu out of range", n); return ZKHIP_EINVAL; }
  const int c = c_override > 0 ? c_override : msm_pick_window_prepared(n);
  if (c < 2 || c > MAX_WINDOW_PREPARED) { set_error("prepare_bases: window bits %d out of range [2,%d]", c, MAX_WINDOW_PREPARED); return ZKHIP_EINVAL; }
  const int W = (256 + c - 1) / c;
  prepared_bases* pb = new prepared_bases();
  pb->n = n; pb->c = c; pb->W = W; pb->table = nullptr; pb->direct = nullptr;
  void *table = nullptr, *tmp = nullptr;
  const size_t tbytes = (size_t)W * n * 64, tmp_pts = align_up((size_t)(W - 1) * n * 144, 256), tmp_pref = align_up((size_t)(W - 1) * n * 36, 256);
  if (hipMalloc(&table, tbytes) != hipSuccess) { delete pb; set_error("prepare_bases: hipMalloc(%zu) failed", tbytes); return ZKHIP_ENOMEM; }
  if (hipMalloc(&tmp, tmp_pts + tmp_pref) != hipSuccess) { (void)hipFree(table); delete pb; set_error("prepare_bases: hipMalloc(%zu) failed", tmp_pts + tmp_pref); return ZKHIP_ENOMEM; }
  hipLaunchKernelGGL(k_build_table, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, d_bases, (uint32_t)n, c, W, (uint32_t*)table,
                     (uint32_t*)tmp, (uint32_t*)((char*)tmp + tmp_pts), msm_narrow_windows(c, W));
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(tmp);
  if (e != hipSuccess) { (void)hipFree(table); delete pb; set_error("prepare_bases: %s", hipGetErrorString(e)); return ZKHIP_EHIP; }
  pb->table = (uint32_t*)table;
  // small sets with an automatic window also get the direct table (a failed allocation leaves the bucket path, which needs nothing more)
  if (c_override <= 0 && msm_direct_wanted(std::max(n, n_whole)) && prepare_direct_table(pb, d_bases, stream) != ZKHIP_OK) pb->direct = nullptr;
  *out = pb;
  return ZKHIP_OK;
}

size_t direct_table_bytes(size_t n);
static std::atomic<size_t> g_direct_bytes{0};          // bytes held by the direct tables of all prepared sets of the process (prepare_direct_table's budget)

void release_prepared(prepared_bases* pb) {
  if (!pb) return;
  if (pb->table) (void)hipFree(pb->table);
  if (pb->direct) { (void)hipFree(pb->direct); g_direct_bytes -= direct_table_bytes(pb->n); }
  delete pb;
}

// ------------------------------------------------------------------------------------------------
// Direct tables: bucket-free MSM for the small circuits' SRS (round 4).
//
// An unmodified `create_proof` commits column by column (`params.commit_lagrange(&column)` in a loop: [DEP] halo2-axiom plonk/prover.rs, reached from
// /root/reference/aggregator/src/wrapper.rs:129 and the voter / state-transition benches, /root/reference/voter/benches/voter_circuit.rs:80,
// /root/reference/aggregator/benches/state_transition_circuit.rs:84), so at k = 13 .. 15 it sees ONE small MSM at a time -- and a Pippenger run of 2^13
// points is 0.23 ms of pure latency on this machine: ~36 dependent launches, of which the bucket reduction (pyramid + weighted sum + combine) is 63 %.
// For a FIXED base set that small, HBM buys the buckets off entirely: with 8-bit signed windows the table holds every multiple a digit can ask for,
//     direct[i][w][m - 1] = m * 2^(8 w) * P_i,    i < n, w < 32, m = 1 .. 128     (n * 256 KiB: 2 GiB at 2^13 points, 8 GiB at 2^15)
// and the MSM is a PLAIN sum of 32 n table points: no sort, no buckets, no weights, no doublings.  Two kernel shapes:
//     k_direct_accumulate   thread (i, g): the 4 digits of window group g of scalar i (recoded from the scalar itself: one Montgomery multiplication and a
//                           byte scan), 4 gathers, first point copied, 3 mixed additions -> 8 n partial sums
//     k_points_sum_quad     512-thread workgroups, one quad per addition: 4 partials per quad, then a shuffle tree over the wavefront's 16 quads and an LDS
//                           tree over the 8 wavefronts: 512 -> 1 in 10 dependent quad additions; two launches take 2^18 partials to the result
// Depth: 3 mixed + ~20 quad additions, against 11 + 11 pyramid / doubling steps, the combine and the sort chain of the bucket method.
// Signed recoding: digit_w = byte_w + carry in [-128, 127] u {128 -> -128 + carry}; the scalar is < 2^254, so the top window's byte is < 64: no carry out.
// ------------------------------------------------------------------------------------------------
constexpr int DIRECT_W = 32, DIRECT_M = 128;   // windows, multiples per window

__global__ void __launch_bounds__(64) k_direct_build(const uint32_t* __restrict__ wtable /* [32][n_all] reduced internal affine: 2^(8 w) P_i */, uint32_t n_all,
                                                     uint32_t i0, uint32_t cnt, uint32_t* __restrict__ direct, uint32_t* __restrict__ tmp_pts,
                                                     uint32_t* __restrict__ tmp_pref) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, T = cnt * DIRECT_W;
  if (tid >= T) return;
  const uint32_t i = i0 + tid / DIRECT_W, w = tid % DIRECT_W;
  const affine_words pt = load_affine(wtable, (size_t)w * n_all + i);
  uint4* const slot = reinterpret_cast<uint4*>(direct + ((size_t)i * DIRECT_W + w) * DIRECT_M * 16);
  if (affine_is_identity(pt)) {
    for (int k = 0; k < DIRECT_M * 4; k++) slot[k] = make_uint4(0, 0, 0, 0);
    return;
  }
  slot[0] = make_uint4(pt.x[0], pt.x[1], pt.x[2], pt.x[3]); slot[1] = make_uint4(pt.x[4], pt.x[5], pt.x[6], pt.x[7]);     // 1 * Q: the window table's own entry
  slot[2] = make_uint4(pt.y[0], pt.y[1], pt.y[2], pt.y[3]); slot[3] = make_uint4(pt.y[4], pt.y[5], pt.y[6], pt.y[7]);
  const fe x = fe_unpack<0>(pt.x), y = fe_unpack<0>(pt.y);
  xyzz acc = xyzz_identity();
  xyzz_madd(acc, x, y);
  fe pref = fe_one<Fq>();
  for (int k = 2; k <= DIRECT_M; k++) {             // k Q by repeated mixed addition (k = 2 takes the doubling branch); never the identity: k < r
    xyzz_madd(acc, x, y);
    store_xyzz(tmp_pts, (size_t)(k - 2) * T + tid, acc);
    store_fe9_generic(tmp_pref, (size_t)(k - 2) * T + tid, pref);
    pref = fe_mul<Fq>(pref, fe_mul<Fq>(acc.ZZ, acc.ZZZ));
  }
  fe inv = fq_inverse(pref);
  for (int k = DIRECT_M; k >= 2; k--) {
    const xyzz Q = load_xyzz(tmp_pts, (size_t)(k - 2) * T + tid);
    const fe pre = load_fe9_generic(tmp_pref, (size_t)(k - 2) * T + tid);
    const fe winv = fe_mul<Fq>(inv, pre);
    inv = fe_mul<Fq>(inv, fe_mul<Fq>(Q.ZZ, Q.ZZZ));
    store_table_point(direct, ((size_t)i * DIRECT_W + w) * DIRECT_M + (k - 1), fe_mul<Fq>(fe_mul<Fq>(winv, Q.ZZZ), Q.X), fe_mul<Fq>(fe_mul<Fq>(winv, Q.ZZ), Q.Y));
  }
}

template <int WPT>      // windows per thread: 1, 2, 4 or 8 (32 / WPT threads per scalar)
__device__ __forceinline__ void direct_accumulate_body(const uint32_t* __restrict__ scalars, uint32_t n, uint32_t off,
                                                       const uint32_t* __restrict__ direct, uint32_t* __restrict__ partials) {
  constexpr uint32_t GROUPS = DIRECT_W / WPT;
  constexpr int WORDS = WPT >= 4 ? WPT / 4 : 1;      // scalar words a thread reads its digits from
  constexpr int PER = WPT >= 4 ? 4 : WPT;            // windows per word
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= n * GROUPS) return;
  const uint32_t i = tid / GROUPS, g = tid % GROUPS;
  uint32_t w[8];
  load_words(scalars + (size_t)i * 8, w);
  fe c32;
#pragma unroll
  for (int k = 0; k < NL; k++) c32.l[k] = FrParams::FROM_EXT_CANON[k];
  fe sv = fe_canon_lt2p<FrParams>(fe_mul<FrParams>(c32, fe_unpack<0>(w)));       // Montgomery words -> the canonical integer (k_digits does the same)
  fe_pack(sv, w);
  // carry into this thread's first window: a byte scan from the bottom (carry_out = byte + carry_in >= 128), written over all 8 words with
  // compile-time indices so that w[] stays in registers
  uint32_t carry = 0;
  const uint32_t first = g * WPT;
#pragma unroll
  for (int wi = 0; wi < 8; wi++) {
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const uint32_t win = (uint32_t)(wi * 4 + b);
      const uint32_t v = ((w[wi] >> (8 * b)) & 255u) + carry;
      if (win < first) carry = v >= 128u ? 1u : 0u;
    }
  }
  xyzz acc = xyzz_identity();
  // one scalar word (up to 4 windows) at a time: digits, the gathers in flight together, then the additions (the first point of a thread is a copy).
  // (compile-time loops: under `#pragma unroll` the compiler keeps a loop whose body holds the asm-block multiplications rolled, which indexes
  // the point array dynamically and puts it into scratch)
  static_for<0, WORDS>([&](auto hc) {
    constexpr int h = decltype(hc)::value;
    const uint32_t word = first / 4 + h, byte0 = WPT >= 4 ? 0u : first % 4;
    uint32_t own = 0;
#pragma unroll
    for (int wi = 0; wi < 8; wi++) own = ((uint32_t)wi == word) ? w[wi] : own;
    int32_t dg[PER];
#pragma unroll
    for (int b = 0; b < PER; b++) {
      const uint32_t v = ((own >> (8 * (byte0 + b))) & 255u) + carry;
      if (v >= 128u) { dg[b] = (int32_t)v - 256; carry = 1; } else { dg[b] = (int32_t)v; carry = 0; }
    }
    affine_words pts[PER];
    static_for<0, PER>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      const uint32_t mag = (uint32_t)(dg[b] < 0 ? -dg[b] : dg[b]);
      const size_t idx = (((size_t)(off + i) * DIRECT_W + first + PER * h + b) * DIRECT_M) + (mag ? mag - 1 : 0);
      pts[b] = load_affine(direct, idx);
    });
    static_for<0, PER>([&](auto bc) {
      constexpr int b = decltype(bc)::value;
      if (dg[b] != 0 && !affine_is_identity(pts[b])) {
        fe x2 = fe_unpack<0>(pts[b].x), y2 = fe_unpack<0>(pts[b].y);
        if (dg[b] < 0) y2 = fe_neg_red(y2, Fq::P2_S1);        // limbs < 2^30, < 2p
        if (xyzz_is_identity(acc)) { acc.X = x2; acc.Y = dg[b] < 0 ? fe_norm(y2) : y2; acc.ZZ = acc.ZZZ = fe_one<Fq>(); }
        else xyzz_madd_nz<true>(acc, x2, y2);
      }
    });
  });
  store_xyzz(partials, tid, acc);
}
// Windows per thread.  Everything here is latency: a thread's k - 1 dependent mixed additions cost ~8 us each, the quad tree 80 / 89 / 102 / 111 us for
// 2^13 / 2^15 / 2^16 / 2^17 partial sums (measured, profiles/r04_small_msm_direct_ab.txt).  Up to 2^15 partial sums fewer windows per thread win
// (2^10 scalars: 1 window 0.094 ms, 4 windows 0.099); beyond that the tree's growth costs more than the additions saved (2^12: 1 window 0.121, 4 windows
// 0.110), and past 2^17 partial sums the first tree launch would need a second round of workgroups: 1 window per thread up to 2^10 scalars, 2 at 2^11,
// 4 at 2^12, 2 at 2^13, 4 at 2^14, 8 above (the measured optimum at every size; it is not monotone because both costs are step functions).  1 .. 4 windows: launches sized for 4 waves per SIMD (<= 128 VGPRs, no scratch); 8 windows: the scalar's words stay
// live across two rounds of gathers, it may take the registers it needs.
// (2^13 scalars, measured both ways on one box: 2 windows per thread 0.1275 ms, 4 windows 0.1333.)
static inline int direct_wpt(size_t n) {
  if (n <= 1024) return 1;
  if (n <= 2048) return 2;
  if (n <= 4096) return 4;
  if (n <= 8192) return 2;       // 2^17 partial sums: still one round of tree workgroups, and two dependent additions less per thread
  if (n <= 16384) return 4;
  return 8;
}
#define ZK_DIRECT_KERNEL(W, OCC_LO, OCC_HI)                                                                                                          \
  __global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(OCC_LO, OCC_HI))) k_direct_accumulate##W(                                 \
      const uint32_t* __restrict__ scalars, uint32_t n, uint32_t off, const uint32_t* __restrict__ direct, uint32_t* __restrict__ partials) {         \
    direct_accumulate_body<W>(scalars, n, off, direct, partials);                                                                                     \
  }
ZK_DIRECT_KERNEL(1, 4, 4)
ZK_DIRECT_KERNEL(2, 4, 4)
ZK_DIRECT_KERNEL(4, 4, 4)
ZK_DIRECT_KERNEL(8, 2, 3)
#undef ZK_DIRECT_KERNEL

// out[b] = sum of in[b * 128 S .. ): 128 quads per workgroup, quad t adds the S points t, t + 128, ... of the workgroup's slice (coalesced rows), a
// shuffle tree adds the 16 quads of a wavefront, an LDS tree the 8 wavefronts.  jac_out != nullptr (single workgroup): the Jacobian result.
__global__ void __launch_bounds__(512) k_points_sum_quad(const uint32_t* __restrict__ in, uint32_t m, uint32_t S, uint32_t* __restrict__ out,
                                                         uint32_t* __restrict__ jac_out) {
  __shared__ __attribute__((aligned(16))) uint32_t xch[8 * 36];
  const uint32_t q = threadIdx.x & 3, quad = threadIdx.x >> 2, wave = threadIdx.x >> 6;
  const uint32_t base = blockIdx.x * 128u * S;
  xyzz acc = xyzz_identity();
#pragma unroll 1
  for (uint32_t s_ = 0; s_ < S; s_++) {
    const uint32_t idx = base + s_ * 128u + quad;
    if (idx < m) acc = xyzz_add_quad(acc, load_xyzz(in, idx), q);          // uniform over the quad
  }
#pragma unroll 1
  for (int mask = 4; mask < 64; mask <<= 1) acc = xyzz_add_quad(acc, xyzz_shfl_xor(acc, mask), q);   // every quad of the wavefront ends with its sum
  if ((threadIdx.x & 63) == 0) store_xyzz(xch, wave, acc);
  __syncthreads();
  if (wave == 0) {
    acc = (quad < 8) ? load_xyzz(xch, quad) : xyzz_identity();
#pragma unroll 1
    for (int mask = 4; mask < 32; mask <<= 1) acc = xyzz_add_quad(acc, xyzz_shfl_xor(acc, mask), q);
    if (threadIdx.x == 0) {
      if (jac_out) store_jacobian(acc, jac_out);
      else store_xyzz(out, blockIdx.x, acc);
    }
  }
}

size_t direct_table_bytes(size_t n) { return n * (size_t)DIRECT_W * DIRECT_M * 64; }

static inline size_t direct_partials(size_t n) { return n * (size_t)DIRECT_W / (size_t)direct_wpt(n); }
size_t msm_direct_workspace_bytes(size_t n) { return align_up(direct_partials(n) * 144, 256) + align_up(((direct_partials(n) + 511) / 512 + 1) * 144, 256); }

// builds pb->direct from the points (device-resident affine bases); returns ZKHIP_ENOMEM when the table does not fit (the caller keeps the bucket path)
int prepare_direct_table(prepared_bases* pb, const uint32_t* d_bases, hipStream_t stream) {
  const size_t n = pb->n;
  void *direct = nullptr, *wt = nullptr, *tmp = nullptr;
  const size_t slice = std::min<size_t>(n, 4096), T = slice * DIRECT_W;
  const size_t wt_bytes = (size_t)DIRECT_W * n * 64, wtmp_pts = align_up((size_t)(DIRECT_W - 1) * n * 144, 256), wtmp_pref = align_up((size_t)(DIRECT_W - 1) * n * 36, 256);
  const size_t tmp_pts = align_up((size_t)(DIRECT_M - 1) * T * 144, 256), tmp_pref = align_up((size_t)(DIRECT_M - 1) * T * 36, 256);
  const size_t tmp_bytes = std::max(wtmp_pts + wtmp_pref, tmp_pts + tmp_pref);
  // (round 4 advice) the table is an accelerator, not a requirement: 256 KiB per point (8 GiB at 2^15) + up to ~3 GiB of build temporaries must not
  // starve the allocations that matter (a k = 22 SRS pair holds 7 GiB of wide tables; the rust shim pins g and g_lagrange of every ParamsKZG,
  // clones included).  It is built only while ALL direct tables of the process stay within $ZKHIP_DIRECT_BUDGET_GIB (default 24) and the table
  // plus its temporaries leave at least half of the currently free HBM; otherwise the set keeps the bucket path (same results).
  {
    static const size_t budget = [] { const char* e = getenv("ZKHIP_DIRECT_BUDGET_GIB"); const long v = e ? atol(e) : 24; return (size_t)(v >= 0 && v <= 4096 ? v : 24) << 30; }();
    size_t free_b = 0, total_b = 0;
    const size_t want = direct_table_bytes(n) + wt_bytes + tmp_bytes;
    if (g_direct_bytes.load() + direct_table_bytes(n) > budget || hipMemGetInfo(&free_b, &total_b) != hipSuccess || want > free_b / 2) {
      (void)hipGetLastError();
      set_error("direct table: %zu bytes not built (budget %zu, in use %zu, free %zu)", direct_table_bytes(n), budget, g_direct_bytes.load(), free_b);
      return ZKHIP_ENOMEM;
    }
  }
  if (hipMalloc(&direct, direct_table_bytes(n)) != hipSuccess) { (void)hipGetLastError(); set_error("direct table: hipMalloc(%zu) failed", direct_table_bytes(n)); return ZKHIP_ENOMEM; }
  if (hipMalloc(&wt, wt_bytes) != hipSuccess || hipMalloc(&tmp, tmp_bytes) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(direct); if (wt) (void)hipFree(wt);
    set_error("direct table: temporary allocation failed");
    return ZKHIP_ENOMEM;
  }
  // the 32 window points 2^(8 w) P_i of every base: the prepared-table builder with 8-bit windows
  hipLaunchKernelGGL(k_build_table, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, d_bases, (uint32_t)n, 8, DIRECT_W, (uint32_t*)wt, (uint32_t*)tmp,
                     (uint32_t*)((char*)tmp + wtmp_pts), 0);
  for (size_t i0 = 0; i0 < n; i0 += slice) {
    const size_t cnt = std::min(slice, n - i0);
    hipLaunchKernelGGL(k_direct_build, dim3((unsigned)((cnt * DIRECT_W + 63) / 64)), dim3(64), 0, stream, (const uint32_t*)wt, (uint32_t)n, (uint32_t)i0, (uint32_t)cnt,
                       (uint32_t*)direct, (uint32_t*)tmp, (uint32_t*)((char*)tmp + tmp_pts));
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  (void)hipFree(wt); (void)hipFree(tmp);
  if (e != hipSuccess) { (void)hipFree(direct); set_error("direct table: %s", hipGetErrorString(e)); return ZKHIP_EHIP; }
  pb->direct = (uint32_t*)direct;
  g_direct_bytes += direct_table_bytes(n);
  return ZKHIP_OK;
}

// d_scalars: n x 8 words; points [off, off + n) of the prepared set; d_out: 24 words.  ws: msm_direct_workspace_bytes(n).
int msm_g1_direct(const uint32_t* d_scalars, size_t n, const prepared_bases* pb, size_t off, uint32_t* d_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!pb || !pb->direct || off + n > pb->n) { set_error("msm_direct: bad range"); return ZKHIP_EINVAL; }
  if (n == 0) { hipLaunchKernelGGL(k_sum_jacobian, dim3(1), dim3(64), 0, stream, (const uint32_t*)nullptr, 0, d_out, 1u); HIPCHK(hipGetLastError()); return ZKHIP_OK; }
  if (ws_bytes < msm_direct_workspace_bytes(n)) { set_error("msm_direct: workspace too small"); return ZKHIP_EINVAL; }
  uint32_t* const partials = (uint32_t*)ws;
  uint32_t* const level1 = (uint32_t*)((char*)ws + align_up(direct_partials(n) * 144, 256));
  prof_begin(stream);
  const int wpt = direct_wpt(n);
  uint32_t m = (uint32_t)direct_partials(n);
  const dim3 grid((m + 127) / 128), block(128);
  const uint32_t* const tab = (const uint32_t*)pb->direct;
  if (wpt == 1) hipLaunchKernelGGL(k_direct_accumulate1, grid, block, 0, stream, d_scalars, (uint32_t)n, (uint32_t)off, tab, partials);
  else if (wpt == 2) hipLaunchKernelGGL(k_direct_accumulate2, grid, block, 0, stream, d_scalars, (uint32_t)n, (uint32_t)off, tab, partials);
  else if (wpt == 4) hipLaunchKernelGGL(k_direct_accumulate4, grid, block, 0, stream, d_scalars, (uint32_t)n, (uint32_t)off, tab, partials);
  else hipLaunchKernelGGL(k_direct_accumulate8, grid, block, 0, stream, d_scalars, (uint32_t)n, (uint32_t)off, tab, partials);
  prof_mark(stream, "direct_accumulate");
  const uint32_t* cur = partials;
  uint32_t* nxt = level1;
  // 512 -> 1 per workgroup (S = 4) while more than 512 points remain; the last launch is one workgroup with S = ceil(m / 128)
  while (m > 512) {
    const uint32_t outs = (m + 511) / 512;
    hipLaunchKernelGGL(k_points_sum_quad, dim3(outs), dim3(512), 0, stream, cur, m, 4u, nxt, (uint32_t*)nullptr);
    // ping-pong inside the workspace: level 1 writes behind the partials, level 2 (<= 512 points) overwrites the head of the partials
    const uint32_t* t = cur; cur = nxt; nxt = (uint32_t*)t;
    m = outs;
  }
  hipLaunchKernelGGL(k_points_sum_quad, dim3(1), dim3(512), 0, stream, cur, m, (m + 127) / 128, (uint32_t*)nullptr, d_out);
  prof_mark(stream, "direct_sum");
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// FFT over G1 points: `best_fft::<G1>` [DEP halo2-axiom arithmetic.rs, FftGroup for the curve group] as `g_to_lagrange` uses it
// (ParamsKZG::setup / from_parts [DEP poly/kzg/commitment.rs], reached from /root/reference/voter/benches/voter_circuit.rs:60;
// SURVEY.md section 8(f) row 4): a[i] <- sum_j omega^(i j) a[j], natural order in and out.  Same radix-2 structure as the scalar
// transform, but a butterfly's twiddle product is a 254-bit scalar multiplication (~254 doublings + ~127 additions), so the kernel
// is pure field arithmetic: one thread per butterfly, one launch per level, points kept as XYZZ (144 B) in a work array between
// levels.  Thread t handles twiddle index j = t / (n/m) of block t % (n/m): for all but the last six levels the 64 lanes of a
// wavefront share j, hence the scalar's bits, and the add-or-not branch of the double-and-add loop is uniform.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ xyzz xyzz_negate(const xyzz& a) {
  xyzz r = a;
  r.Y = fe_norm(fe_neg_red(a.Y, Fq::P6_S1));      // Y N < 5p -> 6p - Y
  return r;
}

// k * B for a canonical 254-bit integer k (8 words), B in XYZZ form; MSB-first double-and-add
__device__ __forceinline__ xyzz xyzz_scalar_mul(const xyzz& B, const uint32_t (&kw)[8]) {
  xyzz acc = xyzz_identity();
  if (xyzz_is_identity(B)) return acc;
  int bit = 253;
  while (bit >= 0 && !((kw[bit >> 5] >> (bit & 31)) & 1)) bit--;
#pragma unroll 1
  for (; bit >= 0; bit--) {
    acc = xyzz_dbl(acc);
    if ((kw[bit >> 5] >> (bit & 31)) & 1) acc = xyzz_add(acc, B);
  }
  return acc;
}

// Fr element in external Montgomery words -> canonical integer words
__device__ __forceinline__ void fr_ext_to_integer(const fe& internal_reduced, uint32_t (&kw)[8]) {
  fe raw1;
#pragma unroll
  for (int i = 0; i < NL; i++) raw1.l[i] = FrParams::RAW_ONE[i];
  fe_pack(fe_canon_lt2p<FrParams>(fe_mul<FrParams>(raw1, internal_reduced)), kw);
}

__device__ __forceinline__ fe fr_ext_to_internal(const uint32_t* ext_words) {   // x * 2^256 -> x * 2^261, reduced
  fe k;
#pragma unroll
  for (int i = 0; i < NL; i++) k.l[i] = FrParams::FROM_EXT[i];
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = ext_words[i];
  return fe_mul<FrParams>(k, fe_unpack<0>(w));
}

__device__ __forceinline__ fe fr_pow_small(fe base, uint32_t e) {
  fe acc = fe_one<FrParams>();
  while (e) {
    if (e & 1) acc = fe_mul<FrParams>(acc, base);
    base = fe_sqr<FrParams>(base);
    e >>= 1;
  }
  return acc;
}

struct fr_words { uint32_t w[8]; };

// work[bitrev(i)] = scale * in[i]   (FORMAT 0: affine 64 B points, 1: Jacobian 96 B points; scale = nullptr-equivalent: has_scale = 0)
template <int FORMAT>
__global__ void __launch_bounds__(64) k_g1fft_load(const uint32_t* __restrict__ in, uint32_t n, int log_n, fr_words scale, int has_scale,
                                                   uint32_t* __restrict__ work) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  xyzz P;
  if (FORMAT == 0) {
    const affine_words pt = load_affine(in, i);
    P = xyzz_identity();
    if (!affine_is_identity(pt)) xyzz_madd(P, fe_from_ext_lazy(pt.x), fe_from_ext_lazy(pt.y));
  } else {
    P = load_jacobian(in + (size_t)i * 24);
  }
  if (has_scale) {
    uint32_t kw[8];
    fr_ext_to_integer(fr_ext_to_internal(scale.w), kw);
    P = xyzz_scalar_mul(P, kw);
  }
  store_xyzz(work, log_n ? (__brev(i) >> (32 - log_n)) : 0u, P);
}

// level s = 1 .. log_n of the decimation-in-time transform on bit-reversed input: m = 2^s, butterflies (i0, i0 + m/2)
__global__ void __launch_bounds__(64) k_g1fft_level(uint32_t* __restrict__ work, uint32_t n, int s, int log_n, fr_words omega) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n / 2) return;
  const uint32_t blocks = n >> s, half = 1u << (s - 1);          // n/m blocks of m points
  const uint32_t j = t / blocks, blk = t - j * blocks;           // lanes of a wavefront share j while blocks >= 64
  const uint32_t i0 = (blk << s) + j, i1 = i0 + half;
  xyzz T = load_xyzz(work, i1);
  if (j) {                                                        // twiddle omega^(j n/m); j = 0: 1
    uint32_t kw[8];
    fr_ext_to_integer(fr_pow_small(fr_ext_to_internal(omega.w), j * blocks), kw);
    T = xyzz_scalar_mul(T, kw);
  }
  const xyzz A = load_xyzz(work, i0);                             // after the multiplication: one point less alive during it
  store_xyzz(work, i0, xyzz_add(A, T));
  store_xyzz(work, i1, xyzz_add(A, xyzz_negate(T)));
}

__global__ void __launch_bounds__(64) k_g1fft_store_jacobian(const uint32_t* __restrict__ work, uint32_t n, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) store_jacobian(load_xyzz(work, i), out + (size_t)i * 24);
}

// XYZZ work array -> affine points, one inversion per thread chunk (as k_fixed_base_mul)
// points per thread (one shared inversion each): 32 when there are enough points to fill the chip, fewer below that -- a commitment batch of a
// few hundred points is then one inversion + ten multiplications deep instead of 32 x that (the inversion is ~50 multiplications' worth since
// it is division steps, fe_inverse.hpp)
constexpr int AFFINE_CHUNK = 32;
static inline uint32_t affine_chunk(size_t n) { uint32_t ch = AFFINE_CHUNK; while (ch > 1 && n / ch < 65536) ch >>= 1; return ch; }
__global__ void __launch_bounds__(64) k_xyzz_to_affine(const uint32_t* __restrict__ work, uint32_t n, uint32_t* __restrict__ out,
                                                       uint32_t* __restrict__ tmp_pref, uint32_t ch) {
  const uint32_t lo = (blockIdx.x * blockDim.x + threadIdx.x) * ch;
  if (lo >= n) return;
  const uint32_t cnt = min(ch, n - lo);
  fe pref = fe_one<Fq>();
  for (uint32_t i = 0; i < cnt; i++) {
    const xyzz Q = load_xyzz(work, lo + i);
    store_fe9_generic(tmp_pref, lo + i, pref);
    if (!xyzz_is_identity(Q)) pref = fe_mul<Fq>(pref, fe_mul<Fq>(Q.ZZ, Q.ZZZ));
  }
  fe inv = fq_inverse(pref);
  for (uint32_t ii = cnt; ii-- > 0;) {
    const xyzz Q = load_xyzz(work, lo + ii);
    uint32_t* o = out + (size_t)(lo + ii) * 16;
    if (xyzz_is_identity(Q)) {
#pragma unroll
      for (int t = 0; t < 16; t++) o[t] = 0;
      continue;
    }
    const fe pre = load_fe9_generic(tmp_pref, lo + ii);
    const fe winv = fe_mul<Fq>(inv, pre);
    inv = fe_mul<Fq>(inv, fe_mul<Fq>(Q.ZZ, Q.ZZZ));
    const fe x = fe_mul<Fq>(fe_mul<Fq>(winv, Q.ZZZ), Q.X);
    const fe y = fe_mul<Fq>(fe_mul<Fq>(winv, Q.ZZ), Q.Y);
    uint32_t wx[8], wy[8];
    fe_to_ext<Fq>(x, wx);
    fe_to_ext<Fq>(y, wy);
#pragma unroll
    for (int t = 0; t < 8; t++) { o[t] = wx[t]; o[8 + t] = wy[t]; }
  }
}

// `Curve::batch_normalize` [DEP group / halo2curves; create_proof normalises its commitments before they enter the transcript]:
// n Jacobian points (96 B) -> n affine points (64 B), identity -> (0, 0); one inversion per chunk of 32 points
__global__ void __launch_bounds__(64) k_jacobian_to_xyzz(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ work) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) store_xyzz(work, i, load_jacobian(in + (size_t)i * 24));
}

size_t g1_batch_normalize_workspace(size_t n) { return align_up(n * 144, 256) + align_up(n * 36, 256) + 256; }

int g1_batch_normalize_device(const uint32_t* d_in, size_t n, uint32_t* d_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (n == 0) return ZKHIP_OK;
  if (n >= (1ull << 31)) { set_error("batch_normalize: n too large"); return ZKHIP_EINVAL; }
  if (ws_bytes < g1_batch_normalize_workspace(n)) { set_error("batch_normalize: workspace too small"); return ZKHIP_EINVAL; }
  uint32_t* work = (uint32_t*)ws;
  uint32_t* pref = (uint32_t*)((char*)ws + align_up(n * 144, 256));
  hipLaunchKernelGGL(k_jacobian_to_xyzz, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, d_in, (uint32_t)n, work);
  { const uint32_t ach = affine_chunk(n); hipLaunchKernelGGL(k_xyzz_to_affine, dim3((unsigned)(((n + ach - 1) / ach + 63) / 64)), dim3(64), 0, stream, work, (uint32_t)n, d_out, pref, ach); }
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}

size_t g1_fft_workspace(size_t n) { return align_up(n * 144, 256) + align_up(n * 36, 256) + 256; }

// in_format / out_format: 0 = affine (64 B), 1 = Jacobian (96 B).  scale_ext: optional factor applied to every input point.
int g1_fft_device(const uint32_t* d_in, int in_format, uint32_t* d_out, int out_format, uint32_t log_n, const uint32_t omega_ext[8],
                  const uint32_t* scale_ext, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (log_n > 26) { set_error("g1_fft: log_n = %u out of range", log_n); return ZKHIP_EINVAL; }
  const size_t n = (size_t)1 << log_n;
  if (ws_bytes < g1_fft_workspace(n)) { set_error("g1_fft: workspace too small"); return ZKHIP_EINVAL; }
  uint32_t* work = (uint32_t*)ws;
  uint32_t* pref = (uint32_t*)((char*)ws + align_up(n * 144, 256));
  fr_words om, sc;
  for (int i = 0; i < 8; i++) { om.w[i] = omega_ext[i]; sc.w[i] = scale_ext ? scale_ext[i] : 0u; }
  const unsigned gb = (unsigned)((n + 63) / 64);
  if (in_format == 0) hipLaunchKernelGGL(k_g1fft_load<0>, dim3(gb), dim3(64), 0, stream, d_in, (uint32_t)n, (int)log_n, sc, scale_ext ? 1 : 0, work);
  else hipLaunchKernelGGL(k_g1fft_load<1>, dim3(gb), dim3(64), 0, stream, d_in, (uint32_t)n, (int)log_n, sc, scale_ext ? 1 : 0, work);
  for (uint32_t s = 1; s <= log_n; s++)
    hipLaunchKernelGGL(k_g1fft_level, dim3((unsigned)((n / 2 + 63) / 64)), dim3(64), 0, stream, work, (uint32_t)n, (int)s, (int)log_n, om);
  if (out_format == 1) hipLaunchKernelGGL(k_g1fft_store_jacobian, dim3(gb), dim3(64), 0, stream, work, (uint32_t)n, d_out);
  else { const uint32_t ach = affine_chunk(n); hipLaunchKernelGGL(k_xyzz_to_affine, dim3((unsigned)(((n + ach - 1) / ach + 63) / 64)), dim3(64), 0, stream, work, (uint32_t)n, d_out, pref, ach); }
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}

int sum_jacobian_device(const uint32_t* d_in, int m, uint32_t* d_out, hipStream_t stream, size_t count) {
  if (count == 0) return ZKHIP_OK;
  hipLaunchKernelGGL(k_sum_jacobian, dim3((uint32_t)count), dim3(64), 0, stream, d_in, m, d_out, (uint32_t)count);
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}

}  // namespace zkhip
