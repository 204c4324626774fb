// Radix-2 NTT over BN254 Fr for MI355X (gfx950).
//
// Replaces halo2-axiom `best_fft` and the `EvaluationDomain` passes built on it [DEP]
// (halo2_proofs/src/arithmetic.rs, poly/domain.rs; reached from /root/reference/aggregator/src/wrapper.rs:129
// `create_proof` and wrapper.rs:107-108 keygen).  Same result a[i] <- sum_j a[j] w^(ij), natural order in and
// out; the reference does bit-reverse + in-place radix-2 butterflies over CPU threads, here:
//
//   * Stockham autosort: pass p with Ns = prod(previous radices), R = 2^B reads  x_r = src[j + r N/R]
//     (contiguous in j), multiplies by w_(Ns R)^(k r), k = j mod Ns, does an R-point DFT and writes
//     dst[(j/Ns) Ns R + k + r Ns].  No bit-reversal pass, every pass streams coalesced runs.
//   * One workgroup owns a tile of J consecutive j (J * R = 1024 elements, 36 KiB of LDS as 9-limb values; up to 2048):
//     cooperative coalesced load -> LDS, rounds of 2 radix-2 stages in registers (4 elements per
//     thread; 3 stages / 8 elements as a variant), LDS exchange between rounds, cooperative coalesced store.
//   * B <= 9 bits per pass: 2^24 is three passes.  Twiddles w_M^(k r) come from an M-entry table (laid out [r][k], the order read) when
//     M <= 2^24 (HBM is plentiful: 36 B * M per (omega, log_n), cached) and from two 2^(log M / 2)-entry
//     tables (one extra multiply) above that.
//   * Values stay lazily reduced inside a pass (fp29.hpp); intermediate passes end with a ~50-instruction
//     quotient-estimate reduction (fe_reduce_soft) so the value fits 32 bytes; the last pass ends with the
//     caller's output scale (ifft divisor, coset powers) as its one multiply, or with a soft reduction and two
//     conditional subtractions when there is no scale.
//   * The external Montgomery-256 words are used as they are: x*2^256 is the radix-2^261 Montgomery form of
//     x*2^-5, and the NTT is linear, so no format conversion multiply is needed on either side.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <vector>
#include <array>
#include "fp29.hpp"
#include "zkhip_internal.hpp"

namespace zkhip {

using Fr = FrParams;
constexpr uint32_t NTT_TILE_MAX = 2048;   // largest tile the LDS layout and the launch bounds are sized for
constexpr int NTT_MAX_BITS = 9;
constexpr uint32_t NTT_DIRECT_TABLE_BITS = 24;   // pass twiddles w_M^t as one M-entry table up to M = 2^24 (604 MB), two-level above

struct scale_arg {      // up to 3 external (Montgomery-256) constants applied as c[i % period]; period 0 = none
  uint32_t w[3][8];
  uint32_t period;
};

struct pass_args {
  const uint32_t* src;
  uint32_t* dst;
  uint32_t src_stride, dst_stride; // batch: elements between consecutive polynomials (blockIdx.y selects the polynomial)
  uint32_t L, S, B;               // log2 N, log2 Ns, log2 R
  const uint32_t* tw_local;       // w_R^x, x < R/2              (internal 9-limb form)
  const uint32_t* tw_lo;          // w_M^t, t < 2^h (or t < M when tw_hi == nullptr)
  const uint32_t* tw_hi;          // w_M^(t 2^h)
  uint32_t h;
  uint32_t tw_rk;                 // direct table in (r, k) layout: w_M^(k r) at index r * Ns + k
  uint32_t first, last;
  uint32_t tile;                  // elements per workgroup (<= NTT_TILE_MAX)
  uint32_t in_len;                // first pass: elements >= in_len read as zero
  uint32_t out_len;               // last pass: elements >= out_len are not stored
  scale_arg in_scale, out_scale;
};

__device__ __forceinline__ fe load_fe9(const uint32_t* p, uint32_t idx) {
  fe r;
  const uint32_t* q = p + (size_t)idx * 9;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = q[i];
  return r;
}
__device__ __forceinline__ void store_fe9(uint32_t* p, uint32_t idx, const fe& a) {
  uint32_t* q = p + (size_t)idx * 9;
#pragma unroll
  for (int i = 0; i < 9; i++) q[i] = a.l[i];
}

// LDS tile: element E lives at word 9 E + 8 (E >> 6) + (E >> 8).  With the plain 9-word stride the first round's accesses
// (element 64 m + 8 e + jj over the lanes' (m, jj)) and the bit-reversed initial stores hit 8 banks out of 64; the two pad terms
// make those patterns (nearly) conflict-free and leave the contiguous ones as they were.
__device__ __forceinline__ uint32_t lds_word(uint32_t e) { return 9u * e + 8u * (e >> 6) + (e >> 8); }
constexpr uint32_t NTT_LDS_WORDS = 9u * NTT_TILE_MAX + 8u * 32u + 8u + 16u;
static inline uint32_t lds_word_host(uint32_t e) { return 9u * e + 8u * (e >> 6) + (e >> 8); }
__device__ __forceinline__ fe load_lds9(const uint32_t* p, uint32_t idx) {
  fe r;
  const uint32_t* q = p + lds_word(idx);
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = q[i];
  return r;
}
__device__ __forceinline__ void store_lds9(uint32_t* p, uint32_t idx, const fe& a) {
  uint32_t* q = p + lds_word(idx);
#pragma unroll
  for (int i = 0; i < 9; i++) q[i] = a.l[i];
}

// external constant c*2^256 -> internal c*2^261 (reduced)
__device__ __forceinline__ fe fr_ext_to_internal(const uint32_t (&w)[8]) {
  fe k;
#pragma unroll
  for (int i = 0; i < NL; i++) k.l[i] = Fr::FROM_EXT[i];
  return fe_mul<Fr>(k, fe_unpack<0>(w));
}

__device__ __forceinline__ fe scale_pick(const scale_arg& s, uint32_t idx) {
  uint32_t sel = idx % 3u;   // period is 1 or 3
  if (s.period == 1) sel = 0;
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = sel == 0 ? s.w[0][i] : (sel == 1 ? s.w[1][i] : s.w[2][i]);
  return fr_ext_to_internal(w);
}

// One element from global memory, ready for the first round: zero-padded / input-scaled on the first pass, multiplied by the pass
// twiddle w_M^(k r) on the later ones (row r, column j of the pass's source matrix: element j + r N/R).
template <bool CH>
__device__ __forceinline__ fe ntt_fetch(const pass_args& a, const uint32_t* __restrict__ src, uint32_t r, uint32_t j, uint32_t NR, uint32_t Ns) {
  const uint32_t g = j + r * NR;
  if (a.first) {
    if (g >= a.in_len) return fe_zero();
    uint32_t w[8];
    load_words(src + (size_t)g * 8, w);
    fe x = fe_unpack<0>(w);
    if (a.in_scale.period) x = fe_mul<Fr, CH>(scale_pick(a.in_scale, g), x);
    return x;
  }
  uint32_t w[8];
  load_words(src + (size_t)g * 8, w);
  const uint32_t t = (j & (Ns - 1)) * r;
  fe tw;
  if (a.tw_hi == nullptr) {
    tw = load_fe9(a.tw_lo, a.tw_rk ? (r << a.S) + (j & (Ns - 1)) : t);
  } else {
    tw = fe_mul<Fr, CH>(load_fe9(a.tw_lo, t & ((1u << a.h) - 1)), load_fe9(a.tw_hi, t >> a.h));
  }
  return fe_mul<Fr, CH>(tw, fe_unpack<0>(w));
}

// One result to global memory: destination index d of (row rp, column j) in the Stockham order; the caller's scale or a soft reduction
// on the last pass, the 32-byte intermediate format otherwise.
template <bool CH>
__device__ __forceinline__ void ntt_emit(const pass_args& a, uint32_t* __restrict__ dst, uint32_t rp, uint32_t j, uint32_t B, uint32_t Ns, fe x) {
  const uint32_t d = ((j >> a.S) << (a.S + B)) + (j & (Ns - 1)) + (rp << a.S);
  uint32_t w[8];
  if (a.last) {
    if (d >= a.out_len) return;
    if (a.out_scale.period) fe_pack(fe_canon_lt2p<Fr>(fe_mul<Fr, CH>(scale_pick(a.out_scale, d), x)), w);
    else fe_pack(fe_canon_lt3p<Fr>(fe_reduce_soft<Fr>(x)), w);   // x is N-form < 29p after the last round
  } else {
    fe_pack(fe_reduce_soft<Fr>(x), w);         // < 2p + 2^233 < 2^256: fits the 32-byte intermediate format
  }
  store_words(dst + (size_t)d * 8, w);
}

// CH: fe_mul with single-chain product columns (see the launch site).
// E = elements a thread holds in a round (log2 E radix-2 stages per LDS round trip).  E = 8: 256 threads per 2048-element tile, 3 stages
// per round, 202 VGPRs -> 2 waves per SIMD.  E = 4 (default): 512 threads per tile, 2 stages per round, 123 VGPRs -> 4 waves per SIMD at
// the price of one more LDS round trip per 8-bit pass (+5.6 % VALU instructions).  Measured equal within noise at 2^22 .. 2^26 (the
// kernel runs at the VALU issue rate of its instruction mix either way: DESIGN.md section 4), E = 4 ahead on small transforms.
template <int E, bool CH>
__global__ void __launch_bounds__(2048 / E) __attribute__((amdgpu_waves_per_eu(E == 8 ? 2 : 4, E == 8 ? 2 : 4))) k_ntt_pass(pass_args a) {
  constexpr int VM = E == 8 ? 3 : 2;
  extern __shared__ uint32_t lds[];
  const uint32_t B = a.B, R = 1u << B, L = a.L;
  const uint32_t N = 1u << L;
  const uint32_t tile = a.tile;
  const uint32_t J = tile >> B;
  const uint32_t logJ = 31 - __builtin_clz(J);
  const uint32_t j0 = blockIdx.x * J;
  const uint32_t NR = N >> B;
  const uint32_t Ns = 1u << a.S;
  const uint32_t nthreads = blockDim.x;
  const uint32_t* const src = a.src + (size_t)blockIdx.y * a.src_stride * 8;
  uint32_t* const dst = a.dst + (size_t)blockIdx.y * a.dst_stride * 8;

  // The first round reads its elements from global memory and the last one writes its results there whenever that keeps the runs a
  // wavefront touches contiguous: a thread (m, jj) holds the rows pos[e] of column jj, lanes run over jj first, so a load covers J
  // consecutive elements of a source row (always the case), and a store J consecutive elements of a destination run when Ns >= J
  // (every pass but the first, whose runs are R elements of ONE column: those are transposed through the LDS tile as before).
  // Against staging both ends through LDS: two of the five LDS round trips and barriers of an 8-bit pass less (2^24: see DESIGN.md section 4).
  const bool direct_store = Ns >= J;

  // ---- rounds of <= 3 in-register stages ----------------------------------------------------------
  const uint32_t jj = threadIdx.x & (J - 1), m = threadIdx.x >> logJ;
  // generic stages u < v of a round that starts at stage s
  auto stages = [&](fe (&x)[E], const uint32_t (&pos)[E], uint32_t s, uint32_t v) {
#pragma unroll
    for (int u = 0; u < VM; u++) {
      if ((uint32_t)u < v) {
        const uint32_t st = s + u;
#pragma unroll
        for (int e = 0; e < E; e++) {
          if ((e >> u) & 1) continue;          // e is the upper element of a pair
          const int f = e | (1 << u);
          const uint32_t lo_i = pos[e] & ((1u << st) - 1);
          fe t = fe_mul<Fr, CH>(load_fe9(a.tw_local, lo_i << (B - 1 - st)), x[f]);   // N x (limbs < 2^31.5)
          x[f] = fe_sub_red(x[e], t, Fr::P3_S1);                                   // x - t + 3p
          x[e] = fe_add(x[e], t);
        }
      }
    }
  };
  auto positions = [&](uint32_t (&pos)[E], uint32_t s, uint32_t v) {
#pragma unroll
    for (int e = 0; e < E; e++) {
      const uint32_t rest = (m << (VM - v)) | ((uint32_t)e >> v);
      const uint32_t lo = rest & ((1u << s) - 1), hi = rest >> s;
      pos[e] = (hi << (s + v)) | (((uint32_t)e & ((1u << v) - 1)) << s) | lo;
    }
  };
  uint32_t s = 0;
  {
    // ---- first round: elements straight from global memory (tile position p holds source row bitrev_B(p): the pass's decimation)
    const uint32_t v = B < (uint32_t)VM ? B : (uint32_t)VM;
    fe x[E];
    uint32_t pos[E];
    positions(pos, 0, v);
    // (a compile-time loop: `#pragma unroll` leaves this one rolled -- the body holds two asm-block multiplications -- and a rolled loop
    // indexes x[] dynamically, i.e. keeps it in scratch)
    static_for<0, E>([&](auto ec) {
      constexpr int e = decltype(ec)::value;
      x[e] = ntt_fetch<CH>(a, src, __brev(pos[e]) >> (32 - B), j0 + jj, NR, Ns);
    });
    if (E == 4 && v == (uint32_t)VM && a.first && a.in_len <= (NR << (B - 2))) {
      // zero-padded input of at most N / 4 elements (coeff_to_extended: n coefficients on the 4n-point coset): only source rows below R / 4
      // are non-zero, and those are the thread's element 0 (rows bitrev(4 m + e): e = 1, 2, 3 select the upper three quarters).  Both
      // stages of the round then add or subtract zeros: all four outputs equal x[0] (the general code below would compute x[0] + 3p,
      // x[0] + 6p and x[0] + 3p - w4 * 0 with one multiplication); the three fetches above returned zero without touching memory.
      x[1] = x[0]; x[2] = x[0]; x[3] = x[0];
    } else if (v == (uint32_t)VM) {
      // v == VM, lo == 0: twiddles depend only on the register index and 7 of the 12 (E = 4: 3 of the 4) are w^0 = 1, so
      // those butterflies need no multiply.  Inputs are N-form < 2p.  Bounds (value / limb) are noted per stage.
      const fe w4 = load_fe9(a.tw_local, 1u << (B - 2));
#pragma unroll
      for (int e = 0; e < E; e += 2) {                    // stage 0: all trivial.  x' < 4p / 2^30, y' < 5p / 1.5*2^30
        fe t = x[e + 1];
        x[e + 1] = fe_sub_red(x[e], t, Fr::P3_S1);
        x[e] = fe_add(x[e], t);
      }
#pragma unroll
      for (int e = 0; e < E; e += 4) {                    // stage 1
        fe t = fe_norm(x[e + 2]);                         // trivial pair (e, e+2): t < 4p, N
        x[e + 2] = fe_sub_red(x[e], t, Fr::P6_S1);        // < 10p / 2^31
        x[e] = fe_add(x[e], t);                           // < 8p / 1.5*2^30
        fe u1 = fe_mul<Fr, CH>(w4, x[e + 3]);                 // pair (e+1, e+3): input < 5p / 1.5*2^30
        x[e + 3] = fe_sub_red(x[e + 1], u1, Fr::P3_S1);   // < 8p / 2.5*2^30
        x[e + 1] = fe_add(x[e + 1], u1);                  // < 7p / 2^31
      }
      if constexpr (E == 8) {                             // stage 2
        const fe w8 = load_fe9(a.tw_local, 1u << (B - 3)), w83 = load_fe9(a.tw_local, 3u << (B - 3));
        fe t = fe_norm(x[4]);                             // trivial pair (0, 4): t < 8p, N
        x[4] = fe_sub_red(x[0], t, Fr::P10_S1);           // < 18p / 2.5*2^30
        x[0] = fe_add(x[0], t);                           // < 16p / 2^31
        fe u1 = fe_mul<Fr, CH>(w8, x[5]);                     // inputs: limbs <= 2.5*2^30 < 2^31.5
        x[5] = fe_sub_red(x[1], u1, Fr::P3_S1);
        x[1] = fe_add(x[1], u1);
        fe u2 = fe_mul<Fr, CH>(w4, x[6]);
        x[6] = fe_sub_red(x[2], u2, Fr::P3_S1);
        x[2] = fe_add(x[2], u2);
        fe u3 = fe_mul<Fr, CH>(w83, x[7]);
        x[7] = fe_sub_red(x[3], u3, Fr::P3_S1);
        x[3] = fe_add(x[3], u3);
      }
    } else {
      stages(x, pos, 0, v);
    }
    if (v == B && direct_store) {
      static_for<0, E>([&](auto ec) { constexpr int e = decltype(ec)::value; ntt_emit<CH>(a, dst, pos[e], j0 + jj, B, Ns, fe_norm(x[e])); });
    } else {
#pragma unroll
      for (int e = 0; e < E; e++) store_lds9(lds, pos[e] * J + jj, fe_norm(x[e]));
      __syncthreads();
    }
    s = v;
  }
  while (s < B) {
    const uint32_t v = (B - s) < (uint32_t)VM ? (B - s) : (uint32_t)VM;
    fe x[E];
    uint32_t pos[E];
    positions(pos, s, v);
#pragma unroll
    for (int e = 0; e < E; e++) x[e] = load_lds9(lds, pos[e] * J + jj);
    stages(x, pos, s, v);
    if (s + v == B && direct_store) {
      static_for<0, E>([&](auto ec) { constexpr int e = decltype(ec)::value; ntt_emit<CH>(a, dst, pos[e], j0 + jj, B, Ns, fe_norm(x[e])); });
    } else {
#pragma unroll
      for (int e = 0; e < E; e++) store_lds9(lds, pos[e] * J + jj, fe_norm(x[e]));
      __syncthreads();
    }
    s += v;
  }
  if (direct_store) return;

  // ---- store through the tile: destination order (a, r', b), b = jj mod q fastest, q = min(Ns, J) ----
  const uint32_t q = Ns < J ? Ns : J;
  const uint32_t logq = 31 - __builtin_clz(q);
  for (uint32_t idx = threadIdx.x; idx < tile; idx += nthreads) {
    const uint32_t b = idx & (q - 1), rp = (idx >> logq) & (R - 1), aa = idx >> (logq + B);
    const uint32_t jj = aa * q + b;
    ntt_emit<CH>(a, dst, rp, j0 + jj, B, Ns, load_lds9(lds, rp * J + jj));
  }
}

// N <= 4: direct DFT by one thread
__global__ void k_ntt_tiny(const uint32_t* src0, uint32_t* dst0, uint32_t L, const uint32_t* pw2, uint32_t in_len,
                           uint32_t out_len, scale_arg in_scale, scale_arg out_scale, uint32_t src_stride, uint32_t dst_stride) {
  if (threadIdx.x != 0) return;
  const uint32_t* src = src0 + (size_t)blockIdx.x * src_stride * 8;
  uint32_t* dst = dst0 + (size_t)blockIdx.x * dst_stride * 8;
  const uint32_t N = 1u << L;
  fe x[4], y[4];
  fe one = fe_one<Fr>();
  for (uint32_t i = 0; i < N; i++) {
    if (i < in_len) {
      uint32_t w[8];
      load_words(src + (size_t)i * 8, w);
      x[i] = fe_unpack<0>(w);
      if (in_scale.period) x[i] = fe_mul<Fr>(scale_pick(in_scale, i), x[i]);
    } else x[i] = fe_zero();
  }
  fe om = load_fe9(pw2, 0);   // omega
  fe wi = one;                // omega^i
  for (uint32_t i = 0; i < N; i++) {
    fe acc = fe_zero(), wij = one;
    for (uint32_t j = 0; j < N; j++) {
      acc = fe_norm(fe_add(acc, fe_mul<Fr>(wij, x[j])));
      wij = fe_mul<Fr>(wij, wi);
    }
    y[i] = acc;
    wi = fe_mul<Fr>(wi, om);
  }
  for (uint32_t i = 0; i < N && i < out_len; i++) {
    fe sc = out_scale.period ? scale_pick(out_scale, i) : one;
    uint32_t w[8];
    fe_pack(fe_canon_lt2p<Fr>(fe_mul<Fr>(sc, y[i])), w);
    store_words(dst + (size_t)i * 8, w);
  }
}

// pw2[e] = omega^(2^e), e = 0..L  (internal form)
__global__ void k_ntt_pow2(const uint32_t* omega_ext_dev, uint32_t L, uint32_t* pw2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = omega_ext_dev[i];
  fe x = fr_ext_to_internal(w);
  for (uint32_t e = 0; e <= L; e++) {
    store_fe9(pw2, e, x);
    x = fe_sqr<Fr>(x);
  }
}

// out[i] = (omega^(2^e0))^i for i < count, by square-and-multiply over the bits of i.  rk_shift != 0: the pass-twiddle layout
// out[r * Ns + k] = (omega^(2^e0))^(k r), Ns = 2^rk_shift -- the order k_ntt_pass reads them in (k runs with the lane index, so the
// 36-byte entries of a wavefront's load are neighbours; in the plain power table they were k r apart, one memory transaction each)
__global__ void __launch_bounds__(256) k_ntt_powers(const uint32_t* pw2, uint32_t e0, uint32_t count, uint32_t* out, uint32_t rk_shift) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  fe acc = fe_one<Fr>();
  uint32_t bits = rk_shift ? (i >> rk_shift) * (i & ((1u << rk_shift) - 1)) : i, b = 0;
  while (bits) {
    if (bits & 1) acc = fe_mul<Fr>(acc, load_fe9(pw2, e0 + b));
    bits >>= 1;
    b++;
  }
  store_fe9(out, i, fe_canon_lt2p<Fr>(acc));
}

// a[i] *= table[i % period]   (table in external form, converted per element: elementwise passes are HBM-bound)
__global__ void __launch_bounds__(256) k_mul_periodic(uint32_t* a, size_t n, const uint32_t* table, uint32_t period) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t w[8], t[8];
    load_words(a + i * 8, w);
    load_words(table + (size_t)(i % period) * 8, t);
    fe r = fe_mul<Fr>(fr_ext_to_internal(t), fe_unpack<0>(w));
    fe_pack(fe_canon_lt2p<Fr>(r), w);
    store_words(a + i * 8, w);
  }
}

// ------------------------------------------------------------------------------------------------
// plans: pass split + twiddle tables, cached per (device, omega, log_n) -- the tables live in the HBM of the device that runs the
// transform (round 5: independent transforms of a batch are spread over the devices of zkhip_init, capi.hip transform fan-out)
// ------------------------------------------------------------------------------------------------
struct ntt_plan {
  int device = 0;
  uint32_t L = 0;
  int npass = 0;
  uint32_t B[4] = {0, 0, 0, 0}, S[4] = {0, 0, 0, 0}, h[4] = {0, 0, 0, 0}, rk[4] = {0, 0, 0, 0};
  uint32_t* pw2 = nullptr;
  uint32_t* tw_local[4] = {nullptr, nullptr, nullptr, nullptr};
  uint32_t* tw_lo[4] = {nullptr, nullptr, nullptr, nullptr};
  uint32_t* tw_hi[4] = {nullptr, nullptr, nullptr, nullptr};
  std::vector<void*> allocs;
  size_t bytes = 0;          // device memory held by the tables
  uint64_t last_use = 0;
  ~ntt_plan() {              // a transform queued on some stream may still read the tables
    if (allocs.empty()) return;
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (cur != device) (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    for (void* d : allocs) (void)hipFree(d);
    if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
  }
};

static std::mutex g_plan_mu;
static std::map<std::array<uint32_t, 10>, std::shared_ptr<ntt_plan>> g_plans;   // a transform in flight keeps its plan alive past an eviction

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s failed: %s", #x, hipGetErrorString(e_)); return ZKHIP_EHIP; } } while (0)

static int plan_alloc(ntt_plan* p, uint32_t** out, size_t bytes) {
  void* d = nullptr;
  if (hipMalloc(&d, bytes) != hipSuccess) { (void)hipGetLastError(); set_error("ntt: hipMalloc(%zu) failed", bytes); return ZKHIP_ENOMEM; }
  p->allocs.push_back(d);
  p->bytes += bytes;
  *out = (uint32_t*)d;
  return ZKHIP_OK;
}

static int build_plan(const uint32_t omega_ext[8], uint32_t L, hipStream_t stream, ntt_plan** out) {
  ntt_plan* p = new ntt_plan();
  *out = p;                  // the caller frees a partly built plan on any error return below
  p->L = L;
  HIPCHK(hipGetDevice(&p->device));
  int rc;
  uint32_t* d_omega = nullptr;
  if ((rc = plan_alloc(p, &d_omega, 32)) != ZKHIP_OK) return rc;
  HIPCHK(hipMemcpyAsync(d_omega, omega_ext, 32, hipMemcpyHostToDevice, stream));
  if ((rc = plan_alloc(p, &p->pw2, (size_t)(L + 1) * 36)) != ZKHIP_OK) return rc;
  hipLaunchKernelGGL(k_ntt_pow2, dim3(1), dim3(64), 0, stream, d_omega, L, p->pw2);
  if (L >= 3) {
    p->npass = (int)((L + NTT_MAX_BITS - 1) / NTT_MAX_BITS);
    uint32_t rem = L, s = 0;
    for (int i = 0; i < p->npass; i++) {
      uint32_t b = (rem + (p->npass - i) - 1) / (p->npass - i);
      p->B[i] = b; p->S[i] = s;
      s += b; rem -= b;
    }
    for (int i = 0; i < p->npass; i++) {
      const uint32_t B = p->B[i], logM = p->S[i] + B;
      // local twiddles w_R^x = omega^(x * N/R), x < R/2
      const uint32_t cnt = 1u << (B - 1);
      if ((rc = plan_alloc(p, &p->tw_local[i], (size_t)cnt * 36)) != ZKHIP_OK) return rc;
      hipLaunchKernelGGL(k_ntt_powers, dim3((cnt + 255) / 256), dim3(256), 0, stream, p->pw2, L - B, cnt, p->tw_local[i], 0u);
      if (i == 0) continue;
      // pass twiddles w_M^t = omega^(t * N/M), t < M
      if (logM <= NTT_DIRECT_TABLE_BITS) {   // direct table: 36 B * M of HBM, saves the table-combining multiply
        const uint32_t M = 1u << logM;
        p->h[i] = 0;
        if ((rc = plan_alloc(p, &p->tw_lo[i], (size_t)M * 36)) != ZKHIP_OK) return rc;
        p->rk[i] = 1;                        // (S[i] >= 1 for every pass but the first)
        hipLaunchKernelGGL(k_ntt_powers, dim3((M + 255) / 256), dim3(256), 0, stream, p->pw2, L - logM, M, p->tw_lo[i], p->S[i]);
      } else {
        const uint32_t h = (logM + 1) / 2, nlo = 1u << h, nhi = 1u << (logM - h);
        p->h[i] = h;
        if ((rc = plan_alloc(p, &p->tw_lo[i], (size_t)nlo * 36)) != ZKHIP_OK) return rc;
        if ((rc = plan_alloc(p, &p->tw_hi[i], (size_t)nhi * 36)) != ZKHIP_OK) return rc;
        hipLaunchKernelGGL(k_ntt_powers, dim3((nlo + 255) / 256), dim3(256), 0, stream, p->pw2, L - logM, nlo, p->tw_lo[i], 0u);
        hipLaunchKernelGGL(k_ntt_powers, dim3((nhi + 255) / 256), dim3(256), 0, stream, p->pw2, L - logM + h, nhi, p->tw_hi[i], 0u);
      }
    }
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(stream));   // tables are shared by later calls on any stream
  return ZKHIP_OK;
}

static int get_plan(const uint32_t omega_ext[8], uint32_t L, hipStream_t stream, std::shared_ptr<ntt_plan>* out) {
  std::array<uint32_t, 10> key;
  for (int i = 0; i < 8; i++) key[i] = omega_ext[i];
  key[8] = L;
  int cur_dev = 0;
  HIPCHK(hipGetDevice(&cur_dev));
  key[9] = (uint32_t)cur_dev;
  std::lock_guard<std::mutex> g(g_plan_mu);
  static uint64_t clock = 0;
  auto it = g_plans.find(key);
  if (it != g_plans.end()) { it->second->last_use = ++clock; *out = it->second; return ZKHIP_OK; }
  // The cache is bounded: a 2^24 plan holds 1.2 GB of twiddles (36 B per entry of two direct tables), and a host that walks through many
  // (omega, log n) pairs must not accumulate them.  Beyond the cap the least recently used plans leave the cache (their memory is
  // freed when the last transform holding them has been enqueued, after a device synchronisation: ~ntt_plan).
  constexpr size_t PLAN_CACHE_BYTES = (size_t)12 << 30;
  constexpr size_t PLAN_CACHE_ENTRIES = 48;
  size_t held = 0;                              // the byte cap is per device (each has its own 288 GB)
  for (auto& kv : g_plans) if (kv.first[9] == key[9]) held += kv.second->bytes;
  const size_t incoming = (size_t)36 << (L > 2 ? L : 2);
  while (!g_plans.empty() && (g_plans.size() >= PLAN_CACHE_ENTRIES || held + incoming > PLAN_CACHE_BYTES)) {
    auto victim = g_plans.end();
    for (auto jt = g_plans.begin(); jt != g_plans.end(); ++jt)
      if ((g_plans.size() >= PLAN_CACHE_ENTRIES || jt->first[9] == key[9]) && (victim == g_plans.end() || jt->second->last_use < victim->second->last_use)) victim = jt;
    if (victim == g_plans.end()) break;
    if (victim->first[9] == key[9]) held -= victim->second->bytes;
    g_plans.erase(victim);
  }
  ntt_plan* raw = nullptr;
  int rc = build_plan(omega_ext, L, stream, &raw);
  std::shared_ptr<ntt_plan> p(raw);          // a partly built plan is freed here on an error return
  if (rc != ZKHIP_OK) return rc;
  p->last_use = ++clock;
  g_plans[key] = p;
  *out = p;
  return ZKHIP_OK;
}

void ntt_clear_cache() {
  std::lock_guard<std::mutex> g(g_plan_mu);
  g_plans.clear();
}

static scale_arg make_scale(const uint32_t* ext, uint32_t period) {
  scale_arg s;
  memset(&s, 0, sizeof(s));
  s.period = period;
  for (uint32_t k = 0; k < period && k < 3; k++) memcpy(s.w[k], ext + 8 * k, 32);
  return s;
}

int ntt_passes(uint32_t L) { return L < 3 ? 0 : (int)((L + NTT_MAX_BITS - 1) / NTT_MAX_BITS); }

// Generic transform of `batch` polynomials (polynomial b at d_in + b * in_stride elements, result at d_out + b * out_stride):
// out[i] = out_scale[i % op] * sum_j (in_scale[j % ip] * in[j]) omega^(ij), j < in_len (zero above), i < out_len.
// d_in may equal d_out.  Scales are host arrays of `period` external-form constants (period 0, 1 or 3).
// tmp0 / tmp1: scratch of batch * 2^L elements each, needed when the transform has >= 2 / >= 3 passes (ntt_passes).
int ntt_transform(const uint32_t* d_in, uint32_t in_len, uint32_t in_stride, uint32_t* d_out, uint32_t out_len, uint32_t out_stride,
                  uint32_t batch, uint32_t L, const uint32_t omega_ext[8], const uint32_t* in_scale, uint32_t in_period,
                  const uint32_t* out_scale, uint32_t out_period, uint32_t* tmp0, uint32_t* tmp1, hipStream_t stream) {
  if (batch == 0) return ZKHIP_OK;
  if (batch > 65535) { set_error("ntt: batch %u > 65535", batch); return ZKHIP_EINVAL; }
  if (L > 28) { set_error("ntt: log_n = %u > 28", L); return ZKHIP_EINVAL; }
  if ((in_period != 0 && in_period != 1 && in_period != 3) || (out_period != 0 && out_period != 1 && out_period != 3)) {
    set_error("ntt: scale period must be 0, 1 or 3");
    return ZKHIP_EINVAL;
  }
  std::shared_ptr<ntt_plan> p;
  int rc = get_plan(omega_ext, L, stream, &p);
  if (rc != ZKHIP_OK) return rc;
  const uint32_t N = 1u << L;
  if (in_len > N) in_len = N;
  if (out_len > N) out_len = N;
  scale_arg is = make_scale(in_scale, in_scale ? in_period : 0), os = make_scale(out_scale, out_scale ? out_period : 0);
  if (L < 3) {
    hipLaunchKernelGGL(k_ntt_tiny, dim3(batch), dim3(64), 0, stream, d_in, d_out, L, p->pw2, in_len, out_len, is, os, in_stride, out_stride);
    HIPCHK(hipGetLastError());
    return ZKHIP_OK;
  }
  {
    static std::mutex attr_mu;
    static bool attr_set_dev[64] = {};
    int cur_dev = 0;
    HIPCHK(hipGetDevice(&cur_dev));
    std::lock_guard<std::mutex> ag(attr_mu);
    if (!attr_set_dev[cur_dev & 63]) {
      HIPCHK(hipFuncSetAttribute((const void*)k_ntt_pass<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, NTT_LDS_WORDS * 4));
      HIPCHK(hipFuncSetAttribute((const void*)k_ntt_pass<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NTT_LDS_WORDS * 4));
      attr_set_dev[cur_dev & 63] = true;
    }
  }
  // Elements per workgroup.  1024 (256 threads, 36 KiB of LDS, four workgroups per CU) against the 2048 the kernel was written for (two
  // workgroups per CU): a barrier or an LDS round trip parks 4 of a SIMD's waves instead of 8, and a pass keeps 128-byte runs on both
  // sides.  Measured on one box, 2048 / 1024 / 512: 2^13 0.0465 / 0.0355 / 0.0349 ms, 2^22 0.586 / 0.557 / 0.590, 2^24 2.341 / 2.301 / 2.452
  // (profiles/r03_ntt_tile_ab.txt).  ZKHIP_NTT_TILE = 512 | 1024 | 2048 is the A/B knob.
  static const uint32_t tile_knob = [] { const char* e = getenv("ZKHIP_NTT_TILE"); const int v = e ? atoi(e) : 0; return (v == 512 || v == 1024 || v == 2048) ? (uint32_t)v : 1024u; }();
  const uint32_t tile = N < tile_knob ? N : tile_knob;
  if ((p->npass >= 2 && !tmp0) || (p->npass >= 3 && !tmp1)) { set_error("ntt: missing scratch buffer"); return ZKHIP_EINVAL; }
  uint32_t* tmp[2] = {tmp0, tmp1};
  prof_begin(stream);
  for (int i = 0; i < p->npass; i++) {
    pass_args a;
    memset(&a, 0, sizeof(a));
    a.L = L; a.S = p->S[i]; a.B = p->B[i];
    a.tw_local = p->tw_local[i]; a.tw_lo = p->tw_lo[i]; a.tw_hi = p->tw_hi[i]; a.h = p->h[i]; a.tw_rk = p->rk[i];
    a.first = i == 0; a.last = i == p->npass - 1;
    a.in_len = in_len; a.out_len = out_len; a.tile = tile;
    a.in_scale = is; a.out_scale = os;
    // buffer chain: in -> tmp0 -> tmp1 -> tmp0 -> ... -> out
    a.src = i == 0 ? d_in : tmp[(i - 1) & 1];
    a.dst = a.last ? d_out : tmp[i & 1];
    a.src_stride = i == 0 ? in_stride : N;
    a.dst_stride = a.last ? out_stride : N;
    // single-chain product columns (fp29.hpp fe_mul<P, CHAIN>: one run of multiply-adds per column instead of the partial sums LLVM's
    // reassociation makes): -3 % at every size with 4 elements per thread (2^24: 2.35 -> 2.27 ms, same box).  The 8-elements-per-thread
    // shape of round 1 (2 waves per SIMD) measured equal within noise in round 2 and is no longer instantiated.
    static const bool chain = getenv("ZKHIP_NTT_PLAIN") == nullptr;       // A/B knob
    const unsigned threads = tile >= 4 ? tile / 4 : 1;
    if (chain) hipLaunchKernelGGL((k_ntt_pass<4, true>), dim3(N / tile, batch), dim3(threads), (size_t)(lds_word_host(tile) + 16) * 4, stream, a);
    else hipLaunchKernelGGL((k_ntt_pass<4, false>), dim3(N / tile, batch), dim3(threads), (size_t)(lds_word_host(tile) + 16) * 4, stream, a);
    prof_mark(stream, i == 0 ? "ntt_pass0" : (i == 1 ? "ntt_pass1" : (i == 2 ? "ntt_pass2" : "ntt_pass3")));
  }
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}


int fr_mul_periodic_device(uint32_t* d_a, size_t n, const uint32_t* d_table_ext, uint32_t period, hipStream_t stream) {
  if (period == 0) { set_error("mul_periodic: period 0"); return ZKHIP_EINVAL; }
  if (n == 0) return ZKHIP_OK;
  size_t blocks = (n + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(k_mul_periodic, dim3((uint32_t)blocks), dim3(256), 0, stream, d_a, n, d_table_ext, period);
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}

}  // namespace zkhip
