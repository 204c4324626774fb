// Row programs over Fr columns: the fused quotient-numerator pass of the prover and every other pointwise pass
// (SURVEY.md section 8(f) rows 1-3).  Restates, for one thread per row, what [DEP] halo2-axiom
// halo2_proofs/src/plonk/evaluation.rs does per row on the CPU (`GraphEvaluator::evaluate` + the permutation / lookup terms of
// `evaluate_h`), reached from create_proof, /root/reference/aggregator/src/wrapper.rs:129.
//
// Execution model: the program (16-byte instructions) is the same for every row, so each instruction is fetched with scalar
// loads and decoded with scalar branches once per wavefront; only operand fetches and field arithmetic are vector work.  The
// register file (ZKHIP_VM_REGS field elements) lives in VGPRs: registers are selected by uniform switches over constant array
// indices, never by dynamic indexing (which would spill the file to scratch memory).
//
// Number forms.  A column / constant word vector x*2^256 mod r is read directly as the radix-2^261 Montgomery form of
// x' = x*2^-5 (fp29.hpp), so loads and stores need no conversion and additions are exact.  A product needs one factor scaled by
// 2^5: for a memory operand that is just the other unpacking shift (fe_unpack<5>, free), for a register it is a repack.
// Register invariant: N-form limbs, value < 2r.  fe_mul(a < 2r, 32b < 64r) < (128/169 + 1) r < 2r; sums and differences are brought
// back under 2r with one conditional subtraction of 2r.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fp29.hpp"
#include "fr_vec.hpp"
#include "zkhip_internal.hpp"

namespace zkhip {

constexpr int VM_MAX_REGS = ZKHIP_VM_REGS;   // kernel variants exist for 6 / 8 / 12 / 16 registers (occupancy 4 / 3 / 2 / 2 waves per SIMD)
constexpr uint32_t POW_LO_BITS = 12;   // omega^row = hi[row >> 12] * lo[row & 4095]

struct vm_launch {
  const void* prog;             // n_insns x vm_uop (64 bytes: the instruction with its operands' addresses resolved by the host)
  uint32_t n_insns;
  uint32_t result_reg;
  const uint32_t* const* cols;  // n_columns device pointers
  const uint32_t* consts;       // n_constants x 8 words
  const uint32_t* rot_off;      // per rotation slot: (rotation * rot_scale) mod rows
  const uint32_t* pow_lo;       // omega^j, j < 2^POW_LO_BITS (nullptr: ROWPOW unused)
  const uint32_t* pow_hi;       // omega^(j << POW_LO_BITS)
  uint32_t* out;
  uint64_t rows;
  uint32_t accumulate;
  const struct vm_part* parts;  // nullptr, or one record per blockIdx.y: several programs over the same columns in ONE launch (row_vm_device_multi)
};
// what differs between the programs of a multi-program launch (the column table, the power tables and the rows are shared)
struct vm_part {
  const void* prog;
  const uint32_t* consts;
  const uint32_t* rot_off;
  uint32_t* out;
  uint32_t n_insns, result_reg;
};
static_assert(sizeof(vm_part) == 40, "the kernel reads a part record as five 64-bit words");

// 2r as normalised limbs
struct fr_two_p {
  uint32_t l[NL];
  constexpr fr_two_p() : l{} {
    uint32_t carry = 0;
    for (int i = 0; i < NL; i++) {
      const uint32_t v = (Fr::P[i] << 1) | carry;
      l[i] = i < NL - 1 ? (v & LMASK) : v;
      carry = i < NL - 1 ? (Fr::P[i] >> (LB - 1)) : 0;
    }
  }
};
__device__ constexpr fr_two_p FR_2P{};

// N-form value < 4r -> N-form value < 2r, same residue
__device__ __forceinline__ fe vm_condsub_2p(const fe& s) {
  fe d;
  int32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const int32_t t = (int32_t)s.l[i] - (int32_t)FR_2P.l[i] + borrow;
    borrow = t >> 31;
    d.l[i] = i < NL - 1 ? ((uint32_t)t & LMASK) : (uint32_t)t;
  }
  fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = borrow ? s.l[i] : d.l[i];
  return r;
}
__device__ __forceinline__ fe vm_add(const fe& a, const fe& b) { return vm_condsub_2p(fe_norm(fe_add(a, b))); }
// a - b + 2r in (0, 4r), carried with signed limbs
__device__ __forceinline__ fe vm_sub(const fe& a, const fe& b) {
  fe s;
  int32_t carry = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const int32_t t = (int32_t)a.l[i] - (int32_t)b.l[i] + (int32_t)FR_2P.l[i] + carry;
    if (i < NL - 1) { s.l[i] = (uint32_t)t & LMASK; carry = t >> LB; }
    else s.l[i] = (uint32_t)t;
  }
  return vm_condsub_2p(s);
}
__device__ __forceinline__ fe vm_times32(const fe& a) {   // a < 2r < 2^255: repack with the other shift
  uint32_t w[8];
  fe_pack(a, w);
  return fe_unpack<5>(w);
}

#define VM_REG_CASES(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
static_assert(VM_MAX_REGS == 16, "VM_REG_CASES lists the registers");

// Register selection is spelled out with distinct inline-asm markers per case: without them LLVM sinks the
// `r[k] = v` stores into one store through a phi of addresses, which keeps the whole file in scratch memory.
template <int R>
__device__ __forceinline__ fe vm_reg_get(const fe (&r)[R], uint32_t i) {
  fe v = r[0];
  switch (i) {
#define X(k) case k: if constexpr (k < R) { asm volatile("; vm get r" #k); v = r[k]; } break;
    VM_REG_CASES(X)
#undef X
    default: break;
  }
  return v;
}
template <int R>
__device__ __forceinline__ void vm_reg_set(fe (&r)[R], uint32_t i, const fe& v) {
  switch (i) {
#define X(k) case k: if constexpr (k < R) { r[k] = v; asm volatile("; vm set r" #k ::: "memory"); } break;
    VM_REG_CASES(X)
#undef X
    default: break;
  }
}

// Memory operands (a column at a rotation, a constant) are loaded ONE INSTRUCTION AHEAD: the loop decodes instruction pc + 1 before it
// executes instruction pc and issues the loads of its first two operands into a second word buffer, so a load's latency runs under the
// previous instruction's arithmetic (a field multiplication is ~250 wave-instructions).  Before: every instruction loaded its operands and
// waited -- SQ counters on the wrapper quotient: 48.9 % of the wave cycles parked at s_waitcnt (profiles/r03_rowvm_prefetch_ab.txt).
__device__ __forceinline__ bool vm_is_mem(uint32_t opnd) { const uint32_t kind = opnd & 0xff; return kind == ZKHIP_SRC_COLUMN || kind == ZKHIP_SRC_CONST; }
__device__ __forceinline__ bool vm_uses_b(uint32_t op) { return op == ZKHIP_OP_MUL || op == ZKHIP_OP_MAD || op == ZKHIP_OP_ADD || op == ZKHIP_OP_SUB; }

// The program, the rotation offsets, the column pointers and the constants are the same for every lane and are not written by the kernel:
// they are read through the CONSTANT address space, i.e. with scalar loads (s_load_*) into SGPRs.  Read as ordinary global memory the
// compiler must assume that the kernel's own stores alias them, and every instruction fetch / pointer fetch became a VECTOR load followed by
// s_waitcnt vmcnt(0) -- which also drains every operand load in flight, so nothing could be prefetched (ISA of round 2's kernel).
#if defined(__HIP_DEVICE_COMPILE__)
#define ZK_CONSTANT_AS __attribute__((address_space(4)))
#define ZK_GLOBAL_AS __attribute__((address_space(1)))
#else        // (the host pass only parses the kernels)
#define ZK_CONSTANT_AS
#define ZK_GLOBAL_AS
#endif
typedef const ZK_CONSTANT_AS uint32_t* vm_c32;
typedef const ZK_CONSTANT_AS uint64_t* vm_c64;
struct vm_u128 { uint32_t x, y, z, w; };                      // (a plain struct: HIP's uint4 has no constructor from another address space)
typedef const ZK_CONSTANT_AS vm_u128* vm_c128;
typedef const ZK_GLOBAL_AS vm_u128* vm_g128;

// the 8 words of a memory operand.  Branch-free: the address is built with scalar selects -- a column's base pointer and rotation offset, a
// constant's address, or (register / PREV / ROWPOW operands, whose words are never looked at) the start of the constants -- and the two
// 16-byte loads are issued UNCONDITIONALLY.  With loads behind uniform branches the compiler placed s_waitcnt vmcnt(0) at the joins and at
// the loop header, which serialised every prefetch; straight-line loads let it count (vmcnt(4): the four loads of the next instruction stay
// in flight while this one's words are used).  A dummy load is one cache line for the whole wavefront.
__device__ __forceinline__ void vm_load_words(const vm_launch& L, uint32_t opnd, uint64_t row, uint32_t (&w)[8]) {
  const uint32_t kind = opnd & 0xff, rot = (opnd >> 8) & 0xff, index = opnd >> 16;
  const bool is_col = kind == ZKHIP_SRC_COLUMN, is_const = kind == ZKHIP_SRC_CONST;
  const uint64_t col_base = ((vm_c64)L.cols)[is_col ? index : 0u];
  const uint32_t off = ((vm_c32)L.rot_off)[is_col ? rot : 0u];
  const uint64_t base = is_col ? col_base : (uint64_t)(L.consts + (size_t)(is_const ? index : 0u) * 8);
  const uint64_t rr = (row + off) & (is_col ? L.rows - 1 : 0ull);
  const vm_g128 p = (vm_g128)(base + rr * 32);
  const vm_u128 lo = p[0], hi = p[1];
  w[0] = lo.x; w[1] = lo.y; w[2] = lo.z; w[3] = lo.w; w[4] = hi.x; w[5] = hi.y; w[6] = hi.z; w[7] = hi.w;
}

// operand value from its prefetched words (memory kinds) or from the register file / prev / omega^row; TIMES32: scaled by 2^5
// out[row] as it was (PREV) and omega^row (ROWPOW) are per-row values a program reads a handful of times: they live in LDS (limb-major,
// one word per lane and limb: conflict-free), not in 18 VGPRs next to the register file and the prefetch buffers
constexpr uint32_t VM_THREADS = 256;
__device__ __forceinline__ fe vm_lds_get(const uint32_t* __restrict__ base) {
  fe v;
#pragma unroll
  for (int i = 0; i < NL; i++) v.l[i] = base[i * VM_THREADS + threadIdx.x];
  return v;
}
__device__ __forceinline__ void vm_lds_put(uint32_t* __restrict__ base, const fe& v) {
#pragma unroll
  for (int i = 0; i < NL; i++) base[i * VM_THREADS + threadIdx.x] = v.l[i];
}

template <bool TIMES32, int R>
__device__ __forceinline__ fe vm_operand(uint32_t opnd, const uint32_t (&w)[8], const fe (&r)[R], const uint32_t* s_prev, const uint32_t* s_pow) {
  const uint32_t kind = opnd & 0xff, index = opnd >> 16;
  if (kind == ZKHIP_SRC_COLUMN || kind == ZKHIP_SRC_CONST) return TIMES32 ? fe_unpack<5>(w) : fe_unpack<0>(w);
  fe v = kind == ZKHIP_SRC_REG ? vm_reg_get(r, index) : vm_lds_get(kind == ZKHIP_SRC_PREV ? s_prev : s_pow);
  return TIMES32 ? vm_times32(v) : v;
}

// one instruction: operands a / b come with their words already loaded (wa / wb); a memory operand c (MAD's addend) is loaded here
template <int R>
__device__ __forceinline__ void vm_execute(const vm_launch& L, uint32_t head, uint32_t oa, uint32_t ob, uint32_t oc, const uint32_t (&wa)[8], const uint32_t (&wb)[8],
                                           uint64_t row, fe (&r)[R], const uint32_t* s_prev, const uint32_t* s_pow) {
  const uint32_t op = head & 0xff, dst = (head >> 8) & 0xff;
  const fe a = vm_operand<false, R>(oa, wa, r, s_prev, s_pow);
  fe t;
  if (op == ZKHIP_OP_MUL || op == ZKHIP_OP_SQR || op == ZKHIP_OP_MAD) {
    const fe b32 = op == ZKHIP_OP_SQR ? vm_times32(a) : vm_operand<true, R>(ob, wb, r, s_prev, s_pow);
    t = fe_mul<Fr, true>(a, b32);      // single-chain product columns (fp29.hpp): -2 % on the wrapper quotient, same box
    if (op == ZKHIP_OP_MAD) {
      uint32_t wc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (vm_is_mem(oc)) vm_load_words(L, oc, row, wc);
      t = vm_add(t, vm_operand<false, R>(oc, wc, r, s_prev, s_pow));
    }
  } else if (op == ZKHIP_OP_ADD) {
    t = vm_add(a, vm_operand<false, R>(ob, wb, r, s_prev, s_pow));
  } else if (op == ZKHIP_OP_SUB) {
    t = vm_sub(a, vm_operand<false, R>(ob, wb, r, s_prev, s_pow));
  } else if (op == ZKHIP_OP_NEG) {
    t = vm_sub(fe_zero(), a);
  } else if (op == ZKHIP_OP_DBL) {
    t = vm_add(a, a);
  } else {
    t = a;   // MOV
  }
  vm_reg_set(r, dst, t);
}

// One instruction as the kernel reads it: the 16-byte instruction of the ABI plus, for its operands a and b, everything the address of their
// words needs -- base (a column's pointer, a constant's address, or a dummy for operands that are not in memory), rotation offset in rows and
// a row mask (all ones for a column, zero otherwise).  Resolved on the host (row_vm_device), so a fetch is ONE scalar load of 64 bytes and the
// operand loads depend on nothing else; the loop fetches two instructions ahead and loads operands one instruction ahead.
struct vm_uop {
  uint32_t head, oa, ob, oc;
  uint64_t base_a, base_b;
  uint32_t off_a, off_b, mask_a, mask_b;
  uint32_t pad[4];
};
static_assert(sizeof(vm_uop) == 64, "micro-op layout");
struct vm_decoded { uint32_t head, oa, ob, oc; uint64_t base_a, base_b; uint32_t off_a, off_b, mask_a, mask_b; };
__device__ __forceinline__ vm_decoded vm_decode(const vm_launch& L, uint32_t pc) {
  const vm_c128 q = (vm_c128)L.prog + (size_t)pc * 4;          // same instruction for every lane: scalar loads, decoded on the scalar unit
  const vm_u128 q0 = q[0], q1 = q[1], q2 = q[2];
  vm_decoded d;
  d.head = q0.x; d.oa = q0.y; d.ob = q0.z; d.oc = q0.w;
  d.base_a = (uint64_t)q1.x | ((uint64_t)q1.y << 32); d.base_b = (uint64_t)q1.z | ((uint64_t)q1.w << 32);
  d.off_a = q2.x; d.off_b = q2.y; d.mask_a = q2.z; d.mask_b = q2.w;
  return d;
}
__device__ __forceinline__ void vm_load_resolved(uint64_t base, uint32_t off, uint32_t mask, uint64_t row, uint64_t rows, uint32_t (&w)[8]) {
  const uint64_t rr = (row + off) & (rows - 1) & (uint64_t)(int64_t)(int32_t)mask;      // mask = 0: the base itself (a constant / a dummy)
  const vm_g128 p = (vm_g128)(base + rr * 32);
  const vm_u128 lo = p[0], hi = p[1];
  w[0] = lo.x; w[1] = lo.y; w[2] = lo.z; w[3] = lo.w; w[4] = hi.x; w[5] = hi.y; w[6] = hi.z; w[7] = hi.w;
}
// issue the loads of an instruction's operands a / b -- unconditionally (see vm_load_words)
__device__ __forceinline__ void vm_prefetch(const vm_launch& L, const vm_decoded& d, uint64_t row, uint32_t (&wa)[8], uint32_t (&wb)[8]) {
  vm_load_resolved(d.base_a, d.off_a, d.mask_a, row, L.rows, wa);
  vm_load_resolved(d.base_b, d.off_b, d.mask_b, row, L.rows, wb);
}

template <int R>
__global__ void __launch_bounds__(VM_THREADS) k_row_vm(const vm_launch L0) {
  __shared__ uint32_t s_prev[NL * VM_THREADS], s_pow[NL * VM_THREADS];
  vm_launch L = L0;
  if (L0.parts) {                                          // uniform: the record of this workgroup's program, through scalar loads
    const vm_c64 q = (vm_c64)(L0.parts + blockIdx.y);
    const uint64_t w4 = q[4];
    L.prog = (const void*)q[0];
    L.consts = (const uint32_t*)q[1];
    L.rot_off = (const uint32_t*)q[2];
    L.out = (uint32_t*)q[3];
    L.n_insns = (uint32_t)w4;
    L.result_reg = (uint32_t)(w4 >> 32);
  }
  const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= L.rows) return;
  fe r[R];
#pragma unroll
  for (int i = 0; i < R; i++) r[i] = fe_zero();
  vm_lds_put(s_prev, L.accumulate ? load_ext(L.out, row) : fe_zero());      // (a lane reads back only what it wrote: no barrier)
  if (L.pow_lo) {
    uint32_t w[8];
    load_words(L.pow_lo + (row & ((1u << POW_LO_BITS) - 1)) * 8, w);
    vm_lds_put(s_pow, fe_mul<Fr>(load_ext(L.pow_hi, row >> POW_LO_BITS), fe_unpack<5>(w)));
  } else {
    vm_lds_put(s_pow, fe_zero());
  }
  if constexpr (R > 12) {
    // the 16-register file leaves no room for a second buffer pair at 2 waves per SIMD: operands are loaded when the instruction runs
    uint32_t wa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (uint32_t pc = 0; pc < L.n_insns; pc++) {
      const vm_decoded d = vm_decode(L, pc);
      vm_prefetch(L, d, row, wa, wb);
      vm_execute<R>(L, d.head, d.oa, d.ob, d.oc, wa, wb, row, r, s_prev, s_pow);
    }
    uint32_t w[8];
    fe_pack(fe_canon_lt2p<Fr>(vm_reg_get(r, L.result_reg)), w);
    store_words(L.out + row * 8, w);
    return;
  }
  // Software pipeline, two word-buffer pairs alternated by a loop unrolled twice (no register copies):
  //     [fetch instruction pc + 2]  ->  [issue the operand loads of pc + 1]  ->  [execute pc]
  // The program is padded with no-op micro-ops (row_vm_device), so the fetches and loads past the end are harmless and the loop body has no
  // conditions besides its exit.  (Loads two instructions ahead -- a third buffer pair, 127 VGPRs -- measured equal: 13.24-13.34 vs
  // 13.33-13.65 ms on the wrapper quotient; the cycles still parked, 19 %, are not operand loads.)
  uint32_t wa0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wb0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wa1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, wb1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  vm_decoded cur = vm_decode(L, 0), nxt = vm_decode(L, 1);
  vm_prefetch(L, cur, row, wa0, wb0);
#pragma unroll 1
  for (uint32_t pc = 0; pc < L.n_insns; pc += 2) {
    const vm_decoded nn = vm_decode(L, pc + 2);
    vm_prefetch(L, nxt, row, wa1, wb1);
    vm_execute<R>(L, cur.head, cur.oa, cur.ob, cur.oc, wa0, wb0, row, r, s_prev, s_pow);
    if (pc + 1 >= L.n_insns) break;
    cur = nn;
    const vm_decoded n3 = vm_decode(L, pc + 3);
    vm_prefetch(L, cur, row, wa0, wb0);
    vm_execute<R>(L, nxt.head, nxt.oa, nxt.ob, nxt.oc, wa1, wb1, row, r, s_prev, s_pow);
    nxt = n3;
  }
  uint32_t w[8];
  fe_pack(fe_canon_lt2p<Fr>(vm_reg_get(r, L.result_reg)), w);
  store_words(L.out + row * 8, w);
}

// table[i] = omega^(i << shift), i < count, external words
__global__ void __launch_bounds__(256) k_vm_pow_table(const fe_arg* __restrict__ omega_dev, uint32_t shift, uint32_t count, uint32_t* __restrict__ table) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const fe_arg om = *omega_dev;
  fe base = fr_const_internal(om);
  for (uint32_t s = 0; s < shift; s++) base = fe_sqr<Fr>(base);
  const fe v = fr_pow_u32(base, i);                               // internal form, < 2r
  fe k;
#pragma unroll
  for (int j = 0; j < NL; j++) k.l[j] = Fr::TO_EXT[j];
  uint32_t w[8];
  fe_pack(fe_canon_lt2p<Fr>(fe_mul<Fr>(k, v)), w);
  store_words(table + (size_t)i * 8, w);
}

// out[i] = a[i] * b[i]
__global__ void __launch_bounds__(256) k_fr_pointwise_mul(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, size_t n, uint32_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t wb[8];
  load_words(b + i * 8, wb);
  const fe p = fe_mul<Fr>(load_ext(a, i), fe_unpack<5>(wb));
  uint32_t w[8];
  fe_pack(fe_canon_lt2p<Fr>(p), w);
  store_words(out + i * 8, w);
}

// out[j] = a[ia[j]] * b[ib[j]]: the inner loop of `permutation::keygen::Assembly::build_pk` [DEP] -- sigma_i[j] = delta^(column of the
// cell (i, j) maps to) * omega^(its row) -- as a gather.  Indices are reduced modulo the table lengths, so a bad index cannot fault.
__global__ void __launch_bounds__(256) k_fr_gather_mul(const uint32_t* __restrict__ a, uint32_t a_len, const uint32_t* __restrict__ ia,
                                                       const uint32_t* __restrict__ b, uint32_t b_len, const uint32_t* __restrict__ ib, size_t n,
                                                       uint32_t* __restrict__ out) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  uint32_t wb[8];
  load_words(b + (size_t)(ib[j] % b_len) * 8, wb);
  const fe p = fe_mul<Fr>(load_ext(a, ia[j] % a_len), fe_unpack<5>(wb));
  uint32_t w[8];
  fe_pack(fe_canon_lt2p<Fr>(p), w);
  store_words(out + j * 8, w);
}

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s failed: %s", #x, hipGetErrorString(e_)); return ZKHIP_EHIP; } } while (0)

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static bool fr_is_canonical(const uint64_t* w) {
  static const uint64_t R[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
  for (int i = 3; i >= 0; i--) {
    if (w[i] < R[i]) return true;
    if (w[i] > R[i]) return false;
  }
  return false;
}

int row_vm_validate(const zkhip_vm_program* p, uint32_t n_columns, uint32_t log_rows, int accumulate) {
  (void)accumulate;
  if (!p) { set_error("eval_rows: null program"); return ZKHIP_EINVAL; }
  if (log_rows > 28) { set_error("eval_rows: log_rows %u > 28", log_rows); return ZKHIP_EINVAL; }
  if (p->n_insns == 0 || !p->insns) { set_error("eval_rows: empty program"); return ZKHIP_EINVAL; }
  if (p->n_insns > (1u << 20)) { set_error("eval_rows: program too long (%u instructions)", p->n_insns); return ZKHIP_EINVAL; }
  if (p->result_reg >= (uint32_t)VM_MAX_REGS) { set_error("eval_rows: result register %u out of range", p->result_reg); return ZKHIP_EINVAL; }
  if ((p->n_constants && !p->constants) || (p->n_rotations && !p->rotations)) { set_error("eval_rows: null table"); return ZKHIP_EINVAL; }
  if (p->n_constants > 65536 || n_columns > 65536 || p->n_rotations > 256) { set_error("eval_rows: table too large"); return ZKHIP_EINVAL; }
  for (uint32_t i = 0; i < p->n_constants; i++)
    if (!fr_is_canonical(p->constants + (size_t)i * 4)) { set_error("eval_rows: constant %u is not a canonical Fr", i); return ZKHIP_EINVAL; }
  if (p->omega && !fr_is_canonical(p->omega)) { set_error("eval_rows: omega is not a canonical Fr"); return ZKHIP_EINVAL; }
  for (uint32_t pc = 0; pc < p->n_insns; pc++) {
    const zkhip_vm_insn& in = p->insns[pc];
    if (in.op > ZKHIP_OP_MAD) { set_error("eval_rows: instruction %u: unknown op %u", pc, in.op); return ZKHIP_EINVAL; }
    if (in.dst >= VM_MAX_REGS) { set_error("eval_rows: instruction %u: destination register %u out of range", pc, in.dst); return ZKHIP_EINVAL; }
    const int n_opnd = (in.op == ZKHIP_OP_MAD) ? 3 : ((in.op == ZKHIP_OP_ADD || in.op == ZKHIP_OP_SUB || in.op == ZKHIP_OP_MUL) ? 2 : 1);
    const zkhip_vm_operand* o[3] = {&in.a, &in.b, &in.c};
    for (int k = 0; k < n_opnd; k++) {
      bool ok = true;
      switch (o[k]->kind) {
        case ZKHIP_SRC_CONST: ok = o[k]->index < p->n_constants; break;
        case ZKHIP_SRC_REG: ok = o[k]->index < (uint32_t)VM_MAX_REGS; break;
        case ZKHIP_SRC_COLUMN: ok = o[k]->index < n_columns && o[k]->rot < p->n_rotations; break;
        case ZKHIP_SRC_PREV: break;
        case ZKHIP_SRC_ROWPOW: ok = p->omega != nullptr; break;
        default: ok = false;
      }
      if (!ok) { set_error("eval_rows: instruction %u: operand %d (kind %u, index %u, rot %u) out of range", pc, k, o[k]->kind, o[k]->index, o[k]->rot); return ZKHIP_EINVAL; }
    }
  }
  return ZKHIP_OK;
}

size_t row_vm_workspace_bytes(const zkhip_vm_program* p, uint32_t n_columns, uint32_t log_rows) {
  const size_t rows = (size_t)1 << log_rows;
  return align256(((size_t)p->n_insns + 4) * sizeof(vm_uop)) + align256((size_t)p->n_constants * 32 + 32) + align256((size_t)p->n_rotations * 4 + 4) +
         align256((size_t)n_columns * 8 + 8) + 256 + align256(((size_t)1 << POW_LO_BITS) * 32) + align256(((rows >> POW_LO_BITS) + 1) * 32);
}

// the program, its tables and the column pointers are staged into `ws` (device), then one launch
void vm_staging::release() {
  for (int i = 0; i < 2; i++) {
    if (copied[i]) { (void)hipEventSynchronize(copied[i]); (void)hipEventDestroy(copied[i]); copied[i] = nullptr; }
    if (host[i]) { (void)hipHostFree(host[i]); host[i] = nullptr; cap[i] = 0; }
  }
}

// One program's region of a staged blob -- [micro-ops (+ 4 padding no-ops) | constants | rotation offsets], `region` on the host, `dev`
// its device address, the constants at `o_const` and the rotation offsets at `o_rot` from its start.
// micro-ops: the instruction + its operands' resolved addresses; a product's second factor is fetched scaled by 2^5 -- free for a column /
// constant (the other unpacking shift), a repack for a register -- so the memory operand goes second where the host did not put it there
static void vm_fill_region(const zkhip_vm_program* p, const void* const* d_columns, uint64_t rows, unsigned char* region, uint64_t dev, size_t o_const, size_t o_rot) {
  static_assert(sizeof(zkhip_vm_insn) == 16, "instruction layout");
  std::vector<uint32_t> rot_rows(p->n_rotations ? p->n_rotations : 1, 0u);
  for (uint32_t i = 0; i < p->n_rotations; i++) {
    const int64_t off = ((int64_t)p->rotations[i] * (int64_t)p->rot_scale) % (int64_t)rows;
    rot_rows[i] = (uint32_t)(off < 0 ? off + (int64_t)rows : off);
  }
  const uint64_t d_consts = dev + o_const;
  vm_uop* uops = (vm_uop*)region;
  for (uint32_t pc = 0; pc < p->n_insns + 4; pc++) {
    vm_uop& u = uops[pc];
    std::memset(&u, 0, sizeof(u));
    u.base_a = u.base_b = d_consts;                      // dummy address for operands that are not in memory (and for the padding no-ops)
    if (pc >= p->n_insns) { u.head = ZKHIP_OP_MOV | ((uint32_t)0 << 8); u.oa = ZKHIP_SRC_REG; continue; }   // MOV r0 <- r0 (never executed)
    zkhip_vm_insn in = p->insns[pc];
    const bool a_mem = in.a.kind == ZKHIP_SRC_COLUMN || in.a.kind == ZKHIP_SRC_CONST;
    const bool b_mem = in.b.kind == ZKHIP_SRC_COLUMN || in.b.kind == ZKHIP_SRC_CONST;
    if ((in.op == ZKHIP_OP_MUL || in.op == ZKHIP_OP_MAD) && a_mem && !b_mem) { const zkhip_vm_operand t = in.a; in.a = in.b; in.b = t; }
    std::memcpy(&u, &in, 16);
    const bool uses_b = in.op == ZKHIP_OP_MUL || in.op == ZKHIP_OP_MAD || in.op == ZKHIP_OP_ADD || in.op == ZKHIP_OP_SUB;
    auto resolve = [&](const zkhip_vm_operand& o, bool used, uint64_t* base, uint32_t* off, uint32_t* mask) {
      if (!used) return;
      if (o.kind == ZKHIP_SRC_COLUMN) { *base = (uint64_t)d_columns[o.index]; *off = rot_rows[o.rot]; *mask = 0xffffffffu; }
      else if (o.kind == ZKHIP_SRC_CONST) { *base = d_consts + (uint64_t)o.index * 32; }
    };
    resolve(in.a, true, &u.base_a, &u.off_a, &u.mask_a);
    resolve(in.b, uses_b, &u.base_b, &u.off_b, &u.mask_b);
  }
  if (p->n_constants) std::memcpy(region + o_const, p->constants, (size_t)p->n_constants * 32);
  for (uint32_t i = 0; i < p->n_rotations; i++) std::memcpy(region + o_rot + (size_t)i * 4, &rot_rows[i], 4);      // (a MAD's column addend still goes through the tables)
}

// highest register a program names (the kernel variant is sized for it)
static uint32_t vm_top_register(const zkhip_vm_program* p) {
  uint32_t top = p->result_reg;
  for (uint32_t pc = 0; pc < p->n_insns; pc++) {
    const zkhip_vm_insn& in = p->insns[pc];
    if (in.dst > top) top = in.dst;
    const zkhip_vm_operand* o[3] = {&in.a, &in.b, &in.c};
    for (int k = 0; k < 3; k++) if (o[k]->kind == ZKHIP_SRC_REG && o[k]->index < (uint32_t)VM_MAX_REGS && o[k]->index > top) top = o[k]->index;
  }
  return top;
}

static int vm_launch_kernel(const vm_launch& L, uint32_t top, uint32_t n_parts, hipStream_t stream) {
  const dim3 grid((unsigned)((L.rows + VM_THREADS - 1) / VM_THREADS), n_parts);
  if (top < 6) hipLaunchKernelGGL(k_row_vm<6>, grid, dim3(VM_THREADS), 0, stream, L);
  else if (top < 8) hipLaunchKernelGGL(k_row_vm<8>, grid, dim3(VM_THREADS), 0, stream, L);
  else if (top < 12) hipLaunchKernelGGL(k_row_vm<12>, grid, dim3(VM_THREADS), 0, stream, L);
  else hipLaunchKernelGGL(k_row_vm<16>, grid, dim3(VM_THREADS), 0, stream, L);
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}

// `n_progs` programs over the same columns in ONE launch (blockIdx.y = program), program p writing d_outs[p]: what
// zkhip_fr_eval_rows_sum_device runs its parts with.  Separate launches on separate streams overlap four at a time (the hardware queues);
// one grid has no such limit.  Blob: [columns | omega | part records | region of program 0 | region of program 1 | ...], power tables behind.
// Programs that read ZKHIP_SRC_ROWPOW must agree on omega.  Interpreter only (these are long programs over few rows).
size_t row_vm_multi_workspace_bytes(const zkhip_vm_program* progs, uint32_t n_progs, uint32_t n_columns, uint32_t log_rows) {
  const size_t rows = (size_t)1 << log_rows;
  size_t total = align256((size_t)n_columns * 8 + 8) + 256 + align256((size_t)n_progs * sizeof(vm_part));
  for (uint32_t i = 0; i < n_progs; i++)
    total += align256(((size_t)progs[i].n_insns + 4) * sizeof(vm_uop)) + align256((size_t)progs[i].n_constants * 32 + 32) + align256((size_t)progs[i].n_rotations * 4 + 4);
  return total + align256(((size_t)1 << POW_LO_BITS) * 32) + align256(((rows >> POW_LO_BITS) + 1) * 32);
}

// the pinned buffer of `staging` whose turn it is, at least `bytes` long and zeroed (waits for the copy that last read it); nullptr = failure
static unsigned char* vm_staging_acquire(vm_staging* staging, size_t bytes, int* slot_out) {
  const int slot = staging->turn;
  staging->turn ^= 1;
  if (staging->copied[slot]) { if (hipEventSynchronize(staging->copied[slot]) != hipSuccess) { set_error("eval_rows: event wait failed"); return nullptr; } }
  else if (hipEventCreateWithFlags(&staging->copied[slot], hipEventDisableTiming) != hipSuccess) { set_error("eval_rows: hipEventCreate failed"); return nullptr; }
  if (staging->cap[slot] < bytes) {
    if (staging->host[slot]) (void)hipHostFree(staging->host[slot]);
    staging->host[slot] = nullptr; staging->cap[slot] = 0;
    const size_t want = align256(bytes + bytes / 2 + 4096);
    if (hipHostMalloc(&staging->host[slot], want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); set_error("eval_rows: hipHostMalloc(%zu) failed", want); return nullptr; }
    staging->cap[slot] = want;
  }
  std::memset(staging->host[slot], 0, bytes);
  *slot_out = slot;
  return (unsigned char*)staging->host[slot];
}

int row_vm_device_multi(const zkhip_vm_program* progs, uint32_t n_progs, const void* const* d_columns, uint32_t n_columns, uint32_t log_rows, uint32_t* const* d_outs,
                        void* ws, size_t ws_bytes, hipStream_t stream, vm_staging* staging) {
  const uint64_t rows = (uint64_t)1 << log_rows;
  if (n_progs == 0 || n_progs > 65535) { set_error("eval_rows: %u programs in one launch", n_progs); return ZKHIP_EINVAL; }
  if (ws_bytes < row_vm_multi_workspace_bytes(progs, n_progs, n_columns, log_rows)) { set_error("eval_rows: workspace too small"); return ZKHIP_EINVAL; }
  for (uint32_t i = 0; i < n_columns; i++)
    if (!d_columns[i]) { set_error("eval_rows: column %u is null", i); return ZKHIP_EINVAL; }
  const uint64_t* omega = nullptr;
  uint32_t top = 0;
  for (uint32_t i = 0; i < n_progs; i++) {
    if (progs[i].omega) {
      if (omega && std::memcmp(omega, progs[i].omega, 32) != 0) { set_error("eval_rows: programs of one launch disagree on omega"); return ZKHIP_EINVAL; }
      omega = progs[i].omega;
    }
    const uint32_t t = vm_top_register(&progs[i]);
    if (t > top) top = t;
  }
  const size_t o_cols = 0, o_omega = o_cols + align256((size_t)n_columns * 8 + 8), o_parts = o_omega + 256;
  size_t off = o_parts + align256((size_t)n_progs * sizeof(vm_part));
  std::vector<size_t> o_region(n_progs), o_const(n_progs), o_rot(n_progs);
  for (uint32_t i = 0; i < n_progs; i++) {
    o_region[i] = off;
    o_const[i] = align256(((size_t)progs[i].n_insns + 4) * sizeof(vm_uop));
    o_rot[i] = o_const[i] + align256((size_t)progs[i].n_constants * 32 + 32);
    off += o_rot[i] + align256((size_t)progs[i].n_rotations * 4 + 4);
  }
  const size_t o_lo = off, o_hi = o_lo + align256(((size_t)1 << POW_LO_BITS) * 32);
  // the blob: pinned staging of the caller's scratch set (no wait for the stream), or a local vector (then the stream is synchronised below)
  std::vector<unsigned char> local;
  int slot = -1;
  struct { unsigned char* p; unsigned char* data() const { return p; } } blob{nullptr};
  if (staging) {
    blob.p = vm_staging_acquire(staging, o_lo, &slot);
    if (!blob.p) return ZKHIP_EHIP;
  } else {
    local.assign(o_lo, 0);
    blob.p = local.data();
  }
  char* d = (char*)ws;
  for (uint32_t i = 0; i < n_columns; i++) std::memcpy(blob.data() + o_cols + (size_t)i * 8, &d_columns[i], 8);
  if (omega) std::memcpy(blob.data() + o_omega, omega, 32);
  vm_part* parts = (vm_part*)(blob.data() + o_parts);
  for (uint32_t i = 0; i < n_progs; i++) {
    vm_fill_region(&progs[i], d_columns, rows, blob.data() + o_region[i], (uint64_t)(d + o_region[i]), o_const[i], o_rot[i]);
    parts[i].prog = (const void*)(d + o_region[i]);
    parts[i].consts = (const uint32_t*)(d + o_region[i] + o_const[i]);
    parts[i].rot_off = (const uint32_t*)(d + o_region[i] + o_rot[i]);
    parts[i].out = d_outs[i];
    parts[i].n_insns = progs[i].n_insns;
    parts[i].result_reg = progs[i].result_reg;
  }
  HIPCHK(hipMemcpyAsync(d, blob.data(), o_lo, hipMemcpyHostToDevice, stream));
  if (staging) HIPCHK(hipEventRecord(staging->copied[slot], stream));
  else HIPCHK(hipStreamSynchronize(stream));                 // the blob is a local
  vm_launch L;
  L.prog = parts[0].prog; L.n_insns = parts[0].n_insns; L.result_reg = parts[0].result_reg;
  L.cols = (const uint32_t* const*)(d + o_cols);
  L.consts = parts[0].consts; L.rot_off = parts[0].rot_off;
  L.pow_lo = nullptr; L.pow_hi = nullptr;
  L.out = d_outs[0];
  L.rows = rows;
  L.accumulate = 0;
  L.parts = (const vm_part*)(d + o_parts);
  if (omega) {
    const uint32_t n_lo = 1u << POW_LO_BITS, n_hi = (uint32_t)((rows >> POW_LO_BITS) ? (rows >> POW_LO_BITS) : 1);
    hipLaunchKernelGGL(k_vm_pow_table, dim3((n_lo + 255) / 256), dim3(256), 0, stream, (const fe_arg*)(d + o_omega), 0u, n_lo, (uint32_t*)(d + o_lo));
    hipLaunchKernelGGL(k_vm_pow_table, dim3((n_hi + 255) / 256), dim3(256), 0, stream, (const fe_arg*)(d + o_omega), POW_LO_BITS, n_hi, (uint32_t*)(d + o_hi));
    L.pow_lo = (const uint32_t*)(d + o_lo);
    L.pow_hi = (const uint32_t*)(d + o_hi);
  }
  return vm_launch_kernel(L, top, n_progs, stream);
}

int row_vm_device(const zkhip_vm_program* p, const void* const* d_columns, uint32_t n_columns, uint32_t log_rows, int accumulate,
                  uint32_t* d_out, void* ws, size_t ws_bytes, hipStream_t stream, vm_staging* staging) {
  const uint64_t rows = (uint64_t)1 << log_rows;
  if (ws_bytes < row_vm_workspace_bytes(p, n_columns, log_rows)) { set_error("eval_rows: workspace too small"); return ZKHIP_EINVAL; }
  // one host blob, one copy
  const size_t o_prog = 0;
  const size_t o_const = o_prog + align256(((size_t)p->n_insns + 4) * sizeof(vm_uop));
  const size_t o_rot = o_const + align256((size_t)p->n_constants * 32 + 32);
  const size_t o_cols = o_rot + align256((size_t)p->n_rotations * 4 + 4);
  const size_t o_omega = o_cols + align256((size_t)n_columns * 8 + 8);
  const size_t o_lo = o_omega + 256;
  const size_t o_hi = o_lo + align256(((size_t)1 << POW_LO_BITS) * 32);
  // the blob: pinned staging of the caller's scratch set (no wait for the stream), or a local vector (then the stream is synchronised below)
  std::vector<unsigned char> local;
  unsigned char* blob_p = nullptr;
  int slot = -1;
  if (staging) {
    slot = staging->turn;
    staging->turn ^= 1;
    if (staging->copied[slot]) HIPCHK(hipEventSynchronize(staging->copied[slot]));          // the copy that last read this buffer is done
    else HIPCHK(hipEventCreateWithFlags(&staging->copied[slot], hipEventDisableTiming));
    if (staging->cap[slot] < o_lo) {
      if (staging->host[slot]) (void)hipHostFree(staging->host[slot]);
      staging->host[slot] = nullptr; staging->cap[slot] = 0;
      const size_t want = align256(o_lo + o_lo / 2 + 4096);
      if (hipHostMalloc(&staging->host[slot], want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); set_error("eval_rows: hipHostMalloc(%zu) failed", want); return ZKHIP_ENOMEM; }
      staging->cap[slot] = want;
    }
    blob_p = (unsigned char*)staging->host[slot];
    std::memset(blob_p, 0, o_lo);
  } else {
    local.assign(o_lo, 0);
    blob_p = local.data();
  }
  struct { unsigned char* p; unsigned char* data() const { return p; } } blob{blob_p};
  for (uint32_t i = 0; i < n_columns; i++)
    if (!d_columns[i]) { set_error("eval_rows: column %u is null", i); return ZKHIP_EINVAL; }
  vm_fill_region(p, d_columns, rows, blob.data(), (uint64_t)(char*)ws, o_const, o_rot);
  for (uint32_t i = 0; i < n_columns; i++) std::memcpy(blob.data() + o_cols + (size_t)i * 8, &d_columns[i], 8);
  if (p->omega) std::memcpy(blob.data() + o_omega, p->omega, 32);
  char* d = (char*)ws;
  HIPCHK(hipMemcpyAsync(d, blob.data(), o_lo, hipMemcpyHostToDevice, stream));
  if (staging) HIPCHK(hipEventRecord(staging->copied[slot], stream));
  else HIPCHK(hipStreamSynchronize(stream));   // the blob is a local
  vm_launch L;
  L.parts = nullptr;
  L.prog = (const void*)(d + o_prog);
  L.n_insns = p->n_insns;
  L.result_reg = p->result_reg;
  L.cols = (const uint32_t* const*)(d + o_cols);
  L.consts = (const uint32_t*)(d + o_const);
  L.rot_off = (const uint32_t*)(d + o_rot);
  L.pow_lo = nullptr;
  L.pow_hi = nullptr;
  L.out = d_out;
  L.rows = rows;
  L.accumulate = accumulate ? 1u : 0u;
  if (p->omega) {
    const uint32_t n_lo = 1u << POW_LO_BITS, n_hi = (uint32_t)((rows >> POW_LO_BITS) ? (rows >> POW_LO_BITS) : 1);
    hipLaunchKernelGGL(k_vm_pow_table, dim3((n_lo + 255) / 256), dim3(256), 0, stream, (const fe_arg*)(d + o_omega), 0u, n_lo, (uint32_t*)(d + o_lo));
    hipLaunchKernelGGL(k_vm_pow_table, dim3((n_hi + 255) / 256), dim3(256), 0, stream, (const fe_arg*)(d + o_omega), POW_LO_BITS, n_hi, (uint32_t*)(d + o_hi));
    L.pow_lo = (const uint32_t*)(d + o_lo);
    L.pow_hi = (const uint32_t*)(d + o_hi);
  }
  // a short program over many rows runs as compiled straight-line code when a kernel for its shape exists or can be built (rowvm_jit.hip);
  // any failure there leaves nothing launched and the interpreter below takes the call
  if (row_vm_jit_wanted(p, n_columns, log_rows) &&
      row_vm_jit_launch(p, d_columns, n_columns, log_rows, accumulate, L.consts, L.pow_lo, L.pow_hi, d_out, stream) == ZKHIP_OK)
    return ZKHIP_OK;
  // smallest register-file variant that holds every register the program names
  return vm_launch_kernel(L, vm_top_register(p), 1, stream);
}

int fr_pointwise_mul_device(const uint32_t* d_a, const uint32_t* d_b, size_t n, uint32_t* d_out, hipStream_t stream) {
  if (n == 0) return ZKHIP_OK;
  hipLaunchKernelGGL(k_fr_pointwise_mul, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_a, d_b, n, d_out);
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}

int fr_gather_mul_device(const uint32_t* d_a, uint32_t a_len, const uint32_t* d_ia, const uint32_t* d_b, uint32_t b_len, const uint32_t* d_ib, size_t n,
                         uint32_t* d_out, hipStream_t stream) {
  if (n == 0) return ZKHIP_OK;
  hipLaunchKernelGGL(k_fr_gather_mul, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_a, a_len, d_ia, d_b, b_len, d_ib, n, d_out);
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}

}  // namespace zkhip
