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
  const uint4* prog;            // n_insns x 16 bytes
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
};

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

// operand fetch; TIMES32: the value scaled by 2^5 (second factor of a product)
template <bool TIMES32, int R>
__device__ __forceinline__ fe vm_fetch(const vm_launch& L, uint32_t opnd, uint64_t row, const fe (&r)[R], const fe& prev, const fe& xpow) {
  const uint32_t kind = opnd & 0xff, rot = (opnd >> 8) & 0xff, index = opnd >> 16;
  if (kind == ZKHIP_SRC_COLUMN || kind == ZKHIP_SRC_CONST) {
    uint32_t w[8];
    if (kind == ZKHIP_SRC_COLUMN) {
      const uint64_t rr = (row + L.rot_off[rot]) & (L.rows - 1);
      load_words(L.cols[index] + rr * 8, w);
    } else {
      load_words(L.consts + (size_t)index * 8, w);
    }
    return TIMES32 ? fe_unpack<5>(w) : fe_unpack<0>(w);
  }
  fe v = kind == ZKHIP_SRC_REG ? vm_reg_get(r, index) : (kind == ZKHIP_SRC_PREV ? prev : xpow);
  return TIMES32 ? vm_times32(v) : v;
}

template <int R>
__global__ void __launch_bounds__(256) k_row_vm(const vm_launch L) {
  const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= L.rows) return;
  fe r[R];
#pragma unroll
  for (int i = 0; i < R; i++) r[i] = fe_zero();
  fe prev = fe_zero(), xpow = fe_zero();
  if (L.accumulate) prev = load_ext(L.out, row);
  if (L.pow_lo) {
    uint32_t w[8];
    load_words(L.pow_lo + (row & ((1u << POW_LO_BITS) - 1)) * 8, w);
    xpow = fe_mul<Fr>(load_ext(L.pow_hi, row >> POW_LO_BITS), fe_unpack<5>(w));
  }
#pragma unroll 1
  for (uint32_t pc = 0; pc < L.n_insns; pc++) {
    const uint4 q = L.prog[pc];
    // same instruction for every lane: keep the decode on the scalar unit
    const uint32_t head = __builtin_amdgcn_readfirstlane(q.x), oa = __builtin_amdgcn_readfirstlane(q.y),
                   ob = __builtin_amdgcn_readfirstlane(q.z), oc = __builtin_amdgcn_readfirstlane(q.w);
    const uint32_t op = head & 0xff, dst = (head >> 8) & 0xff;
    const fe a = vm_fetch<false, R>(L, oa, row, r, prev, xpow);
    fe t;
    if (op == ZKHIP_OP_MUL || op == ZKHIP_OP_SQR || op == ZKHIP_OP_MAD) {
      const fe b32 = op == ZKHIP_OP_SQR ? vm_times32(a) : vm_fetch<true, R>(L, ob, row, r, prev, xpow);
      t = fe_mul<Fr, true>(a, b32);      // single-chain product columns (fp29.hpp): -2 % on the wrapper quotient, same box
      if (op == ZKHIP_OP_MAD) t = vm_add(t, vm_fetch<false, R>(L, oc, row, r, prev, xpow));
    } else if (op == ZKHIP_OP_ADD) {
      t = vm_add(a, vm_fetch<false, R>(L, ob, row, r, prev, xpow));
    } else if (op == ZKHIP_OP_SUB) {
      t = vm_sub(a, vm_fetch<false, R>(L, ob, row, r, prev, xpow));
    } else if (op == ZKHIP_OP_NEG) {
      t = vm_sub(fe_zero(), a);
    } else if (op == ZKHIP_OP_DBL) {
      t = vm_add(a, a);
    } else {
      t = a;   // MOV
    }
    vm_reg_set(r, dst, t);
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
  return align256((size_t)p->n_insns * 16) + align256((size_t)p->n_constants * 32 + 32) + align256((size_t)p->n_rotations * 4 + 4) +
         align256((size_t)n_columns * 8 + 8) + 256 + align256(((size_t)1 << POW_LO_BITS) * 32) + align256(((rows >> POW_LO_BITS) + 1) * 32);
}

// the program, its tables and the column pointers are staged into `ws` (device), then one launch
int row_vm_device(const zkhip_vm_program* p, const void* const* d_columns, uint32_t n_columns, uint32_t log_rows, int accumulate,
                  uint32_t* d_out, void* ws, size_t ws_bytes, hipStream_t stream) {
  const uint64_t rows = (uint64_t)1 << log_rows;
  if (ws_bytes < row_vm_workspace_bytes(p, n_columns, log_rows)) { set_error("eval_rows: workspace too small"); return ZKHIP_EINVAL; }
  // one host blob, one copy
  const size_t o_prog = 0;
  const size_t o_const = o_prog + align256((size_t)p->n_insns * 16);
  const size_t o_rot = o_const + align256((size_t)p->n_constants * 32 + 32);
  const size_t o_cols = o_rot + align256((size_t)p->n_rotations * 4 + 4);
  const size_t o_omega = o_cols + align256((size_t)n_columns * 8 + 8);
  const size_t o_lo = o_omega + 256;
  const size_t o_hi = o_lo + align256(((size_t)1 << POW_LO_BITS) * 32);
  std::vector<unsigned char> blob(o_lo, 0);
  static_assert(sizeof(zkhip_vm_insn) == 16, "instruction layout");
  std::memcpy(blob.data() + o_prog, p->insns, (size_t)p->n_insns * 16);
  // a product's second factor is fetched scaled by 2^5: free for a column / constant (the other unpacking shift), a repack for a
  // register -- so put the memory operand second where the host did not
  for (uint32_t pc = 0; pc < p->n_insns; pc++) {
    zkhip_vm_insn* in = (zkhip_vm_insn*)(blob.data() + o_prog) + pc;
    const bool a_mem = in->a.kind == ZKHIP_SRC_COLUMN || in->a.kind == ZKHIP_SRC_CONST;
    const bool b_mem = in->b.kind == ZKHIP_SRC_COLUMN || in->b.kind == ZKHIP_SRC_CONST;
    if ((in->op == ZKHIP_OP_MUL || in->op == ZKHIP_OP_MAD) && a_mem && !b_mem) { const zkhip_vm_operand t = in->a; in->a = in->b; in->b = t; }
  }
  if (p->n_constants) std::memcpy(blob.data() + o_const, p->constants, (size_t)p->n_constants * 32);
  for (uint32_t i = 0; i < p->n_rotations; i++) {
    const int64_t off = ((int64_t)p->rotations[i] * (int64_t)p->rot_scale) % (int64_t)rows;
    const uint32_t v = (uint32_t)(off < 0 ? off + (int64_t)rows : off);
    std::memcpy(blob.data() + o_rot + (size_t)i * 4, &v, 4);
  }
  for (uint32_t i = 0; i < n_columns; i++) {
    if (!d_columns[i]) { set_error("eval_rows: column %u is null", i); return ZKHIP_EINVAL; }
    std::memcpy(blob.data() + o_cols + (size_t)i * 8, &d_columns[i], 8);
  }
  if (p->omega) std::memcpy(blob.data() + o_omega, p->omega, 32);
  char* d = (char*)ws;
  HIPCHK(hipMemcpyAsync(d, blob.data(), blob.size(), hipMemcpyHostToDevice, stream));
  HIPCHK(hipStreamSynchronize(stream));   // the blob is a local
  vm_launch L;
  L.prog = (const uint4*)(d + o_prog);
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
  // smallest register-file variant that holds every register the program names
  uint32_t top = p->result_reg;
  for (uint32_t pc = 0; pc < p->n_insns; pc++) {
    const zkhip_vm_insn& in = p->insns[pc];
    if (in.dst > top) top = in.dst;
    const zkhip_vm_operand* o[3] = {&in.a, &in.b, &in.c};
    for (int k = 0; k < 3; k++) if (o[k]->kind == ZKHIP_SRC_REG && o[k]->index < (uint32_t)VM_MAX_REGS && o[k]->index > top) top = o[k]->index;
  }
  const dim3 grid((unsigned)((rows + 255) / 256));
  if (top < 6) hipLaunchKernelGGL(k_row_vm<6>, grid, dim3(256), 0, stream, L);
  else if (top < 8) hipLaunchKernelGGL(k_row_vm<8>, grid, dim3(256), 0, stream, L);
  else if (top < 12) hipLaunchKernelGGL(k_row_vm<12>, grid, dim3(256), 0, stream, L);
  else hipLaunchKernelGGL(k_row_vm<16>, grid, dim3(256), 0, stream, L);
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
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
