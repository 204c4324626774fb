// Row programs compiled at run time (round 5; round-4 review item 7): the interpreter of rowvm.hip spends about half of
// its wave-instructions on interpretation -- register-file selects, operand-kind dispatch, the repack of a register that becomes a product's
// second factor, a conditional subtraction after every addition -- and the wrapper's quotient program is the same 121 instructions for
// every proof of a circuit.  Here such a program is turned into straight-line HIP source (registers are named variables, every operand kind
// and rotation is resolved in the text, additions stay lazy with bounds tracked AT CODE-GENERATION TIME so that a reduction is emitted only
// where a product needs it), compiled once per (program shape, rows, device) with hiprtc and cached; constants -- the challenges change with
// every proof -- stay a device table, so one compilation serves every proof of the circuit.
//
// What it replaces on the reference side: the per-row `GraphEvaluator::evaluate` loop of [DEP] halo2-axiom plonk/evaluation.rs `evaluate_h`,
// reached from create_proof (/root/reference/aggregator/src/wrapper.rs:129); gate shape of the wrapper: wrapper.rs:792-797.
//
// hiprtc is opened with dlopen on first use (no link dependency); without it, or when a compilation fails, or for programs outside the limits
// below (the 4 k-instruction programs of the wide circuits: minutes of compile time), row_vm_device runs the interpreter -- the same results,
// both on the GPU.  $ZKHIP_VM_JIT: 0 = never, 1 (default) = programs of at most 256 instructions over at least 2^18 rows, 2 = every program of
// at most 256 instructions (tests).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "zkhip_internal.hpp"
#include "embedded_headers.inc"

namespace zkhip {

namespace {

constexpr uint32_t JIT_MAX_INSNS = 256;
constexpr uint32_t JIT_MAX_COLUMNS = 96;        // column pointers travel in the kernel argument block
constexpr uint32_t JIT_POW_LO_BITS = 12;        // rowvm.hip POW_LO_BITS

// ---- hiprtc, by name ---------------------------------------------------------------------------------------------------------------------
typedef void* rtc_program;
struct rtc_api {
  int (*create)(rtc_program*, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
  int (*compile)(rtc_program, int, const char* const*) = nullptr;
  int (*log_size)(rtc_program, size_t*) = nullptr;
  int (*log)(rtc_program, char*) = nullptr;
  int (*code_size)(rtc_program, size_t*) = nullptr;
  int (*code)(rtc_program, char*) = nullptr;
  int (*destroy)(rtc_program*) = nullptr;
  bool ok = false;
};
const rtc_api& rtc() {
  static const rtc_api api = [] {
    rtc_api a;
    void* h = nullptr;
    for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
      h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
    if (!h) return a;
    a.create = (decltype(a.create))dlsym(h, "hiprtcCreateProgram");
    a.compile = (decltype(a.compile))dlsym(h, "hiprtcCompileProgram");
    a.log_size = (decltype(a.log_size))dlsym(h, "hiprtcGetProgramLogSize");
    a.log = (decltype(a.log))dlsym(h, "hiprtcGetProgramLog");
    a.code_size = (decltype(a.code_size))dlsym(h, "hiprtcGetCodeSize");
    a.code = (decltype(a.code))dlsym(h, "hiprtcGetCode");
    a.destroy = (decltype(a.destroy))dlsym(h, "hiprtcDestroyProgram");
    a.ok = a.create && a.compile && a.log_size && a.log && a.code_size && a.code && a.destroy;
    return a;
  }();
  return api;
}

int jit_mode() {
  static const int m = [] { const char* e = getenv("ZKHIP_VM_JIT"); const int v = e ? atoi(e) : 1; return (v >= 0 && v <= 2) ? v : 1; }();
  return m;
}

// ---- code generation ---------------------------------------------------------------------------------------------------------------------
// Bounds are in units of r.  Every named value is in N form (fp29.hpp): a load gives < 1, a product of operands <= 2 gives
// 2 * 32 * 2 / 169.3 + 1 < 1.76 (the second factor is scaled by 2^5), a sum adds the bounds, a difference a - b adds K with K r the smallest
// borrow-proof constant of FrParams above b's bound.  Before a product an operand above 2 is brought back: one conditional subtraction of 2r
// up to 4, the quotient-estimate reduction (fe_reduce_soft, < 2r + 2^233) above that.
const int RED_K[] = {2, 3, 4, 6, 8, 10, 12, 16, 32, 64};

struct gen {
  std::string body;
  double bound[ZKHIP_VM_REGS];
  bool used[ZKHIP_VM_REGS];
  int tmp = 0;
  const zkhip_vm_program* p;
  uint64_t rows;
  std::vector<uint32_t> rot_rows;

  void line(const std::string& s) { body += "  " + s + "\n"; }
  std::string reg(uint32_t i) { used[i] = true; return "r" + std::to_string(i); }
  std::string fresh() { return "t" + std::to_string(tmp++); }

  // `name` (bound b) -> at most 2
  void reduce(const std::string& name, double& b) {
    if (b <= 2.0) return;
    if (b <= 4.0) { line(name + " = condsub2(" + name + ");"); b = 2.0; }                        // < 2r
    else { line(name + " = fe_reduce_soft<Fr>(" + name + ");"); b = 2.001; }                   // < 2r + 2^233: NOT below 2r (a 3r - b would borrow)
  }
  static int red_for(double b) {
    for (int k : RED_K) if ((double)(k - 1) >= b) return k;
    return 0;
  }

  struct val { std::string expr; double bound; bool is_reg; uint32_t reg; };

  // operand as a value expression; sh5: scaled by 2^5 (a product's second factor).  Memory operands are loaded into a fresh variable.
  val operand(const zkhip_vm_operand& o, bool sh5) {
    val v{"", 1.0, false, 0};
    const std::string sh = sh5 ? "5" : "0";
    switch (o.kind) {
      case ZKHIP_SRC_COLUMN: {
        const std::string t = fresh();
        const uint32_t off = rot_rows[o.rot];
        line("const fe " + t + " = ldx<" + sh + ">(A.cols[" + std::to_string(o.index) + "], (row + " + std::to_string(off) + "ull) & (A.rows - 1));");
        v.expr = t;
        break;
      }
      case ZKHIP_SRC_CONST: {
        const std::string t = fresh();
        line("const fe " + t + " = ldx<" + sh + ">(A.consts, " + std::to_string(o.index) + "ull);");
        v.expr = t;
        break;
      }
      case ZKHIP_SRC_REG: {
        v.is_reg = true; v.reg = o.index;
        const std::string r = reg(o.index);
        if (sh5) { reduce(r, bound[o.index]); v.expr = "times32(" + r + ")"; }      // the repack needs a value below 2^256
        else v.expr = r;
        v.bound = bound[o.index];
        break;
      }
      case ZKHIP_SRC_PREV: v.expr = sh5 ? "times32(vprev)" : "vprev"; v.bound = 1.0; break;
      default: v.expr = sh5 ? "times32(vpow)" : "vpow"; v.bound = 1.76; break;     // ZKHIP_SRC_ROWPOW
    }
    return v;
  }
  static bool is_mem(const zkhip_vm_operand& o) { return o.kind == ZKHIP_SRC_COLUMN || o.kind == ZKHIP_SRC_CONST; }

  // product a * b as an expression; *out_bound = Ba * 32 Bb / 169.3 + 1 (the Montgomery product's lazy bound, radix 2^261 = 169.3 r), kept at or
  // below 2 by reducing a register operand first when Ba * Bb > 5.29 (a memory operand is canonical: bound 1; the repack of a register that
  // becomes the second factor needs it below 2^256 = 5.29 r)
  std::string product(zkhip_vm_operand a, zkhip_vm_operand b, double* out_bound) {
    if (is_mem(a) && !is_mem(b)) { const zkhip_vm_operand t = a; a = b; b = t; }       // the memory operand takes the free scaling
    auto bnd = [&](const zkhip_vm_operand& o) { return o.kind == ZKHIP_SRC_REG ? bound[o.index] : (o.kind == ZKHIP_SRC_ROWPOW ? 1.76 : 1.0); };
    for (int pass = 0; pass < 2 && bnd(a) * bnd(b) > 5.29; pass++) {
      const zkhip_vm_operand& big = bnd(a) >= bnd(b) ? a : b;
      if (big.kind != ZKHIP_SRC_REG) break;
      reduce(reg(big.index), bound[big.index]);
    }
    if (b.kind == ZKHIP_SRC_REG && bound[b.index] > 5.29) reduce(reg(b.index), bound[b.index]);
    const val va = operand(a, false);
    std::string vb;
    double bb;
    if (b.kind == ZKHIP_SRC_REG) { vb = "times32(" + reg(b.index) + ")"; bb = bound[b.index]; }
    else { const val v = operand(b, true); vb = v.expr; bb = v.bound; }
    *out_bound = va.bound * bb * 32.0 / 169.3 + 1.0;
    return "fe_mul<Fr, false>(" + va.expr + ", " + vb + ")";
  }

  void set(uint32_t dst, const std::string& expr, double b) {
    line(reg(dst) + " = " + expr + ";");
    bound[dst] = b;
  }

  bool emit(const zkhip_vm_insn& in) {
    switch (in.op) {
      case ZKHIP_OP_MOV: { const val a = operand(in.a, false); set(in.dst, a.expr, a.bound); break; }
      case ZKHIP_OP_MUL: { double pb; const std::string e = product(in.a, in.b, &pb); set(in.dst, e, pb); break; }
      case ZKHIP_OP_SQR: {
        if (in.a.kind == ZKHIP_SRC_REG && bound[in.a.index] > 2.3) reduce(reg(in.a.index), bound[in.a.index]);     // 2.3^2 = 5.29
        val a = operand(in.a, false);
        if (!a.is_reg) { const std::string t = fresh(); line("const fe " + t + " = " + a.expr + ";"); a.expr = t; }
        set(in.dst, "fe_mul<Fr, false>(" + a.expr + ", times32(" + a.expr + "))", a.bound * a.bound * 32.0 / 169.3 + 1.0);
        break;
      }
      case ZKHIP_OP_MAD: {
        const std::string t = fresh();
        double pb;
        const std::string e = product(in.a, in.b, &pb);
        line("const fe " + t + " = " + e + ";");
        val c = operand(in.c, false);
        if (c.is_reg && c.bound > 60.0) { reduce(reg(c.reg), bound[c.reg]); c.bound = bound[c.reg]; }
        set(in.dst, "fe_norm(fe_add(" + t + ", " + c.expr + "))", pb + c.bound);
        break;
      }
      case ZKHIP_OP_ADD: {
        val a = operand(in.a, false), b = operand(in.b, false);
        if (a.bound + b.bound > 60.0) {
          if (a.is_reg) { reduce(reg(a.reg), bound[a.reg]); a.bound = bound[a.reg]; }
          if (b.is_reg) { reduce(reg(b.reg), bound[b.reg]); b.bound = bound[b.reg]; }
        }
        set(in.dst, "fe_norm(fe_add(" + a.expr + ", " + b.expr + "))", a.bound + b.bound);
        break;
      }
      case ZKHIP_OP_SUB: {
        val a = operand(in.a, false), b = operand(in.b, false);
        if (a.is_reg && a.bound > 40.0) { reduce(reg(a.reg), bound[a.reg]); a.bound = bound[a.reg]; }
        if (b.is_reg && b.bound > 15.0) { reduce(reg(b.reg), bound[b.reg]); b.bound = bound[b.reg]; }
        const int k = red_for(b.bound);
        if (!k) return false;
        set(in.dst, "fe_norm(fe_sub_red(" + a.expr + ", " + b.expr + ", Fr::P" + std::to_string(k) + "_S1))", a.bound + (double)k);
        break;
      }
      case ZKHIP_OP_NEG: {
        val a = operand(in.a, false);
        if (a.is_reg && a.bound > 15.0) { reduce(reg(a.reg), bound[a.reg]); a.bound = bound[a.reg]; }
        const int k = red_for(a.bound);
        if (!k) return false;
        set(in.dst, "fe_norm(fe_neg_red(" + a.expr + ", Fr::P" + std::to_string(k) + "_S1))", (double)k);
        break;
      }
      case ZKHIP_OP_DBL: {
        val a = operand(in.a, false);
        if (a.is_reg && a.bound > 30.0) { reduce(reg(a.reg), bound[a.reg]); a.bound = bound[a.reg]; }
        set(in.dst, "fe_norm(fe_dbl(" + a.expr + "))", 2.0 * a.bound);
        break;
      }
      default: return false;
    }
    return true;
  }
};

const char* const JIT_PRELUDE_TYPES =
    "typedef unsigned int uint32_t; typedef int int32_t; typedef unsigned long long uint64_t; typedef long long int64_t;\n"
    "namespace std { template <class T, T v> struct integral_constant { static constexpr T value = v; using value_type = T; constexpr operator T() const { return v; } }; }\n";

const char* const JIT_PRELUDE_HELPERS = R"ZKJIT(
using namespace zkhip;
using Fr = FrParams;
// limb i of 2r in N form
__device__ constexpr uint32_t two_r_limb(int i) {
  uint32_t carry = 0, out = 0;
  for (int j = 0; j <= i; j++) {
    const uint32_t t = 2u * Fr::P[j] + carry;
    out = j < NL - 1 ? (t & LMASK) : t;
    carry = j < NL - 1 ? (t >> LB) : 0;
  }
  return out;
}
// N-form value < 4r -> N-form value < 2r, same residue
__device__ __forceinline__ fe condsub2(const fe& s) {
  fe d;
  int32_t borrow = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const int32_t t = (int32_t)s.l[i] - (int32_t)two_r_limb(i) + borrow;
    borrow = t >> 31;
    d.l[i] = i < NL - 1 ? ((uint32_t)t & LMASK) : (uint32_t)t;
  }
  fe r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.l[i] = borrow ? s.l[i] : d.l[i];
  return r;
}
// value < 2^256 scaled by 2^5: repack with the other shift
__device__ __forceinline__ fe times32(const fe& a) {
  uint32_t w[8];
  fe_pack(a, w);
  return fe_unpack<5>(w);
}
// element `idx` of an array of external words, unpacked with shift SH (0: the value, 5: the value times 2^5)
template <int SH>
__device__ __forceinline__ fe ldx(const uint32_t* base, uint64_t idx) {
  uint32_t w[8];
  load_words(base + idx * 8, w);
  return fe_unpack<SH>(w);
}
)ZKJIT";

std::string cat(const char* const* chunks) {
  std::string s;
  for (; *chunks; chunks++) s += *chunks;
  return s;
}

// the whole translation unit for program p over 2^log_rows rows; empty on a program the generator does not take
std::string generate(const zkhip_vm_program* p, uint32_t n_columns, uint32_t log_rows) {
  gen g;
  g.p = p;
  g.rows = (uint64_t)1 << log_rows;
  for (int i = 0; i < ZKHIP_VM_REGS; i++) { g.bound[i] = 0.0; g.used[i] = false; }
  g.rot_rows.assign(p->n_rotations ? p->n_rotations : 1, 0u);
  for (uint32_t i = 0; i < p->n_rotations; i++) {
    const int64_t off = ((int64_t)p->rotations[i] * (int64_t)p->rot_scale) % (int64_t)g.rows;
    g.rot_rows[i] = (uint32_t)(off < 0 ? off + (int64_t)g.rows : off);
  }
  bool uses_prev = false, uses_pow = false;
  for (uint32_t pc = 0; pc < p->n_insns; pc++) {
    const zkhip_vm_insn& in = p->insns[pc];
    const int n_opnd = (in.op == ZKHIP_OP_MAD) ? 3 : ((in.op == ZKHIP_OP_ADD || in.op == ZKHIP_OP_SUB || in.op == ZKHIP_OP_MUL) ? 2 : 1);
    const zkhip_vm_operand* o[3] = {&in.a, &in.b, &in.c};
    for (int k = 0; k < n_opnd; k++) { uses_prev |= o[k]->kind == ZKHIP_SRC_PREV; uses_pow |= o[k]->kind == ZKHIP_SRC_ROWPOW; }
    g.line("// " + std::to_string(pc));
    if (!g.emit(in)) return std::string();
  }
  // result: canonical words
  {
    const std::string r = g.reg(p->result_reg);
    double& b = g.bound[p->result_reg];
    if (b > 3.0) g.reduce(r, b);                 // condsub -> < 2r; the soft reduction -> < 2r + 2^233 < 3r
    g.line("{ uint32_t w[8]; fe_pack(fe_canon_lt3p<Fr>(" + r + "), w); store_words(A.out + row * 8, w); }");
  }
  std::string src = JIT_PRELUDE_TYPES;
  src += cat(JIT_HDR_CONSTANTS);
  src += cat(JIT_HDR_MAC_BLOCKS);
  src += cat(JIT_HDR_FP29);
  src += JIT_PRELUDE_HELPERS;
  src += "struct jit_args { const uint32_t* cols[" + std::to_string(n_columns ? n_columns : 1) +
         "]; const uint32_t* consts; const uint32_t* pow_lo; const uint32_t* pow_hi; uint32_t* out; uint64_t rows; uint32_t accumulate; };\n";
  src += "extern \"C\" __global__ void __launch_bounds__(256) zk_row_jit(const jit_args A) {\n";
  src += "  const uint64_t row = (uint64_t)blockIdx.x * 256 + threadIdx.x;\n  if (row >= A.rows) return;\n";
  std::string decl = "  fe";
  bool any = false;
  for (int i = 0; i < ZKHIP_VM_REGS; i++)
    if (g.used[i]) { decl += std::string(any ? ", " : " ") + "r" + std::to_string(i) + " = fe_zero()"; any = true; }
  if (any) src += decl + ";\n";
  if (uses_prev) src += "  const fe vprev = A.accumulate ? ldx<0>(A.out, row) : fe_zero();\n";
  if (uses_pow)
    src += "  const fe vpow = fe_mul<Fr, false>(ldx<0>(A.pow_hi, row >> " + std::to_string(JIT_POW_LO_BITS) + "), ldx<5>(A.pow_lo, row & " +
           std::to_string((1u << JIT_POW_LO_BITS) - 1) + "ull));\n";
  src += g.body;
  src += "}\n";
  return src;
}

// ---- cache ---------------------------------------------------------------------------------------------------------------------------------
struct compiled { hipModule_t mod = nullptr; hipFunction_t fn = nullptr; bool failed = false; };
std::mutex g_jit_mu;
std::map<std::string, compiled> g_jit_cache;       // key: device | rows | columns | the instruction bytes | scaled rotations | result register

std::string cache_key(const zkhip_vm_program* p, uint32_t n_columns, uint32_t log_rows, int device) {
  std::string k;
  auto put = [&](const void* d, size_t n) { k.append((const char*)d, n); };
  put(&device, sizeof(device)); put(&log_rows, sizeof(log_rows)); put(&n_columns, sizeof(n_columns));
  put(&p->n_insns, sizeof(p->n_insns)); put(p->insns, (size_t)p->n_insns * sizeof(zkhip_vm_insn));
  put(&p->n_rotations, sizeof(p->n_rotations));
  if (p->n_rotations) put(p->rotations, (size_t)p->n_rotations * 4);
  put(&p->rot_scale, sizeof(p->rot_scale)); put(&p->result_reg, sizeof(p->result_reg));
  return k;
}

}  // namespace

bool row_vm_jit_wanted(const zkhip_vm_program* p, uint32_t n_columns, uint32_t log_rows) {
  const int m = jit_mode();
  if (m == 0 || p->n_insns > JIT_MAX_INSNS || n_columns > JIT_MAX_COLUMNS) return false;
  return m == 2 || log_rows >= 18;
}

// Launches the compiled kernel for p (compiling it on first use).  ZKHIP_OK: launched.  Any other status: nothing was launched and the caller
// runs the interpreter (the reason is in set_error's text for the log, not an error of the call).
int row_vm_jit_launch(const zkhip_vm_program* p, const void* const* d_columns, uint32_t n_columns, uint32_t log_rows, int accumulate, const uint32_t* d_consts,
                      const uint32_t* d_pow_lo, const uint32_t* d_pow_hi, uint32_t* d_out, hipStream_t stream) {
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return ZKHIP_EHIP;
  const std::string key = cache_key(p, n_columns, log_rows, device);
  compiled c;
  {
    std::lock_guard<std::mutex> g(g_jit_mu);
    auto it = g_jit_cache.find(key);
    if (it == g_jit_cache.end()) {
      compiled n;
      n.failed = true;
      const rtc_api& R = rtc();
      const std::string src = R.ok ? generate(p, n_columns, log_rows) : std::string();
      if (!src.empty()) {
        rtc_program prog = nullptr;
        if (R.create(&prog, src.c_str(), "zk_row_jit.hip", 0, nullptr, nullptr) == 0) {
          const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
          const int rc = R.compile(prog, 3, opts);
          size_t cs = 0;
          if (rc == 0 && R.code_size(prog, &cs) == 0 && cs) {
            std::vector<char> code(cs);
            if (R.code(prog, code.data()) == 0 && hipModuleLoadData(&n.mod, code.data()) == hipSuccess &&
                hipModuleGetFunction(&n.fn, n.mod, "zk_row_jit") == hipSuccess)
              n.failed = false;
            else (void)hipGetLastError();
          } else if (getenv("ZKHIP_VM_JIT_LOG")) {
            size_t ls = 0;
            (void)R.log_size(prog, &ls);
            std::string log(ls + 1, '\0');
            if (ls) (void)R.log(prog, &log[0]);
            fprintf(stderr, "zkhip: row-program compilation failed (%d):\n%s\n", rc, log.c_str());
          }
          (void)R.destroy(&prog);
        }
      }
      it = g_jit_cache.emplace(key, n).first;
    }
    c = it->second;
  }
  if (c.failed) { set_error("eval_rows: no compiled kernel for this program (hiprtc missing, compilation failed, or unsupported shape)"); return ZKHIP_EINVAL; }
  // kernel arguments: the jit_args struct by value
  std::vector<unsigned char> args((size_t)(n_columns ? n_columns : 1) * 8 + 4 * 8 + 8 + 8, 0);
  size_t o = 0;
  for (uint32_t i = 0; i < n_columns; i++) { std::memcpy(args.data() + o, &d_columns[i], 8); o += 8; }
  if (n_columns == 0) o += 8;
  std::memcpy(args.data() + o, &d_consts, 8); o += 8;
  std::memcpy(args.data() + o, &d_pow_lo, 8); o += 8;
  std::memcpy(args.data() + o, &d_pow_hi, 8); o += 8;
  std::memcpy(args.data() + o, &d_out, 8); o += 8;
  const uint64_t rows = (uint64_t)1 << log_rows;
  std::memcpy(args.data() + o, &rows, 8); o += 8;
  const uint32_t acc = accumulate ? 1u : 0u;
  std::memcpy(args.data() + o, &acc, 4); o += 8;
  size_t arg_size = o;
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, args.data(), HIP_LAUNCH_PARAM_BUFFER_SIZE, &arg_size, HIP_LAUNCH_PARAM_END};
  const unsigned blocks = (unsigned)((rows + 255) / 256);
  if (hipModuleLaunchKernel(c.fn, blocks, 1, 1, 256, 1, 1, 0, stream, nullptr, config) != hipSuccess) {
    (void)hipGetLastError();
    set_error("eval_rows: launch of the compiled kernel failed");
    return ZKHIP_EHIP;
  }
  return ZKHIP_OK;
}

// the generated source / a compile-only run of it (hiprtc cross-compiles for gfx950 without a device): the CPU-side test of the generator
int row_vm_jit_source(const zkhip_vm_program* p, uint32_t n_columns, uint32_t log_rows, std::string* out) {
  *out = generate(p, n_columns, log_rows);
  if (out->empty()) { set_error("vm_jit: the generator does not take this program"); return ZKHIP_EINVAL; }
  return ZKHIP_OK;
}
int row_vm_jit_compile_only(const zkhip_vm_program* p, uint32_t n_columns, uint32_t log_rows, size_t* code_bytes) {
  const rtc_api& R = rtc();
  if (!R.ok) { set_error("vm_jit: libhiprtc.so is not available"); return ZKHIP_ENODEV; }
  std::string src;
  int rc = row_vm_jit_source(p, n_columns, log_rows, &src);
  if (rc != ZKHIP_OK) return rc;
  rtc_program prog = nullptr;
  if (R.create(&prog, src.c_str(), "zk_row_jit.hip", 0, nullptr, nullptr) != 0) { set_error("vm_jit: hiprtcCreateProgram failed"); return ZKHIP_EHIP; }
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
  const int crc = R.compile(prog, 3, opts);
  size_t cs = 0;
  if (crc == 0) (void)R.code_size(prog, &cs);
  if (crc != 0 || cs == 0) {
    size_t ls = 0;
    (void)R.log_size(prog, &ls);
    std::string log(ls + 1, '\0');
    if (ls) (void)R.log(prog, &log[0]);
    set_error("vm_jit: compilation failed (%d): %.400s", crc, log.c_str());
    (void)R.destroy(&prog);
    return ZKHIP_EHIP;
  }
  (void)R.destroy(&prog);
  if (code_bytes) *code_bytes = cs;
  return ZKHIP_OK;
}

void row_vm_jit_clear() {
  std::lock_guard<std::mutex> g(g_jit_mu);
  for (auto& kv : g_jit_cache) if (kv.second.mod) (void)hipModuleUnload(kv.second.mod);
  g_jit_cache.clear();
}

}  // namespace zkhip
