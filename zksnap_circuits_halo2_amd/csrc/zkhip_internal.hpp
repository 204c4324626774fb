// Internal declarations shared by the translation units of libzkhip.so (not installed).
#pragma once
#include <string>
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include "../../include/zkhip.h"

namespace zkhip {

void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

// profiler (capi.hip): when enabled, device paths drop named HIP events on their stream between phases
void prof_begin(hipStream_t stream);
void prof_mark(hipStream_t stream, const char* name);

// msm.hip
int g1_gen_walk_device(const uint32_t t0_ext[8], const uint32_t d_ext[8], size_t n, uint32_t* d_out, void* ws, size_t ws_bytes, hipStream_t stream);
size_t g1_gen_walk_workspace(size_t n);
size_t g1_fixed_base_table_bytes();
size_t g1_fixed_base_workspace(size_t n);
int g1_fixed_base_table_build(uint32_t* d_table, void* ws, size_t ws_bytes, hipStream_t stream);
int g1_fixed_base_mul_device(const uint32_t* d_scalars, size_t n, const uint32_t* d_table, uint32_t* d_out, void* ws, size_t ws_bytes, hipStream_t stream);
int msm_pick_window(size_t n);
size_t msm_workspace_bytes(size_t n, int c, bool prepared = false, size_t batch = 1, size_t xyzz_bytes = 144);
// what the curve-specific kernels (accumulate, combine, bucket reduction) need once the scalars are sorted into tasks
struct msm_tasks_view {
  int c, W, WB;                    // window bits, windows, bucket sets
  uint32_t B, NB, task_shift;      // buckets per set, buckets in all, log2 task length
  size_t max_tasks, nk;
  const void* tasks;               // task_t {bucket, start, len}
  uint32_t* ntasks;                // device: number of tasks
  uint32_t* max_parts;             // device: largest task count of one bucket
  const uint4* order;              // execution order records (task | 0x80000000 + bucket, start, len, first reference)
  const uint32_t* sorted;          // point references (bit 31 = negate), bucket by bucket
  const uint32_t* task_off;        // first task of every bucket (NB + 1)
  uint32_t *partials, *pyrA, *pyrB, *winsum;   // work arrays of xyzz_bytes-sized points
  int combine_lanes;               // G1 combine step: lanes per bucket, partials it sums per bucket, and the list of buckets with more
  uint32_t seq_parts;
  const uint32_t* heavy_count;
  const void* heavy;               // uint2 (bucket, slice) entries
  uint32_t heavy_cap;
  uint32_t* heavy_done;
  uint32_t* endo;                  // G1 general path with GLV digits: room for the endomorphism images of the n bases
};
// glv (general path of either curve): the scalars are split with the curve's endomorphism (msm.hip k_digits_glv): the sort sees 2 n points and
// ceil(128 / c) windows, point references >= n mean "the image of point ref - n"
bool msm_uses_glv(bool prepared, size_t batch, size_t xyzz_bytes);
int msm_glv_windows(int c);
int msm_build_tasks(const uint32_t* d_scalars, size_t n, size_t batch, size_t scalar_stride, int c, bool shared_buckets, uint32_t ref_base, uint32_t ref_stride,
                    size_t xyzz_bytes, void* ws, size_t ws_bytes, hipStream_t stream, msm_tasks_view* out, bool glv = false);
// msm_g2.hip
size_t msm_g2_workspace_bytes(size_t n);
int msm_g2_device(const uint32_t* d_scalars, const uint32_t* d_bases, size_t n, uint32_t* d_out, void* ws, size_t ws_bytes, hipStream_t stream);
int test_g2_op(int op, const uint32_t* d_a, const uint32_t* d_b, uint32_t* d_out, size_t n, hipStream_t stream);
struct prepared_bases {   // table[w * n + i] = 2^(o_w) * P_i with o_w the bit offset of window w (c w; balanced widths above 16 bits: msm.hip win_off)
  uint32_t* table;
  size_t n;
  int c, W;
  uint32_t* direct = nullptr;   // small sets only (msm.hip "Direct tables"): [i][w < 32][m - 1 < 128] = m 2^(8 w) P_i, or null
};
// bucket-free MSM over a direct table (single vector, n <= 2^15 by default)
size_t direct_table_bytes(size_t n);
size_t msm_direct_workspace_bytes(size_t n);
int prepare_direct_table(prepared_bases* pb, const uint32_t* d_bases, hipStream_t stream);
int msm_g1_direct(const uint32_t* d_scalars, size_t n, const prepared_bases* pb, size_t off, uint32_t* d_out, void* ws, size_t ws_bytes, hipStream_t stream);
int msm_pick_window_prepared(size_t n);
int msm_g1_device(const uint32_t* d_scalars, const uint32_t* d_bases, size_t n, uint32_t* d_out, void* ws, size_t ws_bytes,
                  int c_override, hipStream_t stream, const prepared_bases* prepared = nullptr, size_t prepared_off = 0, size_t batch = 1,
                  size_t scalar_stride = 0);
// chunked prepared MSM: pieces of at most chunk_cap scalars, sorted and accumulated one by one into one bucket set, one reduction at the end
size_t msm_chunk_workspace_bytes(size_t chunk_cap, int c);
int msm_chunk_add(const uint32_t* d_scalars, size_t n, const prepared_bases* pb, size_t pb_off, size_t chunk_cap, bool first, void* ws, size_t ws_bytes, hipStream_t stream);
int msm_chunk_finish(const prepared_bases* pb, size_t chunk_cap, uint32_t* d_out, void* ws, size_t ws_bytes, hipStream_t stream);
// n_whole: size of the array this set is a shard of (0 = n): the direct table is meant for SMALL base sets, not for the shards of a large one
int prepare_bases_device(const uint32_t* d_bases, size_t n, hipStream_t stream, prepared_bases** out, int c_override = 0, size_t n_whole = 0);
void release_prepared(prepared_bases* pb);
// d_out[b] = sum over i < m of d_in[i * count + b], b < count (count = 1: the plain fold of m points)
int sum_jacobian_device(const uint32_t* d_in, int m, uint32_t* d_out, hipStream_t stream, size_t count = 1);
size_t g1_batch_normalize_workspace(size_t n);
int g1_batch_normalize_device(const uint32_t* d_in, size_t n, uint32_t* d_out, void* ws, size_t ws_bytes, hipStream_t stream);
size_t g1_fft_workspace(size_t n);
int g1_fft_device(const uint32_t* d_in, int in_format, uint32_t* d_out, int out_format, uint32_t log_n, const uint32_t omega_ext[8],
                  const uint32_t* scale_ext, void* ws, size_t ws_bytes, hipStream_t stream);

// ntt.hip
int ntt_passes(uint32_t L);
int ntt_transform(const uint32_t* d_in, uint32_t in_len, uint32_t in_stride, uint32_t* d_out, uint32_t out_len, uint32_t out_stride,
                  uint32_t batch, uint32_t L, const uint32_t omega_ext[8], const uint32_t* in_scale, uint32_t in_period,
                  const uint32_t* out_scale, uint32_t out_period, uint32_t* tmp0, uint32_t* tmp1, hipStream_t stream);
int fr_mul_periodic_device(uint32_t* d_a, size_t n, const uint32_t* d_table_ext, uint32_t period, hipStream_t stream);
int fr_scale_device(uint32_t* d_a, size_t n, const uint32_t scale_ext[8], hipStream_t stream);
void ntt_clear_cache();

// Small host arrays of a `_device` call (lists of column addresses, coefficients) on their way to device memory WITHOUT waiting for the stream:
// copied into a pinned slot first (the caller's memory is free when the call returns), then from there with hipMemcpyAsync.  Eight slots used in
// turn, an event per slot: the host runs up to eight uploads ahead of the stream before it has to wait for the oldest.  Owned by the caller's
// scratch set.  Anything larger than a slot -- or a null ring -- takes the plain path: copy from the caller's memory and synchronise the stream.
struct arg_ring {
  static constexpr int SLOTS = 8;
  static constexpr size_t SLOT_BYTES = 32768;
  void* host = nullptr;
  hipEvent_t ev[SLOTS] = {};
  bool used[SLOTS] = {};
  int turn = 0;
  int upload(void* dst, const void* src, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return ZKHIP_OK;
    if (bytes > SLOT_BYTES) return plain(dst, src, bytes, stream);
    if (!host && hipHostMalloc(&host, SLOTS * SLOT_BYTES, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); host = nullptr; return plain(dst, src, bytes, stream); }
    const int s = turn;
    turn = (turn + 1) % SLOTS;
    if (!ev[s] && hipEventCreateWithFlags(&ev[s], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ev[s] = nullptr; return plain(dst, src, bytes, stream); }
    if (used[s] && hipEventSynchronize(ev[s]) != hipSuccess) { set_error("arg_ring: event wait failed"); return ZKHIP_EHIP; }
    void* slot = (char*)host + (size_t)s * SLOT_BYTES;
    std::memcpy(slot, src, bytes);
    if (hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, stream) != hipSuccess || hipEventRecord(ev[s], stream) != hipSuccess) { set_error("arg_ring: copy failed"); return ZKHIP_EHIP; }
    used[s] = true;
    return ZKHIP_OK;
  }
  static int plain(void* dst, const void* src, size_t bytes, hipStream_t stream) {
    if (hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { set_error("upload of call arguments failed"); return ZKHIP_EHIP; }
    return ZKHIP_OK;
  }
  void release() {
    for (int i = 0; i < SLOTS; i++) if (ev[i]) { if (used[i]) (void)hipEventSynchronize(ev[i]); (void)hipEventDestroy(ev[i]); ev[i] = nullptr; used[i] = false; }
    if (host) { (void)hipHostFree(host); host = nullptr; }
  }
};
static inline int upload_args(arg_ring* ring, void* dst, const void* src, size_t bytes, hipStream_t stream) {
  return ring ? ring->upload(dst, src, bytes, stream) : arg_ring::plain(dst, src, bytes, stream);
}

// poly.hip
size_t poly_workspace_bytes(size_t n);
int fr_eval_polynomial_device(const uint32_t* d_a, size_t n, const uint32_t x_host[8], uint32_t* d_result, void* ws, size_t ws_bytes, hipStream_t stream);
size_t poly_batch_workspace_bytes(size_t n, size_t count);
int fr_eval_polynomial_batch_device(const void* const* d_polys_host, size_t count, size_t n, const uint32_t x_host[8], uint32_t* d_results,
                                    void* ws, size_t ws_bytes, hipStream_t stream, arg_ring* ring = nullptr);
int fr_kate_division_device(const uint32_t* d_a, size_t n, const uint32_t b_host[8], uint32_t* d_q, void* ws, size_t ws_bytes, hipStream_t stream);
int fr_prefix_product_device(const uint32_t* d_v, size_t n, uint32_t* d_out, void* ws, size_t ws_bytes, hipStream_t stream);
int fr_batch_invert_device(uint32_t* d_a, size_t n, void* ws, size_t ws_bytes, hipStream_t stream);
size_t lincomb_workspace_bytes(size_t count, size_t n);
int fr_linear_combination_device(const void* const* d_cols_host, const uint32_t* coeffs_host, size_t count, size_t n, uint32_t* d_out, void* ws, size_t ws_bytes,
                                 hipStream_t stream, arg_ring* ring = nullptr);
size_t perm_workspace_bytes(uint32_t nperm, uint32_t chunk, uint32_t log_n);
int fr_permutation_products_device(const void* const* d_values_host, const void* const* d_sigmas_host, uint32_t nperm, uint32_t chunk, uint32_t log_n,
                                   size_t usable, const uint32_t beta[8], const uint32_t gamma[8], const uint32_t delta[8], const uint32_t omega[8],
                                   uint32_t* d_z, void* ws, size_t ws_bytes, hipStream_t stream, arg_ring* ring = nullptr);

// rowvm.hip
int row_vm_validate(const zkhip_vm_program* p, uint32_t n_columns, uint32_t log_rows, int accumulate);
size_t row_vm_workspace_bytes(const zkhip_vm_program* p, uint32_t n_columns, uint32_t log_rows);
// Pinned host staging for the program blob of a row program (two buffers, used alternately; an event per buffer says when the copy that read
// it has completed).  With it row_vm_device never waits for the stream: the host builds program i + 1 while program i runs.  Owned by the
// caller's scratch set (one user at a time); nullptr = build the blob in a local buffer and synchronise the stream after the copy.
struct vm_staging {
  void* host[2] = {nullptr, nullptr};
  size_t cap[2] = {0, 0};
  hipEvent_t copied[2] = {nullptr, nullptr};
  int turn = 0;
  void release();
};
size_t row_vm_multi_workspace_bytes(const zkhip_vm_program* progs, uint32_t n_progs, uint32_t n_columns, uint32_t log_rows);
int row_vm_device_multi(const zkhip_vm_program* progs, uint32_t n_progs, const void* const* d_columns, uint32_t n_columns, uint32_t log_rows, uint32_t* const* d_outs,
                        void* ws, size_t ws_bytes, hipStream_t stream, vm_staging* staging = nullptr);
// rowvm_jit.hip: row programs compiled at run time with hiprtc (straight-line code per program shape, cached per device); row_vm_device
// tries it first for short programs over many rows and runs the interpreter otherwise
bool row_vm_jit_wanted(const zkhip_vm_program* p, uint32_t n_columns, uint32_t log_rows);
int row_vm_jit_launch(const zkhip_vm_program* p, const void* const* d_columns, uint32_t n_columns, uint32_t log_rows, int accumulate, const uint32_t* d_consts,
                      const uint32_t* d_pow_lo, const uint32_t* d_pow_hi, uint32_t* d_out, hipStream_t stream);
void row_vm_jit_clear();
int row_vm_jit_source(const zkhip_vm_program* p, uint32_t n_columns, uint32_t log_rows, std::string* out);
int row_vm_jit_compile_only(const zkhip_vm_program* p, uint32_t n_columns, uint32_t log_rows, size_t* code_bytes);
int row_vm_device(const zkhip_vm_program* p, const void* const* d_columns, uint32_t n_columns, uint32_t log_rows, int accumulate,
                  uint32_t* d_out, void* ws, size_t ws_bytes, hipStream_t stream, vm_staging* staging = nullptr);
int fr_pointwise_mul_device(const uint32_t* d_a, const uint32_t* d_b, size_t n, uint32_t* d_out, hipStream_t stream);
int fr_gather_mul_device(const uint32_t* d_a, uint32_t a_len, const uint32_t* d_ia, const uint32_t* d_b, uint32_t b_len, const uint32_t* d_ib, size_t n,
                         uint32_t* d_out, hipStream_t stream);

// lookup.hip
size_t lookup_permute_workspace_bytes(size_t usable_rows);
int lookup_permute_device(const uint32_t* d_input, const uint32_t* d_table, size_t usable_rows, uint32_t* d_out_input, uint32_t* d_out_table,
                          void* ws, size_t ws_bytes, hipStream_t stream);

// serde.hip
int g1_compress_device(const uint32_t* d_points, size_t n, uint32_t* d_out, int layout, hipStream_t stream);
int g1_decompress_device(const uint32_t* d_in, size_t n, uint32_t* d_points, int layout, unsigned long long* d_first_bad, hipStream_t stream);

// selftest.hip
int test_field_op(int field, int op, const uint32_t* d_a, const uint32_t* d_b, uint32_t* d_out, size_t n, hipStream_t stream);
int g1_check_points_device(const uint32_t* d_points, size_t n, unsigned long long* d_first_bad, hipStream_t stream);
int test_g1_op(int op, const uint32_t* d_a, const uint32_t* d_b, uint32_t* d_out, size_t n, hipStream_t stream);

}  // namespace zkhip
