// The lookup argument's `permute_expression_pair` ([DEP] halo2-axiom halo2_proofs/src/plonk/lookup/prover.rs, reached from
// create_proof, /root/reference/aggregator/src/wrapper.rs:129).  Given the compressed input
// column A and table column S over the usable rows, the reference
//   * sorts A (by the canonical integer value of each element: `Ord for Fr` compares `to_repr()`)            -> A'
//   * at the first row of every run of equal values in A' puts that value into S' and takes one instance of it out of a
//     BTreeMap (value -> count) of the table (an input value missing from the table is an error),
//   * hands the map's leftover entries, in ascending order, to the remaining rows taken from the *end* (Vec::pop).
// Restated for a GPU as sorts, a binary search and two scans:
//   keys      canonical integers of A and S (one Montgomery multiply each)
//   sort      both key arrays (rocPRIM: radix sort of the low limbs when every key fits 64 bits -- range checks --, else a merge sort
//             with a 256-bit comparison; plain library sorts, not hot kernels)
//   mark      first-of-run rows of A'; each looks up its value in the sorted table (lower bound) and marks that instance used
//   scan      rank of every repeated row among the repeated rows; rank of every unused table instance among the unused ones
//   assign    S'[first row] = A'[row];  S'[repeated row of rank r] = unused instance of rank R - 1 - r   (R = number of repeated rows)
// Rows >= usable_rows (the blinding rows) are left to the caller.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_merge_sort.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <cstdint>
#include "fp29.hpp"
#include "fr_vec.hpp"
#include "zkhip_internal.hpp"

namespace zkhip {

struct key256 {   // canonical integer, little-endian 64-bit limbs
  uint64_t w[4];
};

struct key256_less {
  __host__ __device__ bool operator()(const key256& a, const key256& b) const {
    if (a.w[3] != b.w[3]) return a.w[3] < b.w[3];
    if (a.w[2] != b.w[2]) return a.w[2] < b.w[2];
    if (a.w[1] != b.w[1]) return a.w[1] < b.w[1];
    return a.w[0] < b.w[0];
  }
};

__device__ __forceinline__ bool key_eq(const key256& a, const key256& b) {
  return a.w[0] == b.w[0] && a.w[1] == b.w[1] && a.w[2] == b.w[2] && a.w[3] == b.w[3];
}

// Montgomery words x*2^256 -> the integer x (one multiply by 2^5 in the radix-2^261 domain).  *wide is set when a key does not fit 64
// bits: range-check lookups (the common case: values below 2^lookup_bits) then sort 8-byte keys with a radix sort instead.
__global__ void __launch_bounds__(256) k_lookup_keys(const uint32_t* __restrict__ in, size_t n, key256* __restrict__ keys, uint64_t* __restrict__ low,
                                                     uint32_t* __restrict__ wide) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  fe c;
#pragma unroll
  for (int k = 0; k < NL; k++) c.l[k] = Fr::FROM_EXT_CANON[k];
  uint32_t w[8];
  fe_pack(fe_canon_lt2p<Fr>(fe_mul<Fr>(load_ext(in, i), c)), w);
  key256 out;
#pragma unroll
  for (int k = 0; k < 4; k++) out.w[k] = (uint64_t)w[2 * k] | ((uint64_t)w[2 * k + 1] << 32);
  keys[i] = out;
  low[i] = out.w[0];
  if (out.w[1] | out.w[2] | out.w[3]) atomicOr(wide, 1u);
}

__global__ void __launch_bounds__(256) k_lookup_widen(const uint64_t* __restrict__ low, size_t n, key256* __restrict__ keys) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  key256 out;
  out.w[0] = low[i]; out.w[1] = 0; out.w[2] = 0; out.w[3] = 0;
  keys[i] = out;
}

// first-of-run flags of the sorted input; every first row claims the first instance of its value in the sorted table
__global__ void __launch_bounds__(256) k_lookup_mark(const key256* __restrict__ a, const key256* __restrict__ t, uint32_t n,
                                                     uint32_t* __restrict__ repeated, uint32_t* __restrict__ unused, uint32_t* __restrict__ error) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const key256 v = a[i];
  const bool first = i == 0 || !key_eq(v, a[i - 1]);
  repeated[i] = first ? 0u : 1u;
  if (!first) return;
  uint32_t lo = 0, hi = n;                     // lower bound of v in t
  const key256_less less;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (less(t[mid], v)) lo = mid + 1; else hi = mid;
  }
  if (lo < n && key_eq(t[lo], v)) unused[lo] = 0u;      // distinct values have distinct lower bounds: no two rows write one slot
  else atomicOr(error, 1u);
}

__global__ void __launch_bounds__(256) k_fill_u32(uint32_t* __restrict__ p, uint32_t n, uint32_t v) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// leftover[rank] = unused table instance, in ascending order
__global__ void __launch_bounds__(256) k_lookup_compact(const key256* __restrict__ t, const uint32_t* __restrict__ unused,
                                                        const uint32_t* __restrict__ unused_rank, uint32_t n, key256* __restrict__ leftover) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n && unused[j]) leftover[unused_rank[j]] = t[j];
}

__device__ __forceinline__ void store_key_as_fr(const key256& k, uint32_t* out, size_t i) {   // integer x -> Montgomery words x*2^256
  uint32_t w[8];
#pragma unroll
  for (int j = 0; j < 4; j++) { w[2 * j] = (uint32_t)k.w[j]; w[2 * j + 1] = (uint32_t)(k.w[j] >> 32); }
  fe r2;
#pragma unroll
  for (int j = 0; j < NL; j++) r2.l[j] = Fr::R2[j];
  const fe internal = fe_mul<Fr>(fe_unpack<0>(w), r2);       // x * 2^261
  uint32_t o[8];
  fe_to_ext<Fr>(internal, o);
  store_words(out + i * 8, o);
}

__global__ void __launch_bounds__(256) k_lookup_assign(const key256* __restrict__ a, const uint32_t* __restrict__ repeated,
                                                       const uint32_t* __restrict__ repeated_rank, const key256* __restrict__ leftover,
                                                       uint32_t n, uint32_t n_repeated, uint32_t* __restrict__ out_input,
                                                       uint32_t* __restrict__ out_table) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const key256 v = a[i];
  store_key_as_fr(v, out_input, i);
  // Vec::pop hands the ascending leftovers to the repeated rows from the last one backwards
  store_key_as_fr(repeated[i] ? leftover[n_repeated - 1 - repeated_rank[i]] : v, out_table, i);
}

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("%s failed: %s", #x, hipGetErrorString(e_)); return ZKHIP_EHIP; } } while (0)

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t sort_temp_bytes(size_t u) {
  size_t bytes = 0;
  (void)rocprim::merge_sort(nullptr, bytes, (key256*)nullptr, (key256*)nullptr, u, key256_less{});
  return bytes;
}
static size_t radix_temp_bytes(size_t u) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_keys(nullptr, bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, u);
  return bytes;
}
static size_t scan_temp_bytes(size_t u) {
  size_t bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, u, rocprim::plus<uint32_t>{});
  return bytes;
}

size_t lookup_permute_workspace_bytes(size_t u) {
  const size_t st = sort_temp_bytes(u), rt = radix_temp_bytes(u);
  return 5 * al256(u * sizeof(key256)) + 4 * al256(u * sizeof(uint64_t)) + 4 * al256((u + 1) * sizeof(uint32_t)) + al256(st > rt ? st : rt) +
         al256(scan_temp_bytes(u + 1)) + 512;
}

// d_input / d_table: n elements each; the first `u` rows are permuted into d_out_input / d_out_table (which may not alias the inputs)
int lookup_permute_device(const uint32_t* d_input, const uint32_t* d_table, size_t u, uint32_t* d_out_input, uint32_t* d_out_table,
                          void* ws, size_t ws_bytes, hipStream_t stream) {
  if (u == 0) return ZKHIP_OK;
  if (u >= (1ull << 31)) { set_error("lookup_permute: %zu rows is too many", u); return ZKHIP_EINVAL; }
  if (ws_bytes < lookup_permute_workspace_bytes(u)) { set_error("lookup_permute: workspace too small"); return ZKHIP_EINVAL; }
  char* p = (char*)ws;
  auto carve = [&](size_t bytes) { void* r = p; p += al256(bytes); return r; };
  key256* ka = (key256*)carve(u * sizeof(key256));
  key256* ks = (key256*)carve(u * sizeof(key256));
  key256* sa = (key256*)carve(u * sizeof(key256));
  key256* st = (key256*)carve(u * sizeof(key256));
  key256* leftover = (key256*)carve(u * sizeof(key256));
  uint32_t* repeated = (uint32_t*)carve((u + 1) * 4);
  uint32_t* unused = (uint32_t*)carve((u + 1) * 4);
  uint32_t* repeated_rank = (uint32_t*)carve((u + 1) * 4);
  uint32_t* unused_rank = (uint32_t*)carve((u + 1) * 4);
  uint64_t* la = (uint64_t*)carve(u * sizeof(uint64_t));
  uint64_t* ls = (uint64_t*)carve(u * sizeof(uint64_t));
  uint64_t* la_sorted = (uint64_t*)carve(u * sizeof(uint64_t));
  uint64_t* ls_sorted = (uint64_t*)carve(u * sizeof(uint64_t));
  size_t sort_bytes = sort_temp_bytes(u), radix_bytes = radix_temp_bytes(u), scan_bytes = scan_temp_bytes(u + 1);
  void* sort_tmp = carve(sort_bytes > radix_bytes ? sort_bytes : radix_bytes);
  void* scan_tmp = carve(scan_bytes);
  uint32_t* flags = (uint32_t*)carve(512);         // [0] error, [1] some key wider than 64 bits
  const uint32_t n = (uint32_t)u;
  const dim3 grid((unsigned)((u + 255) / 256)), grid1((unsigned)((u + 256) / 256)), block(256);
  HIPCHK(hipMemsetAsync(flags, 0, 512, stream));
  hipLaunchKernelGGL(k_lookup_keys, grid, block, 0, stream, d_input, u, ka, la, flags + 1);
  hipLaunchKernelGGL(k_lookup_keys, grid, block, 0, stream, d_table, u, ks, ls, flags + 1);
  uint32_t wide = 1;
  HIPCHK(hipMemcpyAsync(&wide, flags + 1, 4, hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  if (wide) {                                      // full-width values (theta-compressed multi-column lookups): 256-bit comparison sort
    HIPCHK(rocprim::merge_sort(sort_tmp, sort_bytes, ka, sa, u, key256_less{}, stream));
    HIPCHK(rocprim::merge_sort(sort_tmp, sort_bytes, ks, st, u, key256_less{}, stream));
  } else {                                         // every key fits 64 bits: radix sort of the low limbs
    HIPCHK(rocprim::radix_sort_keys(sort_tmp, radix_bytes, la, la_sorted, u, 0, 64, stream));
    HIPCHK(rocprim::radix_sort_keys(sort_tmp, radix_bytes, ls, ls_sorted, u, 0, 64, stream));
    hipLaunchKernelGGL(k_lookup_widen, grid, block, 0, stream, (const uint64_t*)la_sorted, u, sa);
    hipLaunchKernelGGL(k_lookup_widen, grid, block, 0, stream, (const uint64_t*)ls_sorted, u, st);
  }
  hipLaunchKernelGGL(k_fill_u32, grid1, block, 0, stream, unused, n + 1, 1u);
  HIPCHK(hipMemsetAsync(repeated + u, 0, 4, stream));
  hipLaunchKernelGGL(k_lookup_mark, grid, block, 0, stream, (const key256*)sa, (const key256*)st, n, repeated, unused, flags);
  // exclusive scans over u + 1 entries: the last entry of each is the total
  HIPCHK(rocprim::exclusive_scan(scan_tmp, scan_bytes, repeated, repeated_rank, 0u, u + 1, rocprim::plus<uint32_t>{}, stream));
  HIPCHK(rocprim::exclusive_scan(scan_tmp, scan_bytes, unused, unused_rank, 0u, u + 1, rocprim::plus<uint32_t>{}, stream));
  uint32_t host[3] = {0, 0, 0};
  HIPCHK(hipMemcpyAsync(&host[0], flags, 4, hipMemcpyDeviceToHost, stream));
  HIPCHK(hipMemcpyAsync(&host[1], repeated_rank + u, 4, hipMemcpyDeviceToHost, stream));
  HIPCHK(hipMemcpyAsync(&host[2], unused_rank + u, 4, hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  // `unused` had its sentinel entry u set to 1 by the fill: the scan total at u does not include it
  if (host[0] != 0 || host[1] != host[2]) {
    set_error("lookup_permute: an input value is missing from the table (the reference returns Error::ConstraintSystemFailure)");
    return ZKHIP_EINVAL;
  }
  hipLaunchKernelGGL(k_lookup_compact, grid, block, 0, stream, (const key256*)st, (const uint32_t*)unused, (const uint32_t*)unused_rank, n, leftover);
  hipLaunchKernelGGL(k_lookup_assign, grid, block, 0, stream, (const key256*)sa, (const uint32_t*)repeated, (const uint32_t*)repeated_rank,
                     (const key256*)leftover, n, host[1], d_out_input, d_out_table);
  HIPCHK(hipGetLastError());
  return ZKHIP_OK;
}

}  // namespace zkhip
