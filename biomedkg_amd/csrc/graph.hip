// Graph preparation on device: COO (int64 edge_index, as handed over by the reference's batch
// contract -- SURVEY.md 8b) -> destination-sorted int32 CSR pieces, GCN normalisation weights
// (PyG gcn_norm / add_remaining_self_loops semantics) and R-GCN (relation, destination) segments.
// Stable LSD radix sort (rocPRIM) keeps equal-key edges in input order, so every segment sum that
// follows has a fixed, input-determined order (bitwise reproducible; no atomics anywhere here).
#include <cstring>

#include "common.h"

// rocPRIM's one-sweep radix sort zeroes its digit histograms and look-back states with hipMemsetAsync, and a memset node of
// a hipGraph re-executes wrongly on this stack (common.h: stale words from the second replay on; a captured 1.5 M-key
// radix sort page-faulted on its second replay).  rocPRIM is header-only: inside THIS translation unit its two memsets are
// routed to the library's fill kernel -- a kernel node replays like any other -- so captured steps sort with the radix sort
// too (a NeighborLoader batch's 221 k-edge by-source sort: ~6 graph nodes and 43 us instead of the merge sort's ~20 and
// 116).  (hip_runtime_api.h is already in: the macro meets only rocPRIM's calls.)
namespace bkg {
inline hipError_t fill_bytes_async(void* dst, int value, size_t bytes, hipStream_t s) {
  if (bytes == 0) return hipSuccess;
  if (bytes % 4 == 0 && reinterpret_cast<uintptr_t>(dst) % 4 == 0) {
    const uint32_t b = (uint32_t)(value & 0xff), v = b | (b << 8) | (b << 16) | (b << 24);
    fill_u32(dst, v, (int64_t)(bytes / 4), s);
    return hipGetLastError();
  }
  return (hipMemsetAsync)(dst, value, bytes, s);      // (not a shape rocPRIM's sorts and scans ask for)
}
}  // namespace bkg
#define hipMemsetAsync(dst, value, bytes, stream) ::bkg::fill_bytes_async((dst), (value), (bytes), (stream))

#include <rocprim/block/block_radix_sort.hpp>
#include <rocprim/device/device_merge_sort.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#undef hipMemsetAsync

#ifndef BKG_SMALL_SORT
#define BKG_SMALL_SORT 1      // key sets of <= 8192 entries: one single-workgroup launch instead of a device sort
#endif

namespace bkg {

__global__ void make_keys_kernel(const int64_t* __restrict__ lo, const int64_t* __restrict__ hi, int64_t n_lo,
                                 uint32_t* __restrict__ keys, uint32_t* __restrict__ iota, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t k = lo[i] + (hi ? hi[i] * n_lo : 0);
  keys[i] = (uint32_t)k;
  iota[i] = (uint32_t)i;
}

// Up to 1024 * IPT keys in ONE launch of one workgroup: keys made, sorted (stable LSD radix over the low `bits` bits, in
// LDS) and written with their original positions.  The mini-batch steps sort seven key sets of a few thousand entries per
// step; rocPRIM's device sorts are four to six launches each, and a replayed step is bound by its node count.  Any stable
// sort yields the same (keys, permutation): bit-equal with the device sorts.  Pads (positions >= n) carry all-ones keys and
// the highest positions, so they stay behind every real key.
// rowptr (nullable): rowptr[q] = first position with key >= q * stride, q < n_q -- the lower-bound pass that follows
// every one of these sorts, over the sorted keys while they are still in LDS.
template <int IPT>
__global__ __launch_bounds__(1024) void small_sort_kernel(const int64_t* __restrict__ lo, const int64_t* __restrict__ hi,
                                                          int64_t n_lo, int n, int bits, uint32_t* __restrict__ sorted_keys,
                                                          uint32_t* __restrict__ perm, int64_t n_q, uint32_t stride,
                                                          int32_t* __restrict__ rowptr) {
  using sort_t = rocprim::block_radix_sort<uint32_t, 1024, IPT, uint32_t>;
  __shared__ typename sort_t::storage_type storage;
  static_assert(sizeof(typename sort_t::storage_type) >= 1024 * IPT * sizeof(uint32_t), "the sorted keys are parked in the sort's LDS");
  uint32_t keys[IPT], vals[IPT];
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int idx = threadIdx.x * IPT + i;
    keys[i] = 0xFFFFFFFFu;
    vals[i] = (uint32_t)idx;
    if (idx < n) keys[i] = (uint32_t)(lo[idx] + (hi ? hi[idx] * n_lo : 0));
  }
  sort_t().sort(keys, vals, storage, 0, bits);
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int idx = threadIdx.x * IPT + i;
    if (idx < n) {
      sorted_keys[idx] = keys[i];
      perm[idx] = vals[i];
    }
  }
  if (rowptr == nullptr) return;
  __syncthreads();                                          // everybody is out of the sort's storage
  uint32_t* sk = reinterpret_cast<uint32_t*>(&storage);
#pragma unroll
  for (int i = 0; i < IPT; ++i) sk[threadIdx.x * IPT + i] = keys[i];
  __syncthreads();
  for (int64_t q = threadIdx.x; q < n_q; q += 1024) {
    const uint64_t want = (uint64_t)q * stride;
    int a = 0, b = n;
    while (a < b) {
      const int mid = (a + b) >> 1;
      if ((uint64_t)sk[mid] < want) a = mid + 1; else b = mid;
    }
    rowptr[q] = a;
  }
}
constexpr int SMALL_SORT_MAX = 8192;

// out[s] = first position p in sorted[0..n) with sorted[p] >= s * stride   (s in [0, n_q))
// The 2 T incidences cat(head, tail) of a triplet set as (key, payload) pairs for the (node, role)-sorted one-pass walk
// (csrc/kge_onepass_mx.hip): key = 2 * node + role (role 0: the node is the head), payload = other endpoint | relation << 24.
__global__ void role_incidence_keys_kernel(const int64_t* __restrict__ head, const int64_t* __restrict__ tail,
                                           const int64_t* __restrict__ rel, int64_t T, uint32_t* __restrict__ keys,
                                           uint32_t* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * T) return;
  const int role = i >= T;
  const int64_t t = role ? i - T : i;
  const int64_t h = head[t], tl = tail[t];
  const uint32_t node = (uint32_t)(role ? tl : h), oth = (uint32_t)(role ? h : tl);
  keys[i] = 2u * node + (uint32_t)role;
  vals[i] = oth | ((uint32_t)rel[t] << 24);
}

__global__ void lower_bound_kernel(const uint32_t* __restrict__ sorted, int64_t n, int64_t n_q, uint32_t stride,
                                   int32_t* __restrict__ out) {
  int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_q) return;
  uint64_t q = (uint64_t)s * stride;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if ((uint64_t)sorted[mid] < q) lo = mid + 1; else hi = mid;
  }
  out[s] = (int32_t)lo;
}

__global__ void gather_i64_to_i32_kernel(const int64_t* __restrict__ src, const uint32_t* __restrict__ idx,
                                         int32_t* __restrict__ out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int32_t)src[idx[i]];
}

__global__ void gather_i32_kernel(const int32_t* __restrict__ src, const int32_t* __restrict__ idx,
                                  int32_t* __restrict__ out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = src[idx[i]];
}

__global__ void gather_f32_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                  float* __restrict__ out, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = src[idx[i]];
}

// ---- GCN normalisation (PyG gcn_norm) over the dst-sorted CSR.
// deg[i] = 1 (the one unit self loop) + #{in-edges k of i : src != i and kept}.
// Edge-parallel with integer atomics (exact, order-independent): a row-parallel count would run as
// long as the largest in-degree (10^4-10^5 on PrimeKG-like graphs).  `cnt` aliases the dis buffer.
// One thread per DST-SORTED entry: equal targets are adjacent, so each wave first adds up its runs with
// a segmented shuffle scan and only the last lane of a run issues an atomic (a 4*10^4-in-edge hub costs
// ~700 atomics instead of 4*10^4 serialised ones).
__global__ void gcn_degree_count_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                        const int32_t* __restrict__ perm, const uint8_t* __restrict__ keep,
                                        int n_nodes, int64_t nnz, int32_t* __restrict__ cnt) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool valid = k < nnz;
  int row = -1, v = 0;
  if (valid) {
    int lo = 0, hi = n_nodes;  // largest row with rowptr[row] <= k
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (rowptr[mid] <= k) lo = mid; else hi = mid;
    }
    row = lo;
    v = (col[k] != row && (!keep || keep[perm[k]])) ? 1 : 0;
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(v, off, 64), rr = __shfl_up(row, off, 64);
    if (lane >= off && rr == row) v += t;
  }
  const int next = __shfl_down(row, 1, 64);
  if (valid && v && (lane == 63 || next != row)) atomicAdd(&cnt[row], v);
}

__global__ void gcn_degree_finish_kernel(int n, float* __restrict__ dis, float* __restrict__ self_w) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = reinterpret_cast<const int32_t*>(dis)[i];
  const float d = 1.0f / sqrtf((float)(c + 1));
  dis[i] = d;
  self_w[i] = d * d;
}

// w_edge[e] (original edge order) = kept && src != dst ? dis[src] * dis[dst] : 0
__global__ void gcn_edge_weight_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                       const uint8_t* __restrict__ keep, const float* __restrict__ dis, int64_t n,
                                       float* __restrict__ w_edge) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  int64_t s = src[e], d = dst[e];
  w_edge[e] = (s != d && (!keep || keep[e])) ? dis[s] * dis[d] : 0.f;
}

// ---- R-GCN (relation, dst) segments over edges sorted by key = rel * N + dst.
__global__ void segment_flags_kernel(const uint32_t* __restrict__ sorted, int64_t n, int32_t* __restrict__ flag) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[i] = (i == 0 || sorted[i] != sorted[i - 1]) ? 1 : 0;
}

// seg_of[k] = inclusive_scan(flag)[k] - 1 ; at each segment head record start + key.
__global__ void segment_fill_kernel(const uint32_t* __restrict__ sorted, const int32_t* __restrict__ scan, int64_t n,
                                    uint32_t n_lo, int32_t* __restrict__ seg_of, int32_t* __restrict__ seg_ptr,
                                    int32_t* __restrict__ seg_dst, int32_t* __restrict__ seg_rel,
                                    int32_t* __restrict__ n_seg_out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int s = scan[i] - 1;
  seg_of[i] = s;
  if (i == 0 || sorted[i] != sorted[i - 1]) {
    seg_ptr[s] = (int32_t)i;
    seg_dst[s] = (int32_t)(sorted[i] % n_lo);
    seg_rel[s] = (int32_t)(sorted[i] / n_lo);
  }
  if (i == n - 1) {
    seg_ptr[s + 1] = (int32_t)n;
    *n_seg_out = s + 1;
  }
}

// per sorted edge: w[k] = 1 / (size of its segment)
__global__ void segment_inv_count_kernel(const int32_t* __restrict__ seg_of, const int32_t* __restrict__ seg_ptr,
                                         int64_t n, float* __restrict__ w) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int s = seg_of[i];
  w[i] = 1.0f / (float)(seg_ptr[s + 1] - seg_ptr[s]);
}

// sorted-edge attributes back in ORIGINAL edge order: seg_of_edge[perm[k]] = seg_of[k] (int64: it becomes the column
// of the source-side CSR), w_edge[perm[k]] = inv_cnt[k]
__global__ void edge_order_kernel(const uint32_t* __restrict__ perm, const int32_t* __restrict__ seg_of,
                                  const float* __restrict__ inv_cnt, int64_t n, int64_t* __restrict__ seg_of_edge,
                                  float* __restrict__ w_edge) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t e = perm[k];
  seg_of_edge[e] = seg_of[k];
  w_edge[e] = inv_cnt[k];
}

// fixed segment count (padded mini-batches): every slot starts out as an EMPTY segment of the last relation whose
// target is spread over the nodes; segment_fill_kernel then overwrites the real ones
__global__ void segment_prefill_kernel(int64_t n, int32_t n_nodes, int32_t last_rel, int32_t* __restrict__ seg_ptr,
                                       int32_t* __restrict__ seg_dst, int32_t* __restrict__ seg_rel) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  seg_ptr[i] = (int32_t)n;
  if (i < n) { seg_dst[i] = (int32_t)(i % n_nodes); seg_rel[i] = last_rel; }
}

static inline dim3 grid1(int64_t n, int b = 256) { return dim3((unsigned)((n + b - 1) / b)); }


// ---- a dropout_edge view as CSRs of its own (reference biomedkg/model/gcl.py:42-43,76: PyG dropout_edge returns
// edge_index[:, keep], so GCNConv -- gcn_norm and the propagate -- runs over the KEPT edges only).  The masked form (the
// graph's CSRs, weight 0 on dropped entries) gathers every dropped edge's row all the same: 40 % of a propagate's bytes at
// p = 0.4.  Here both CSRs of the graph (rows = targets, rows = sources) are compacted to the view's entries -- kept and
// not a self loop (add_remaining_self_loops replaces those by the unit loop) -- and weighted in ONE pass: flags of both
// laid end to end -> one exclusive scan -> row pointers read off the scan, whose differences ARE the view's in-degrees
// (deg^-1/2, the loop weight) -> scatter of the column ids with w = deg^-1/2[row] deg^-1/2[col].  No host read; arrays
// keep the graph's length.
__global__ __launch_bounds__(256) void gcn_view_flag_kernel(const int32_t* __restrict__ col_d, const int32_t* __restrict__ perm_d,
                                                            const int32_t* __restrict__ row_d, const int32_t* __restrict__ col_s,
                                                            const int32_t* __restrict__ perm_s, const int32_t* __restrict__ row_s,
                                                            const uint8_t* __restrict__ keep, int64_t nnz,
                                                            int32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * nnz) return;
  const bool second = i >= nnz;
  const int64_t k = second ? i - nnz : i;
  const int32_t* col = second ? col_s : col_d;
  const int32_t* perm = second ? perm_s : perm_d;
  const int32_t* row = second ? row_s : row_d;
  flag[i] = (keep[perm[k]] && col[k] != row[k]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void gcn_view_rows_kernel(const int32_t* __restrict__ rowptr_d, const int32_t* __restrict__ rowptr_s,
                                                            const int32_t* __restrict__ flag, const int32_t* __restrict__ offs,
                                                            int n, int64_t nnz, int32_t* __restrict__ rowptr_out_d,
                                                            int32_t* __restrict__ rowptr_out_s, float* __restrict__ dis,
                                                            float* __restrict__ self_w, int32_t* __restrict__ counts) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r > n) return;
  const int32_t total = nnz ? offs[2 * nnz - 1] + flag[2 * nnz - 1] : 0;
  const int32_t base = nnz ? offs[nnz] : 0;                       // entries the first (target-side) CSR keeps
  auto at_d = [&](int q) { const int32_t p = rowptr_d[q]; return p < nnz ? offs[p] : base; };
  auto at_s = [&](int q) { const int32_t p = rowptr_s[q]; return (p < nnz ? offs[nnz + p] : total) - base; };
  const int32_t beg = at_d(r);
  rowptr_out_d[r] = beg;
  rowptr_out_s[r] = at_s(r);
  if (r < n) {
    const float d = 1.0f / sqrtf((float)(at_d(r + 1) - beg + 1));  // kept in-edges + the unit loop (as gcn_degree_finish_kernel)
    dis[r] = d;
    self_w[r] = d * d;
  }
  if (r == 0) { counts[0] = base; counts[1] = total - base; }
}

__global__ __launch_bounds__(256) void gcn_view_scatter_kernel(const int32_t* __restrict__ col_d, const int32_t* __restrict__ row_d,
                                                               const int32_t* __restrict__ col_s, const int32_t* __restrict__ row_s,
                                                               const int32_t* __restrict__ flag, const int32_t* __restrict__ offs,
                                                               const float* __restrict__ dis, int64_t nnz,
                                                               int32_t* __restrict__ col_out_d, float* __restrict__ w_out_d,
                                                               int32_t* __restrict__ col_out_s, float* __restrict__ w_out_s) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * nnz || !flag[i]) return;
  const bool second = i >= nnz;
  const int64_t k = second ? i - nnz : i;
  const int32_t c = second ? col_s[k] : col_d[k], r = second ? row_s[k] : row_d[k];
  const int32_t pos = offs[i] - (second ? offs[nnz] : 0);
  const float w = dis[r] * dis[c];
  if (second) { col_out_s[pos] = c; w_out_s[pos] = w; } else { col_out_d[pos] = c; w_out_d[pos] = w; }
}

// ---- the entries of ONE weighted CSR whose weight is not zero, as a CSR of its own over the same rows (a dropout_edge view
// of a node-partitioned graph: its CSRs list table slots and carry the self loops as entries, the weights come from the
// global degrees -- biomedkg_amd/partition.py --, so the view is cut out of the weighted form instead of built from a mask)
__global__ __launch_bounds__(256) void csr_flag_kernel(const float* __restrict__ w, int64_t nnz, int32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < nnz) flag[i] = w[i] != 0.f ? 1 : 0;
}

__global__ __launch_bounds__(256) void csr_compact_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                          const float* __restrict__ w, int64_t n_rows, int64_t nnz,
                                                          const int32_t* __restrict__ flag, const int32_t* __restrict__ offs,
                                                          int32_t* __restrict__ rowptr_out, int32_t* __restrict__ col_out,
                                                          float* __restrict__ w_out, int32_t* __restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int32_t total = nnz ? offs[nnz - 1] + flag[nnz - 1] : 0;
  if (i < nnz && flag[i]) {
    col_out[offs[i]] = col[i];
    w_out[offs[i]] = w[i];
  }
  if (i <= n_rows) {
    const int32_t r = rowptr[i];
    rowptr_out[i] = r < nnz ? offs[r] : total;
  }
  if (i == 0) *count = total;
}

}  // namespace bkg

using namespace bkg;

// ----------------------------------------------------------------------------- sort
// The one-sweep radix sort as tuned on gfx950 for this library's (uint32 key, uint32 position) pairs
// (scripts/dev_sort_configs.hip; 8 M pairs of 17-bit keys -- the fresh negatives' incidence lists: 227 -> 142 us):
// 9-bit digits (node ids of the bench graph take 17 bits: TWO passes instead of three of 8), 1024 x 8 items per block with
// the match ranking, and rocPRIM's merge sort only below 128 k pairs (its default hands everything up to 1 M pairs to the
// merge sort: 162 us for 1 M pairs against 52 us here, 116 against 43 us for 300 k -- the mini-batch structures' sizes).
using SortConfig = rocprim::radix_sort_config<
    rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<1024, 8>, 9,
                                        rocprim::block_radix_rank_algorithm::match>,
    131072>;

extern "C" int64_t bkg_sort_workspace_bytes(int64_t n) {
  size_t bytes = 0, merge_bytes = 0;
  uint32_t* p = nullptr;
  rocprim::radix_sort_pairs<SortConfig>(nullptr, bytes, p, p, p, p, (size_t)n, 0, 32, (hipStream_t)0);
  rocprim::merge_sort(nullptr, merge_bytes, p, p, p, p, (size_t)n, rocprim::less<uint32_t>(), (hipStream_t)0);
  bytes = bytes > merge_bytes ? bytes : merge_bytes;
  size_t scan_bytes = 0;
  int32_t* q = nullptr;
  rocprim::inclusive_scan(nullptr, scan_bytes, q, q, (size_t)n, rocprim::plus<int32_t>(), (hipStream_t)0);
  return (int64_t)(bytes > scan_bytes ? bytes : scan_bytes) + 256;
}

// keys = lo[i] + hi[i] * n_lo (hi nullable); stable sort; outputs sorted keys + permutation.
static int sort_edges_impl(const int64_t* lo, const int64_t* hi, int64_t n_lo, int64_t n_hi, int64_t n,
                           uint32_t* keys_tmp, uint32_t* iota_tmp, uint32_t* sorted_keys, uint32_t* perm,
                           void* workspace, int64_t workspace_bytes, int64_t n_q, uint32_t stride, int32_t* rowptr,
                           void* stream) {
  BKG_ARG(n >= 0 && n_lo > 0 && n_hi > 0, "bkg_sort_edges: bad sizes");
  BKG_ARG((double)n_lo * (double)n_hi < 4294967296.0, "bkg_sort_edges: key space exceeds 32 bits");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    if (rowptr && n_q) fill_u32(rowptr, 0u, n_q, s);
    return rowptr ? check_launch("bkg_sort_edges_rowptr") : 0;
  }
  uint64_t space = (uint64_t)n_lo * (uint64_t)n_hi;
  int bits = 1;
  while (bits < 32 && (1ull << bits) < space) ++bits;
  if (n <= SMALL_SORT_MAX && BKG_SMALL_SORT) {
#define BKG_SMALL_SORT_LAUNCH(IPT)                                                                                   \
  hipLaunchKernelGGL(small_sort_kernel<IPT>, dim3(1), dim3(1024), 0, s, lo, hi, n_lo, (int)n, bits, sorted_keys, perm, \
                     n_q, stride, rowptr)
    if (n <= 1024) BKG_SMALL_SORT_LAUNCH(1);
    else if (n <= 2048) BKG_SMALL_SORT_LAUNCH(2);
    else if (n <= 4096) BKG_SMALL_SORT_LAUNCH(4);
    else BKG_SMALL_SORT_LAUNCH(8);
#undef BKG_SMALL_SORT_LAUNCH
    return check_launch("bkg_sort_edges(one workgroup)");
  }
  hipLaunchKernelGGL(make_keys_kernel, grid1(n), dim3(256), 0, s, lo, hi, n_lo, keys_tmp, iota_tmp, n);
  size_t bytes = (size_t)workspace_bytes;
  // (captured or not: the radix sort -- its memsets are fill kernels in this translation unit, see the top of the file;
  // BKG_CAPTURED_MERGE_SORT=1 brings back the merge sort of rounds 2-3 for a stream that is being captured)
  static const bool merge_when_captured = [] { const char* v = getenv("BKG_CAPTURED_MERGE_SORT"); return v && v[0] == '1'; }();
  hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
  hipError_t e = hipSuccess;
  if (merge_when_captured) {
    e = hipStreamIsCapturing(s, &capturing);
    if (e != hipSuccess) { set_error("bkg_sort_edges: hipStreamIsCapturing", e); return (int)e; }
  }
  if (capturing == hipStreamCaptureStatusActive)
    e = rocprim::merge_sort(workspace, bytes, keys_tmp, sorted_keys, iota_tmp, perm, (size_t)n, rocprim::less<uint32_t>(), s);
  else
    e = rocprim::radix_sort_pairs<SortConfig>(workspace, bytes, keys_tmp, sorted_keys, iota_tmp, perm, (size_t)n, 0, bits, s);
  if (e != hipSuccess) { set_error("bkg_sort_edges: sort", e); return (int)e; }
  if (rowptr && n_q)
    hipLaunchKernelGGL(lower_bound_kernel, grid1(n_q), dim3(256), 0, s, sorted_keys, n, n_q, stride, rowptr);
  return check_launch("bkg_sort_edges");
}

extern "C" int bkg_sort_edges(const int64_t* lo, const int64_t* hi, int64_t n_lo, int64_t n_hi, int64_t n,
                              uint32_t* keys_tmp, uint32_t* iota_tmp, uint32_t* sorted_keys, uint32_t* perm,
                              void* workspace, int64_t workspace_bytes, void* stream) {
  return sort_edges_impl(lo, hi, n_lo, n_hi, n, keys_tmp, iota_tmp, sorted_keys, perm, workspace, workspace_bytes, 0, 1,
                         nullptr, stream);
}

// The same outputs for keys that ARE already in ascending order (the neighbour sampler emits a batch's edges target by
// target in local-id order): sorted_keys = the keys, perm = 0 .. n - 1, the row pointers -- two launches, no sort.
extern "C" int bkg_keys_rowptr_presorted(const int64_t* lo, const int64_t* hi, int64_t n_lo, int64_t n_hi, int64_t n,
                                         uint32_t* sorted_keys, uint32_t* perm, int64_t n_queries, uint32_t stride,
                                         int32_t* rowptr, void* stream) {
  BKG_ARG(n >= 0 && n_lo > 0 && n_hi > 0 && n_queries >= 0 && rowptr, "bkg_keys_rowptr_presorted: bad sizes");
  BKG_ARG((double)n_lo * (double)n_hi < 4294967296.0, "bkg_keys_rowptr_presorted: key space exceeds 32 bits");
  hipStream_t s = (hipStream_t)stream;
  if (n) hipLaunchKernelGGL(make_keys_kernel, grid1(n), dim3(256), 0, s, lo, hi, n_lo, sorted_keys, perm, n);
  if (n_queries) {
    if (n) hipLaunchKernelGGL(lower_bound_kernel, grid1(n_queries), dim3(256), 0, s, sorted_keys, n, n_queries, stride, rowptr);
    else fill_u32(rowptr, 0u, n_queries, s);
  }
  return check_launch("bkg_keys_rowptr_presorted");
}

// bkg_sort_edges + bkg_lower_bound_u32(sorted_keys, n, n_queries, stride, rowptr) in one call (small key sets: one launch)
extern "C" int bkg_sort_edges_rowptr(const int64_t* lo, const int64_t* hi, int64_t n_lo, int64_t n_hi, int64_t n,
                                     uint32_t* keys_tmp, uint32_t* iota_tmp, uint32_t* sorted_keys, uint32_t* perm,
                                     int64_t n_queries, uint32_t stride, int32_t* rowptr, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
  BKG_ARG(n_queries >= 0 && rowptr, "bkg_sort_edges_rowptr: bad arguments");
  return sort_edges_impl(lo, hi, n_lo, n_hi, n, keys_tmp, iota_tmp, sorted_keys, perm, workspace, workspace_bytes, n_queries,
                         stride, rowptr, stream);
}

// (node, role)-sorted incidence lists of a triplet set for bkg_kge_onepass_mx_f32 (reference biomedkg/kge_module.py:121-128
// through its autograd: the node-side sums run over these lists): packed [2 T] = the entries' (other endpoint | relation
// << 24) in list order, rowptr2 [2 n_nodes + 1] = first entry of key 2 * node + role.  keys_tmp / vals_tmp / sorted_keys:
// uint32 [2 T] scratch; workspace of bkg_sort_workspace_bytes(2 T).  Node ids < 2^24, relation ids < 256.  No host read.
extern "C" int bkg_kge_role_incidence(const int64_t* head, const int64_t* tail, const int64_t* rel, int64_t T, int64_t n_nodes,
                                      uint32_t* keys_tmp, uint32_t* vals_tmp, uint32_t* sorted_keys, int32_t* packed,
                                      int32_t* rowptr2, void* workspace, int64_t workspace_bytes, void* stream) {
  BKG_ARG(T >= 0 && n_nodes > 0 && n_nodes <= (1ll << 24) && 2 * T < (1ll << 31) && rowptr2,
          "bkg_kge_role_incidence: node ids below 2^24, 2 T below 2^31");
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = 2 * T, n_q = 2 * n_nodes + 1;
  if (n == 0) {
    fill_u32(rowptr2, 0u, n_q, s);
    return check_launch("bkg_kge_role_incidence");
  }
  BKG_ARG(head && tail && rel && keys_tmp && vals_tmp && sorted_keys && packed && workspace, "bkg_kge_role_incidence: null buffer");
  hipLaunchKernelGGL(role_incidence_keys_kernel, grid1(n), dim3(256), 0, s, head, tail, rel, T, keys_tmp, vals_tmp);
  int bits = 1;
  while (bits < 32 && (1ull << bits) < 2ull * (uint64_t)n_nodes) ++bits;
  size_t bytes = (size_t)workspace_bytes;
  hipError_t e = rocprim::radix_sort_pairs<SortConfig>(workspace, bytes, keys_tmp, sorted_keys, vals_tmp,
                                                       reinterpret_cast<uint32_t*>(packed), (size_t)n, 0, bits, s);
  if (e != hipSuccess) { set_error("bkg_kge_role_incidence: sort", e); return (int)e; }
  hipLaunchKernelGGL(lower_bound_kernel, grid1(n_q), dim3(256), 0, s, sorted_keys, n, n_q, 1u, rowptr2);
  return check_launch("bkg_kge_role_incidence");
}

extern "C" int bkg_lower_bound_u32(const uint32_t* sorted, int64_t n, int64_t n_queries, uint32_t stride,
                                   int32_t* out, void* stream) {
  if (n_queries == 0) return 0;
  hipLaunchKernelGGL(lower_bound_kernel, grid1(n_queries), dim3(256), 0, (hipStream_t)stream, sorted, n, n_queries, stride, out);
  return check_launch("bkg_lower_bound_u32");
}

extern "C" int bkg_gather_i64_to_i32(const int64_t* src, const uint32_t* idx, int32_t* out, int64_t n, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(gather_i64_to_i32_kernel, grid1(n), dim3(256), 0, (hipStream_t)stream, src, idx, out, n);
  return check_launch("bkg_gather_i64_to_i32");
}

extern "C" int bkg_gather_i32(const int32_t* src, const int32_t* idx, int32_t* out, int64_t n, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(gather_i32_kernel, grid1(n), dim3(256), 0, (hipStream_t)stream, src, idx, out, n);
  return check_launch("bkg_gather_i32");
}

extern "C" int bkg_gather_f32(const float* src, const int32_t* idx, float* out, int64_t n, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(gather_f32_kernel, grid1(n), dim3(256), 0, (hipStream_t)stream, src, idx, out, n);
  return check_launch("bkg_gather_f32");
}

extern "C" int64_t bkg_csr_compact_workspace_bytes(int64_t nnz) {
  size_t bytes = 0;
  rocprim::exclusive_scan(nullptr, bytes, (int32_t*)nullptr, (int32_t*)nullptr, 0, (size_t)(nnz > 0 ? nnz : 1),
                          rocprim::plus<int32_t>());
  return (int64_t)bytes + 256;
}

// rowptr_out [n_rows + 1], col_out / w_out [nnz] (the first *count entries are meaningful), count (device) = entries kept.
// flag_tmp / offs_tmp: int32 [nnz] scratch; workspace of bkg_csr_compact_workspace_bytes(nnz).  No host read.
extern "C" int bkg_csr_compact(const int32_t* rowptr, const int32_t* col, const float* w, int64_t n_rows, int64_t nnz,
                               int32_t* flag_tmp, int32_t* offs_tmp, int32_t* rowptr_out, int32_t* col_out, float* w_out,
                               int32_t* count, void* workspace, int64_t workspace_bytes, void* stream) {
  BKG_ARG(n_rows >= 0 && nnz >= 0 && nnz < (1ll << 31) && n_rows < (1ll << 31), "bkg_csr_compact: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  if (nnz) {
    hipLaunchKernelGGL(csr_flag_kernel, grid1(nnz), dim3(256), 0, s, w, nnz, flag_tmp);
    size_t bytes = (size_t)workspace_bytes;
    hipError_t e = rocprim::exclusive_scan(workspace, bytes, flag_tmp, offs_tmp, 0, (size_t)nnz, rocprim::plus<int32_t>(), s);
    if (e != hipSuccess) { set_error("bkg_csr_compact: scan", e); return (int)e; }
  }
  const int64_t span = nnz > n_rows + 1 ? nnz : n_rows + 1;
  hipLaunchKernelGGL(csr_compact_kernel, grid1(span), dim3(256), 0, s, rowptr, col, w, n_rows, nnz, flag_tmp, offs_tmp,
                     rowptr_out, col_out, w_out, count);
  return check_launch("bkg_csr_compact");
}

extern "C" int64_t bkg_gcn_view_workspace_bytes(int64_t nnz) {
  size_t bytes = 0;
  rocprim::exclusive_scan(nullptr, bytes, (int32_t*)nullptr, (int32_t*)nullptr, 0, (size_t)(nnz > 0 ? 2 * nnz : 1),
                          rocprim::plus<int32_t>());
  return (int64_t)bytes + 256;
}

// The two CSRs and gcn_norm weights of a dropout_edge view (see gcn_view_flag_kernel).  In: the graph's CSR by target
// (rowptr_d [n + 1], col_d = sources, perm_d = original edge of each entry, row_d = its target: the sorted keys) and by
// source (..._s), keep [nnz] in edge order.  Out: dis-free results -- self_w [n] (the unit loop's weight 1 / deg), the
// view's CSRs rowptr_out_* [n + 1], col_out_* / w_out_* [nnz] (the first counts[0] / counts[1] entries are meaningful).
// dis [n], flag_tmp / offs_tmp [2 nnz]: scratch; workspace of bkg_gcn_view_workspace_bytes(nnz).
extern "C" int bkg_gcn_view(const int32_t* rowptr_d, const int32_t* col_d, const int32_t* perm_d, const int32_t* row_d,
                            const int32_t* rowptr_s, const int32_t* col_s, const int32_t* perm_s, const int32_t* row_s,
                            const uint8_t* keep, int64_t n_nodes, int64_t nnz, int32_t* flag_tmp, int32_t* offs_tmp, float* dis,
                            float* self_w, int32_t* rowptr_out_d, int32_t* col_out_d, float* w_out_d, int32_t* rowptr_out_s,
                            int32_t* col_out_s, float* w_out_s, int32_t* counts, void* workspace, int64_t workspace_bytes,
                            void* stream) {
  BKG_ARG(n_nodes >= 0 && nnz >= 0 && 2 * nnz < (1ll << 31) && n_nodes < (1ll << 31), "bkg_gcn_view: bad sizes");
  BKG_ARG(nnz == 0 || keep, "bkg_gcn_view: a view needs its keep mask");
  hipStream_t s = (hipStream_t)stream;
  if (nnz) {
    hipLaunchKernelGGL(gcn_view_flag_kernel, grid1(2 * nnz), dim3(256), 0, s, col_d, perm_d, row_d, col_s, perm_s, row_s, keep, nnz,
                       flag_tmp);
    size_t bytes = (size_t)workspace_bytes;
    hipError_t e = rocprim::exclusive_scan(workspace, bytes, flag_tmp, offs_tmp, 0, (size_t)(2 * nnz), rocprim::plus<int32_t>(), s);
    if (e != hipSuccess) { set_error("bkg_gcn_view: scan", e); return (int)e; }
  }
  hipLaunchKernelGGL(gcn_view_rows_kernel, grid1(n_nodes + 1), dim3(256), 0, s, rowptr_d, rowptr_s, flag_tmp, offs_tmp, (int)n_nodes,
                     nnz, rowptr_out_d, rowptr_out_s, dis, self_w, counts);
  if (nnz)
    hipLaunchKernelGGL(gcn_view_scatter_kernel, grid1(2 * nnz), dim3(256), 0, s, col_d, row_d, col_s, row_s, flag_tmp, offs_tmp, dis,
                       nnz, col_out_d, w_out_d, col_out_s, w_out_s);
  return check_launch("bkg_gcn_view");
}

// ----------------------------------------------------------------------------- GCN norm
extern "C" int bkg_gcn_norm(const int32_t* rowptr, const int32_t* col, const int32_t* perm, const int64_t* src,
                            const int64_t* dst, const uint8_t* keep, int64_t n_nodes, int64_t n_edges, float* dis,
                            float* self_w, float* w_edge, void* stream) {
  BKG_ARG(n_nodes >= 0 && n_edges >= 0, "bkg_gcn_norm: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  if (n_nodes) {
    fill_u32(dis, 0u, n_nodes, s);
    if (n_edges) hipLaunchKernelGGL(gcn_degree_count_kernel, grid1(n_edges), dim3(256), 0, s, rowptr, col, perm, keep, (int)n_nodes, n_edges, reinterpret_cast<int32_t*>(dis));
    hipLaunchKernelGGL(gcn_degree_finish_kernel, grid1(n_nodes), dim3(256), 0, s, (int)n_nodes, dis, self_w);
  }
  if (n_edges) hipLaunchKernelGGL(gcn_edge_weight_kernel, grid1(n_edges), dim3(256), 0, s, src, dst, keep, dis, n_edges, w_edge);
  return check_launch("bkg_gcn_norm");
}

// ----------------------------------------------------------------------------- R-GCN segments
// sorted: edge keys (rel * N + dst) ascending.  Outputs: seg_of[E], seg_ptr[<=E+1], seg_dst/seg_rel[<=E],
// inv_cnt[E] (per sorted edge), n_seg (device int32).  flag_tmp/scan_tmp: int32[E] scratch.
extern "C" int bkg_rgcn_segments(const uint32_t* sorted, int64_t n, int64_t n_nodes, int32_t* flag_tmp,
                                 int32_t* scan_tmp, int32_t* seg_of, int32_t* seg_ptr, int32_t* seg_dst,
                                 int32_t* seg_rel, float* inv_cnt, int32_t* n_seg, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) { fill_u32(n_seg, 0u, 1, s); return check_launch("bkg_rgcn_segments(empty)"); }
  hipLaunchKernelGGL(segment_flags_kernel, grid1(n), dim3(256), 0, s, sorted, n, flag_tmp);
  size_t bytes = (size_t)workspace_bytes;
  hipError_t e = rocprim::inclusive_scan(workspace, bytes, flag_tmp, scan_tmp, (size_t)n, rocprim::plus<int32_t>(), s);
  if (e != hipSuccess) { set_error("bkg_rgcn_segments: inclusive_scan", e); return (int)e; }
  hipLaunchKernelGGL(segment_fill_kernel, grid1(n), dim3(256), 0, s, sorted, scan_tmp, n, (uint32_t)n_nodes, seg_of, seg_ptr, seg_dst, seg_rel, n_seg);
  hipLaunchKernelGGL(segment_inv_count_kernel, grid1(n), dim3(256), 0, s, seg_of, seg_ptr, n, inv_cnt);
  return check_launch("bkg_rgcn_segments");
}

extern "C" int bkg_rgcn_edge_order(const uint32_t* perm, const int32_t* seg_of, const float* inv_cnt, int64_t n,
                                   int64_t* seg_of_edge, float* w_edge, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(edge_order_kernel, grid1(n), dim3(256), 0, (hipStream_t)stream, perm, seg_of, inv_cnt, n, seg_of_edge, w_edge);
  return check_launch("bkg_rgcn_edge_order");
}

// bkg_rgcn_segments with a FIXED segment count n (no host read of n_seg): slots past the real count are empty segments
// of relation `last_rel`, targets spread over the nodes.
extern "C" int bkg_rgcn_segments_static(const uint32_t* sorted, int64_t n, int64_t n_nodes, int32_t last_rel,
                                        int32_t* flag_tmp, int32_t* scan_tmp, int32_t* seg_of, int32_t* seg_ptr,
                                        int32_t* seg_dst, int32_t* seg_rel, float* inv_cnt, int32_t* n_seg, void* workspace,
                                        int64_t workspace_bytes, void* stream) {
  BKG_ARG(n > 0 && n_nodes > 0 && last_rel >= 0, "bkg_rgcn_segments_static: bad sizes");
  hipLaunchKernelGGL(segment_prefill_kernel, grid1(n + 1), dim3(256), 0, (hipStream_t)stream, n, (int32_t)n_nodes, last_rel,
                     seg_ptr, seg_dst, seg_rel);
  return bkg_rgcn_segments(sorted, n, n_nodes, flag_tmp, scan_tmp, seg_of, seg_ptr, seg_dst, seg_rel, inv_cnt, n_seg,
                           workspace, workspace_bytes, stream);
}
