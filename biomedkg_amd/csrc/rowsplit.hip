// Long-row splitting for the CSR segment-reduce kernels (spmm.hip, kge.hip).
//
// PrimeKG-like graphs are heavily skewed (a handful of rows with 10^4-10^5 entries against a median
// of tens): with one wave per row the launch runs as long as its longest row.  Following
// cdna_hip_programming.md Appendix B "Scatter / gather / embedding", rows longer than `chunk` entries
// are cut into chunks summed by separate waves into a partials buffer, and a second pass adds each
// row's partials in chunk order (fixed order => bitwise reproducible).  The item table is built on the
// device with three exclusive scans; array sizes are upper bounds so no host sync is needed.
#include <cstring>

#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace bkg {

// ---- short rows (pack != 0).  PrimeKG-like row-length histograms are bimodal: a few rows hold most of the entries and
// most rows hold a handful (the bench graph's (relation, target) segments: 81 % have <= 8 entries, together 13 % of the
// entries).  One wave per such row is bound by wave launch and the item -> index -> row latency chain (272 k one-entry
// rows take 91 us however little they move; that class alone ran 107 of the launch's 220 us).  Rows of 0..SHORT_LEN
// entries therefore leave the item list for a compacted list of descriptors {row, first entry, length, 0} that
// bkg_spmm_csr_f32 walks with SEVERAL ROWS PER WAVE (one row per lane group of 4..32 lanes, no cross-lane reduction), and
// their column indices are copied into a padded [n_short][SHORT_LEN] array (pads repeat the row's last entry): the
// kernel reads descriptor and indices in ONE round trip instead of descriptor -> indices, and all its loads are
// unconditional.
constexpr int SHORT_LEN = 8;

// cnt[0][r] = items of row r (0 for a short row of a packed table), cnt[1][r] = partial slots (0 unless long),
// cnt[2][r] = is long, cnt[3][r] = is short
__global__ void rowsplit_count_kernel(const int32_t* __restrict__ rowptr, int n_rows, int chunk, int pack,
                                      int32_t* __restrict__ cnt) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n_rows) return;
  int items = 0, slots = 0, lng = 0, shrt = 0;
  if (r < n_rows) {
    int len = rowptr[r + 1] - rowptr[r];
    items = len <= chunk ? 1 : (len + chunk - 1) / chunk;
    if (items > 1) { slots = items; lng = 1; }
    if (pack && len <= SHORT_LEN) { items = 0; shrt = 1; }
  }
  const int stride = n_rows + 1;
  cnt[r] = items;
  cnt[stride + r] = slots;
  cnt[2 * stride + r] = lng;
  cnt[3 * stride + r] = shrt;
}

__global__ void rowsplit_fill_kernel(const int32_t* __restrict__ rowptr, int n_rows, int chunk, int pack,
                                     const int32_t* __restrict__ scan, int32_t* __restrict__ item_row,
                                     int32_t* __restrict__ item_beg, int32_t* __restrict__ item_end,
                                     int32_t* __restrict__ item_slot, int32_t* __restrict__ long_row,
                                     int32_t* __restrict__ long_slot0, int32_t* __restrict__ long_cnt,
                                     int32_t* __restrict__ short_desc, const int32_t* __restrict__ col,
                                     int32_t* __restrict__ short_col, int32_t* __restrict__ totals) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int stride = n_rows + 1;
  // ONE exclusive scan runs over the four counter arrays laid end to end: array k's prefix sums are its part of the scan
  // minus what the arrays before it add up to (= the scan value at its first element)
  const int b1 = scan[stride], b2 = scan[2 * stride], b3 = scan[3 * stride];
  if (r == n_rows) {
    totals[0] = scan[n_rows];
    totals[1] = scan[stride + n_rows] - b1;
    totals[2] = scan[2 * stride + n_rows] - b2;
    totals[3] = scan[3 * stride + n_rows] - b3;
  }
  if (r >= n_rows) return;
  const int beg = rowptr[r], end = rowptr[r + 1], len = end - beg;
  if (pack && len <= SHORT_LEN) {
    const int i = scan[3 * stride + r] - b3;
    reinterpret_cast<int4*>(short_desc)[i] = make_int4(r, beg, len, 0);
    int c[SHORT_LEN];
#pragma unroll
    for (int j = 0; j < SHORT_LEN; ++j) c[j] = len > 0 ? col[beg + min(j, len - 1)] : 0;
    reinterpret_cast<int4*>(short_col)[2 * i] = make_int4(c[0], c[1], c[2], c[3]);
    reinterpret_cast<int4*>(short_col)[2 * i + 1] = make_int4(c[4], c[5], c[6], c[7]);
    return;
  }
  const int i0 = scan[r];
  if (len <= chunk) {
    item_row[i0] = r; item_beg[i0] = beg; item_end[i0] = end; item_slot[i0] = -1;
    return;
  }
  const int n = (len + chunk - 1) / chunk, s0 = scan[stride + r] - b1, l = scan[2 * stride + r] - b2;
  long_row[l] = r; long_slot0[l] = s0; long_cnt[l] = n;
  for (int c = 0; c < n; ++c) {
    item_row[i0 + c] = r;
    item_beg[i0 + c] = beg + c * chunk;
    item_end[i0 + c] = min(beg + (c + 1) * chunk, end);
    item_slot[i0 + c] = s0 + c;
  }
}

}  // namespace bkg

using namespace bkg;

extern "C" int bkg_rowsplit_build(const int32_t* rowptr, int64_t n_rows, int64_t nnz, int32_t chunk, int32_t pack,
                                  int32_t* cnt_tmp, int32_t* scan_tmp, int32_t* item_row, int32_t* item_beg,
                                  int32_t* item_end, int32_t* item_slot, int32_t* long_row, int32_t* long_slot0,
                                  int32_t* long_cnt, int32_t* short_desc, const int32_t* col, int32_t* short_col,
                                  int32_t* totals, void* workspace, int64_t workspace_bytes, void* stream) {
  BKG_ARG(n_rows >= 0 && nnz >= 0 && chunk >= 16, "bkg_rowsplit_build: bad sizes (chunk >= 16)");
  BKG_ARG(!pack || (short_desc && short_col && col && ((uintptr_t)short_desc | (uintptr_t)short_col) % 16 == 0),
          "bkg_rowsplit_build: pack needs col and 16-byte aligned short_desc / short_col");
  hipStream_t s = (hipStream_t)stream;
  const int n = (int)n_rows;
  dim3 grid(ceil_div(n_rows + 1, 256)), block(256);
  hipLaunchKernelGGL(rowsplit_count_kernel, grid, block, 0, s, rowptr, n, chunk, pack ? 1 : 0, cnt_tmp);
  const size_t stride = (size_t)n_rows + 1;
  {   // the four counter arrays in one scan (two launches instead of eight: a mini-batch step builds two of these tables)
    size_t need = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, need, cnt_tmp, scan_tmp, (int32_t)0, 4 * stride, rocprim::plus<int32_t>(), s);
    if (e != hipSuccess) { set_error("bkg_rowsplit_build: exclusive_scan (size query)", e); return (int)e; }
    BKG_ARG(need <= (size_t)workspace_bytes, "bkg_rowsplit_build: workspace too small for the scan");
    size_t bytes = (size_t)workspace_bytes;
    e = rocprim::exclusive_scan(workspace, bytes, cnt_tmp, scan_tmp, (int32_t)0, 4 * stride, rocprim::plus<int32_t>(), s);
    if (e != hipSuccess) { set_error("bkg_rowsplit_build: exclusive_scan", e); return (int)e; }
  }
  hipLaunchKernelGGL(rowsplit_fill_kernel, grid, block, 0, s, rowptr, n, chunk, pack ? 1 : 0, scan_tmp, item_row, item_beg,
                     item_end, item_slot, long_row, long_slot0, long_cnt, short_desc, col, short_col, totals);
  return check_launch("bkg_rowsplit_build");
}
