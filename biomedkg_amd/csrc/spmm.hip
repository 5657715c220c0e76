// CSR segment-reduce SpMM: the message-passing gather -> aggregate -> "scatter" of GCNConv /
// RGCNConv, done as a destination-sorted segment reduce with plain stores (no float atomics:
// gfx950 global float atomics cap at ~1.3 TB/s, MI355X_MICROARCH.md "Global float atomics").
//
//   out[i, :] = epi( sum_{k in [rowptr[i], rowptr[i+1])} w[k] * x[col[k], :]  (+ self_w[i] * x[i, :]) (+ addend[i, :]) )
//   epi(v)    = dropout(relu(v + bias))   (each stage optional)
//
// One wavefront (64 lanes) owns one work item = one destination row, or one <= chunk-entry slice of
// a long row (rowsplit.hip; partials are added in chunk order by spmm_fixup_kernel).  A row of d
// floats is covered by LPR lanes x VPL float4 each (d = 4*LPR*VPL); when LPR < 64 the 64/LPR lane
// groups take alternating entries of the same item and are combined by shuffles, so the wave never
// diverges on degree.  Item bookkeeping lives in SGPRs (readfirstlane) so the index/weight reads
// are scalar loads and every gathered row is one coalesced 16 B/lane wave-instruction; U entries per
// lane group are kept in flight.
//
// Replaces (functionally) torch_geometric MessagePassing.propagate + scatter for the call sites
// /root/reference/biomedkg/model/encoder.py:52,57,155,160.
#include "common.h"

namespace bkg {

struct SpmmEpilogue {
  const float* self_w;      // [n_rows] or null: adds self_w[i] * x[i,:] (requires x rows == out rows)
  const float* bias;        // [d] or null
  const float* addend;      // [n_rows, ld_add] or null: added before bias / activation
  int64_t ld_add;
  int relu;                 // 0/1
  const uint8_t* keep;      // [n_rows, d] explicit dropout keep mask or null
  float drop_p;             // used when keep == null and drop_p > 0: in-kernel Philox mask
  uint64_t seed;
  uint32_t stream_id;
};

__device__ __forceinline__ void fma4(float4& a, float w, const float4& v) {
  a.x = fmaf(w, v.x, a.x); a.y = fmaf(w, v.y, a.y); a.z = fmaf(w, v.z, a.z); a.w = fmaf(w, v.w, a.w);
}

// acc += sum over entries [beg, end) of this item (all lane groups, then combined across groups)
template <int LPR, int VPL>
__device__ __forceinline__ void spmm_accumulate(float4 (&acc)[VPL], int beg, int end, const int32_t* __restrict__ col,
                                                const float* __restrict__ w, const float* __restrict__ x, int64_t ldx,
                                                int grp, int lig) {
  constexpr int G = 64 / LPR;
  constexpr int U = (VPL == 1) ? 8 : 4;  // entries in flight per lane group
  int e = beg;
  for (; e + U * G <= end; e += U * G) {
    int idx[U];
    float wt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      idx[u] = col[e + u * G + grp];
      wt[u] = w ? w[e + u * G + grp] : 1.0f;
    }
    float4 val[U][VPL];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float4* src = reinterpret_cast<const float4*>(x + (int64_t)idx[u] * ldx) + lig;
#pragma unroll
      for (int v = 0; v < VPL; ++v) val[u][v] = src[v * LPR];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < VPL; ++v) fma4(acc[v], wt[u], val[u][v]);
  }
  for (; e < end; e += G) {
    const int k = e + grp;
    if (k < end) {
      const int idx = col[k];
      const float wt = w ? w[k] : 1.0f;
      const float4* src = reinterpret_cast<const float4*>(x + (int64_t)idx * ldx) + lig;
#pragma unroll
      for (int v = 0; v < VPL; ++v) fma4(acc[v], wt, src[v * LPR]);
    }
  }
  if (G > 1) {
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        acc[v].x += __shfl_xor(acc[v].x, o, 64); acc[v].y += __shfl_xor(acc[v].y, o, 64);
        acc[v].z += __shfl_xor(acc[v].z, o, 64); acc[v].w += __shfl_xor(acc[v].w, o, 64);
      }
  }
}

// self-loop term, addend, bias, ReLU, dropout, store (lane group 0 only)
template <int LPR, int VPL>
__device__ __forceinline__ void spmm_finish(float4 (&acc)[VPL], int row, int lig, const float* __restrict__ x,
                                            int64_t ldx, float* __restrict__ out, int64_t ldo,
                                            const SpmmEpilogue& ep) {
  if (ep.self_w) {
    const float sw = ep.self_w[row];
    const float4* src = reinterpret_cast<const float4*>(x + (int64_t)row * ldx) + lig;
#pragma unroll
    for (int v = 0; v < VPL; ++v) fma4(acc[v], sw, src[v * LPR]);
  }
  constexpr int d = 4 * LPR * VPL;
  const float scale = (ep.keep || ep.drop_p > 0.f) ? 1.0f / (1.0f - ep.drop_p) : 1.0f;
  const uint32_t thresh = (uint32_t)fminf(ep.drop_p * 4294967296.0f, 4294967295.0f);
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int c = 4 * (lig + v * LPR);
    float4 a = acc[v];
    if (ep.addend) {
      const float4 t = *reinterpret_cast<const float4*>(ep.addend + (int64_t)row * ep.ld_add + c);
      a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
    }
    if (ep.bias) {
      const float4 b = *reinterpret_cast<const float4*>(ep.bias + c);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if (ep.relu) {
      a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
    }
    if (ep.keep) {
      const uchar4 k = *reinterpret_cast<const uchar4*>(ep.keep + (int64_t)row * d + c);
      a.x = k.x ? a.x * scale : 0.f; a.y = k.y ? a.y * scale : 0.f;
      a.z = k.z ? a.z * scale : 0.f; a.w = k.w ? a.w * scale : 0.f;
    } else if (ep.drop_p > 0.f) {
      const uint4 r = philox_keep4(ep.seed, (uint64_t)row * d + c, ep.stream_id);
      a.x = r.x >= thresh ? a.x * scale : 0.f; a.y = r.y >= thresh ? a.y * scale : 0.f;
      a.z = r.z >= thresh ? a.z * scale : 0.f; a.w = r.w >= thresh ? a.w * scale : 0.f;
    }
    *(reinterpret_cast<float4*>(out + (int64_t)row * ldo) + lig + v * LPR) = a;
  }
}

// one wave per row (no splitting): small or unskewed inputs
template <int LPR, int VPL>
__global__ __launch_bounds__(256) void spmm_csr_f32_kernel(const int32_t* __restrict__ rowptr,
                                                           const int32_t* __restrict__ col,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ x, int64_t ldx,
                                                           float* __restrict__ out, int64_t ldo, int n_rows,
                                                           SpmmEpilogue ep) {
  const int lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  row = __builtin_amdgcn_readfirstlane(row);
  const int grp = lane / LPR, lig = lane % LPR;
  const int beg = __builtin_amdgcn_readfirstlane(rowptr[row]);
  const int end = __builtin_amdgcn_readfirstlane(rowptr[row + 1]);
  float4 acc[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  spmm_accumulate<LPR, VPL>(acc, beg, end, col, w, x, ldx, grp, lig);
  if (grp == 0) spmm_finish<LPR, VPL>(acc, row, lig, x, ldx, out, ldo, ep);
}

// one wave per work item of the row-split table
template <int LPR, int VPL>
__global__ __launch_bounds__(256) void spmm_items_f32_kernel(bkg_rowsplit rs, const int32_t* __restrict__ col,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ x, int64_t ldx,
                                                             float* __restrict__ out, int64_t ldo,
                                                             float* __restrict__ partials, SpmmEpilogue ep) {
  const int lane = threadIdx.x & 63;
  int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= rs.totals[0]) return;
  item = __builtin_amdgcn_readfirstlane(item);
  const int grp = lane / LPR, lig = lane % LPR;
  const int row = __builtin_amdgcn_readfirstlane(rs.item_row[item]);
  const int beg = __builtin_amdgcn_readfirstlane(rs.item_beg[item]);
  const int end = __builtin_amdgcn_readfirstlane(rs.item_end[item]);
  const int slot = __builtin_amdgcn_readfirstlane(rs.item_slot[item]);
  float4 acc[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  spmm_accumulate<LPR, VPL>(acc, beg, end, col, w, x, ldx, grp, lig);
  if (grp != 0) return;
  if (slot < 0) {
    spmm_finish<LPR, VPL>(acc, row, lig, x, ldx, out, ldo, ep);
  } else {
    constexpr int d = 4 * LPR * VPL;
#pragma unroll
    for (int v = 0; v < VPL; ++v) *(reinterpret_cast<float4*>(partials + (int64_t)slot * d) + lig + v * LPR) = acc[v];
  }
}

// long rows: add the row's partial slots in chunk order, then the epilogue
template <int LPR, int VPL>
__global__ __launch_bounds__(256) void spmm_fixup_f32_kernel(bkg_rowsplit rs, const float* __restrict__ x, int64_t ldx,
                                                             float* __restrict__ out, int64_t ldo,
                                                             const float* __restrict__ partials, SpmmEpilogue ep) {
  constexpr int G = 64 / LPR, d = 4 * LPR * VPL;
  const int lane = threadIdx.x & 63;
  const int l = (blockIdx.x * 4 + (threadIdx.x >> 6)) * G + lane / LPR;  // one lane group per long row
  if (l >= rs.totals[2]) return;
  const int lig = lane % LPR;
  const int row = rs.long_row[l], s0 = rs.long_slot0[l], n = rs.long_cnt[l];
  float4 acc[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int c = 0; c < n; ++c) {
    const float4* src = reinterpret_cast<const float4*>(partials + (int64_t)(s0 + c) * d) + lig;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const float4 t = src[v * LPR];
      acc[v].x += t.x; acc[v].y += t.y; acc[v].z += t.z; acc[v].w += t.w;
    }
  }
  spmm_finish<LPR, VPL>(acc, row, lig, x, ldx, out, ldo, ep);
}

// Any d (not a multiple of 4 / not a supported width): lanes stride over the columns.
__global__ __launch_bounds__(256) void spmm_csr_f32_generic_kernel(const int32_t* __restrict__ rowptr,
                                                                   const int32_t* __restrict__ col,
                                                                   const float* __restrict__ w,
                                                                   const float* __restrict__ x, int64_t ldx,
                                                                   float* __restrict__ out, int64_t ldo, int n_rows,
                                                                   int d, SpmmEpilogue ep) {
  const int lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int start = rowptr[row], end = rowptr[row + 1];
  const float scale = (ep.keep || ep.drop_p > 0.f) ? 1.0f / (1.0f - ep.drop_p) : 1.0f;
  const uint32_t thresh = (uint32_t)fminf(ep.drop_p * 4294967296.0f, 4294967295.0f);
  for (int c = lane; c < d; c += 64) {
    float a = 0.f;
    for (int e = start; e < end; ++e) a = fmaf(w ? w[e] : 1.0f, x[(int64_t)col[e] * ldx + c], a);
    if (ep.self_w) a = fmaf(ep.self_w[row], x[(int64_t)row * ldx + c], a);
    if (ep.addend) a += ep.addend[(int64_t)row * ep.ld_add + c];
    if (ep.bias) a += ep.bias[c];
    if (ep.relu) a = fmaxf(a, 0.f);
    if (ep.keep) {
      a = ep.keep[(int64_t)row * d + c] ? a * scale : 0.f;
    } else if (ep.drop_p > 0.f) {
      const uint64_t eid = (uint64_t)row * d + c;
      const uint4 r = philox_keep4(ep.seed, eid, ep.stream_id);
      const uint32_t rv = (eid & 3) == 0 ? r.x : (eid & 3) == 1 ? r.y : (eid & 3) == 2 ? r.z : r.w;
      a = rv >= thresh ? a * scale : 0.f;
    }
    out[(int64_t)row * ldo + c] = a;
  }
}

struct SpmmArgs {
  const int32_t* rowptr; const int32_t* col; const float* w; const float* x; int64_t ldx; float* out; int64_t ldo;
  int n_rows; const bkg_rowsplit* split; float* partials; SpmmEpilogue ep; hipStream_t s;
};

template <int LPR, int VPL>
static void launch_spmm(const SpmmArgs& a) {
  if (a.split) {
    constexpr int G = 64 / LPR;
    hipLaunchKernelGGL((spmm_items_f32_kernel<LPR, VPL>), dim3(ceil_div(a.split->n_items_max, 4)), dim3(256), 0, a.s,
                       *a.split, a.col, a.w, a.x, a.ldx, a.out, a.ldo, a.partials, a.ep);
    hipLaunchKernelGGL((spmm_fixup_f32_kernel<LPR, VPL>), dim3(ceil_div(a.split->n_long_max, 4 * G)), dim3(256), 0,
                       a.s, *a.split, a.x, a.ldx, a.out, a.ldo, a.partials, a.ep);
  } else {
    hipLaunchKernelGGL((spmm_csr_f32_kernel<LPR, VPL>), dim3(ceil_div(a.n_rows, 4)), dim3(256), 0, a.s, a.rowptr,
                       a.col, a.w, a.x, a.ldx, a.out, a.ldo, a.n_rows, a.ep);
  }
}

}  // namespace bkg

extern "C" int bkg_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* w, const float* x,
                                int64_t ldx, float* out, int64_t ldo, int64_t n_rows, int32_t d,
                                const float* self_w, const float* addend, int64_t ld_add, const float* bias,
                                int32_t relu, const uint8_t* keep, float drop_p, uint64_t seed,
                                uint32_t stream_id, const bkg_rowsplit* split, float* partials, void* stream) {
  using namespace bkg;
  BKG_ARG(n_rows >= 0 && d > 0, "bkg_spmm_csr_f32: bad shape");
  BKG_ARG(drop_p >= 0.f && drop_p < 1.f, "bkg_spmm_csr_f32: drop_p must be in [0,1)");
  BKG_ARG(!split || partials, "bkg_spmm_csr_f32: row splitting needs a partials buffer");
  if (n_rows == 0) return 0;
  SpmmEpilogue ep{self_w, bias, addend, ld_add, relu, keep, drop_p, seed, stream_id};
  const bool vec_ok = (d % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) && (((uintptr_t)x | (uintptr_t)out) % 16 == 0) &&
                      (!bias || ((uintptr_t)bias % 16 == 0)) && (!keep || ((uintptr_t)keep % 4 == 0)) &&
                      (!addend || (((uintptr_t)addend % 16 == 0) && (ld_add % 4 == 0))) &&
                      (!partials || ((uintptr_t)partials % 16 == 0));
  SpmmArgs a{rowptr, col, w, x, ldx, out, ldo, (int)n_rows, split, partials, ep, (hipStream_t)stream};
  if (vec_ok && d == 16) launch_spmm<4, 1>(a);
  else if (vec_ok && d == 32) launch_spmm<8, 1>(a);
  else if (vec_ok && d == 64) launch_spmm<16, 1>(a);
  else if (vec_ok && d == 128) launch_spmm<32, 1>(a);
  else if (vec_ok && d == 256) launch_spmm<64, 1>(a);
  else if (vec_ok && d == 512) launch_spmm<64, 2>(a);
  else if (vec_ok && d == 768) launch_spmm<64, 3>(a);
  else if (vec_ok && d == 1024) launch_spmm<64, 4>(a);
  else
    hipLaunchKernelGGL(spmm_csr_f32_generic_kernel, dim3(ceil_div(n_rows, 4)), dim3(256), 0, a.s, rowptr, col, w, x, ldx,
                       out, ldo, (int)n_rows, d, ep);
  return check_launch("bkg_spmm_csr_f32");
}
