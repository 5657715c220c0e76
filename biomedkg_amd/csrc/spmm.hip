// CSR segment-reduce SpMM: the message-passing gather -> aggregate -> "scatter" of GCNConv /
// RGCNConv, done as a destination-sorted segment reduce with plain stores (no float atomics:
// gfx950 global float atomics cap at ~1.3 TB/s, MI355X_MICROARCH.md "Global float atomics").
//
//   out[i, :] = epi( sum_{k in [rowptr[i], rowptr[i+1])} w[k] * x[col[k], :]  (+ self_w[i] * x[i, :]) (+ addend[i, :]) )
//   epi(v)    = dropout(relu(v + bias))   (each stage optional)
//
// One wavefront (64 lanes) owns one work item = one destination row, or one <= chunk-entry slice of
// a long row (rowsplit.hip; partials are added in chunk order by spmm_fixup_kernel).  A row of d
// values is covered by LPR lanes x VPL 16-byte loads each; when LPR < 64 the 64/LPR lane groups take
// alternating entries of the same item and are combined by shuffles, so the wave never diverges on
// degree.  Item bookkeeping lives in SGPRs (readfirstlane) so the index/weight reads are scalar loads
// and every gathered row is one coalesced 16 B/lane wave-instruction; U entries per lane group are
// kept in flight.
//
// The gathered table x is fp32 (exact mode) or bf16 (BASELINE config 3: "bf16 storage / fp32
// accumulate"): a 16-byte load then carries 8 columns, the row is half the bytes and the launch -- which
// is bound by the bytes leaving L2 (profiles/r01_pmc_traffic.json) -- moves half the traffic.
// Accumulation, epilogue and the output are fp32 in both cases.
//
// Replaces (functionally) torch_geometric MessagePassing.propagate + scatter for the call sites
// /root/reference/biomedkg/model/encoder.py:52,57,155,160.
#include "common.h"
#include "rowpiece.h"

#ifndef SPMM_L32
#define SPMM_L32 32   // lanes per row of a 32-piece row (512 bytes) in the item kernel (swept: docs/experiments.md 9-5)
#endif
#ifndef SPMM_U1
#define SPMM_U1 8    // entries in flight per lane group, one piece per lane (swept: see docs/experiments.md 9-5)
#endif

namespace bkg {

struct SpmmEpilogue {
  const float* self_w;      // [n_rows] or null: adds self_w[i] * x[i,:] (requires x rows == out rows)
  const float* bias;        // [d] or null
  const float* addend;      // [n_rows, ld_add] or null: added before bias / activation
  int64_t ld_add;
  int add_bf16;             // 1: `addend` is a bf16 table (a product that only this epilogue reads)
  int relu;                 // 0/1
  const uint8_t* keep;      // [n_rows, d] explicit dropout keep mask or null
  float drop_p;             // used when keep == null and drop_p > 0: in-kernel Philox mask
  uint64_t seed;
  uint32_t stream_id;
  int out_bf16;             // 1: `out` is a bf16 [n_rows, ldo] table (a result only GEMMs / gathers read)
  const uint32_t* epoch;    // replay epoch of a captured step (common.h with_epoch) or null
  // column halves (bkg_spmm_csr_f32 col_halves = 2): the launch runs TWO problems of half the width side by side, the
  // blocks of XCDs 0-3 (blockIdx % 8 < 4: workgroups go round-robin over the eight XCDs) on columns [0, d), those of XCDs
  // 4-7 on [d, 2 d) -- an L2 then holds 512-byte pieces of HALF the table's rows' bytes instead of whole 1 KB rows.
  // col0: first column of this block's half (set in the kernel); d_full: the row width the dropout mask is indexed by
  // (0: the kernel's own d); part_stride: floats between the two halves' partial-slot buffers
  int halves, col0, d_full;
  int64_t part_stride;
  // a bf16 result through ANOTHER layer's ReLU / dropout backward gate (the result is the gradient of that layer's
  // output `out_gate`): out = bf16(out_gate > 0 ? bf16(v) * out_gate_scale : 0) -- the values of a gate pass over the
  // finished bf16 table (bkg_gate_cast_bf16), without the pass
  const __bf16* out_gate;   // [n_rows, ld_gate] or null
  int64_t ld_gate;
  float out_gate_scale;
};

// block id inside its half; `half` (0 / 1) tells the kernel which columns and which partial-slot buffer are its own
// (halves == 1: the block id itself, half 0)
template <int D>
__device__ __forceinline__ unsigned spmm_half_of(SpmmEpilogue& ep, unsigned& half) {
  unsigned bid = blockIdx.x;
  half = 0u;
  if (ep.halves > 1) {                                  // 2 or 4 column parts: 4 or 2 XCDs each
    const unsigned xcd = bid & 7u, per = 8u / (unsigned)ep.halves;
    half = xcd / per;
    bid = (bid >> 3) * per + xcd % per;
    ep.col0 = (int)half * D;
  }
  return bid;
}

// (entry batches -- EntryIdx / EntryWt, scalar loads for one or two lane groups -- live in rowpiece.h: the KGE backward's
// stream kernel walks its incidence lists the same way)
template <typename T, int LPR, int VPL, int UT>
__device__ __forceinline__ void gather_rows(Piece<T> (&val)[UT][VPL], const EntryIdx<64 / LPR, UT>& idx,
                                            const T* __restrict__ x, int64_t ldx, int grp, int lig) {
  constexpr int EPL = Piece<T>::EPL;
#pragma unroll
  for (int u = 0; u < UT; ++u) {
    const T* src = x + (int64_t)idx.mine(u, grp) * ldx + EPL * lig;
#pragma unroll
    for (int v = 0; v < VPL; ++v) val[u][v].load(src + EPL * v * LPR);
  }
}

// One predicated batch of UT entries per lane group starting at e (entries >= end read a clamped, valid index so that
// the batch's loads are unconditional, and contribute nothing).
template <typename T, int LPR, int VPL, int UT, bool HAS_W>
__device__ __forceinline__ void spmm_tail(float4* acc, int e, int end, const int32_t* __restrict__ col,
                                          const float* __restrict__ w, const T* __restrict__ x, int64_t ldx, int grp,
                                          int lig) {
  constexpr int G = 64 / LPR, Q = Piece<T>::Q;
  EntryIdx<G, UT> idx;
  idx.load(e, end - 1, col, grp);
  EntryWt<G, UT, HAS_W> wt;
  wt.load(e, end - 1, w, grp);
  Piece<T> val[UT][VPL];
  gather_rows<T, LPR, VPL, UT>(val, idx, x, ldx, grp, lig);
#pragma unroll
  for (int u = 0; u < UT; ++u)
    if (e + u * G + grp < end) {  // a padded slot contributes nothing, whatever the clamped row holds
      const float wu = wt.mine(u, grp);
#pragma unroll
      for (int v = 0; v < VPL; ++v) val[u][v].fma(acc + v * Q, wu);
    }
}

// acc += sum over entries [beg, end) of this item (all lane groups, then combined across groups).
// acc has VPL * Q float4: piece v of lane `lig` covers columns EPL * (lig + v * LPR) + [0, EPL).
template <typename T, int LPR, int VPL, bool HAS_W>
__device__ __forceinline__ void spmm_accumulate_w(float4* acc, int beg, int end, const int32_t* __restrict__ col,
                                                  const float* __restrict__ w, const T* __restrict__ x, int64_t ldx,
                                                  int grp, int lig) {
  constexpr int G = 64 / LPR, Q = Piece<T>::Q;
  constexpr int U = (VPL == 1) ? SPMM_U1 : 4;  // entries in flight per lane group
  constexpr int NB = U * G;
  int e = beg;
  if (e + NB <= end) {
    EntryIdx<G, U> idx;
    idx.load(e, NO_CLAMP, col, grp);
    for (;;) {
      Piece<T> val[U][VPL];
      gather_rows<T, LPR, VPL, U>(val, idx, x, ldx, grp, lig);
      EntryWt<G, U, HAS_W> wt;
      wt.load(e, NO_CLAMP, w, grp);              // arrives long before the rows
      e += NB;
      const bool more = e + NB <= end;           // wave-uniform
      if (more) idx.load(e, NO_CLAMP, col, grp); // the next batch's indices, under this batch's row loads
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float wu = wt.mine(u, grp);
#pragma unroll
        for (int v = 0; v < VPL; ++v) val[u][v].fma(acc + v * Q, wu);
      }
      if (!more) break;
    }
  }
  // Remainder (the common case: most rows are shorter than one full batch).  One PREDICATED batch sized to
  // the wave-uniform remainder keeps all its index loads, then all its row loads, in flight together; a
  // one-entry-per-iteration tail would pay the index -> row latency chain once per entry.
  const int rem = end - e;  // wave-uniform, < U * G
  if (rem > 0) {
    if (rem > 4 * G) spmm_tail<T, LPR, VPL, (U > 4 ? U : 4), HAS_W>(acc, e, end, col, w, x, ldx, grp, lig);
    else if (rem > 2 * G) spmm_tail<T, LPR, VPL, 4, HAS_W>(acc, e, end, col, w, x, ldx, grp, lig);
    else if (rem > G) spmm_tail<T, LPR, VPL, 2, HAS_W>(acc, e, end, col, w, x, ldx, grp, lig);
    else spmm_tail<T, LPR, VPL, 1, HAS_W>(acc, e, end, col, w, x, ldx, grp, lig);
  }
  if (G > 1) {
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
      for (int v = 0; v < VPL * Q; ++v) {
        acc[v].x += __shfl_xor(acc[v].x, o, 64); acc[v].y += __shfl_xor(acc[v].y, o, 64);
        acc[v].z += __shfl_xor(acc[v].z, o, 64); acc[v].w += __shfl_xor(acc[v].w, o, 64);
      }
  }
}

template <typename T, int LPR, int VPL>
__device__ __forceinline__ void spmm_accumulate(float4* acc, int beg, int end, const int32_t* __restrict__ col,
                                                const float* __restrict__ w, const T* __restrict__ x, int64_t ldx,
                                                int grp, int lig) {
  if (w) spmm_accumulate_w<T, LPR, VPL, true>(acc, beg, end, col, w, x, ldx, grp, lig);
  else spmm_accumulate_w<T, LPR, VPL, false>(acc, beg, end, col, w, x, ldx, grp, lig);
}

// self-loop term, addend, bias, ReLU, dropout, store (lane group 0 only)
template <typename T, int LPR, int VPL>
__device__ __forceinline__ void spmm_finish(float4* acc, int row, int lig, const T* __restrict__ x, int64_t ldx,
                                            float* __restrict__ out, int64_t ldo, const SpmmEpilogue& ep) {
  constexpr int Q = Piece<T>::Q, EPL = Piece<T>::EPL, d = EPL * LPR * VPL;
  if (ep.self_w) {
    const float sw = ep.self_w[row];
    const T* src = x + (int64_t)row * ldx + EPL * lig;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      Piece<T> p;
      p.load(src + EPL * v * LPR);
      p.fma(acc + v * Q, sw);
    }
  }
  const float scale = (ep.keep || ep.drop_p > 0.f) ? 1.0f / (1.0f - ep.drop_p) : 1.0f;
  const uint32_t thresh = (uint32_t)fminf(ep.drop_p * 4294967296.0f, 4294967295.0f);
  const int64_t dm = ep.d_full ? ep.d_full : d;                   // row width of the dropout mask
  uint2 lo = make_uint2(0u, 0u);
#pragma unroll
  for (int v = 0; v < VPL; ++v)
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int c = EPL * (lig + v * LPR) + 4 * q + ep.col0;      // the column in out / addend / bias / the mask
      float4 a = acc[v * Q + q];
      if (ep.addend) {
        if (ep.add_bf16) {
          const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const __bf16*>(ep.addend) + (int64_t)row * ep.ld_add + c);
          a.x += __uint_as_float(t.x << 16); a.y += __uint_as_float(t.x & 0xffff0000u);
          a.z += __uint_as_float(t.y << 16); a.w += __uint_as_float(t.y & 0xffff0000u);
        } else {
          const float4 t = *reinterpret_cast<const float4*>(ep.addend + (int64_t)row * ep.ld_add + c);
          a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
      }
      if (ep.bias) {
        const float4 b = *reinterpret_cast<const float4*>(ep.bias + c);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      if (ep.relu) {
        a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
      }
      if (ep.keep) {
        const uchar4 k = *reinterpret_cast<const uchar4*>(ep.keep + (int64_t)row * dm + c);
        a.x = k.x ? a.x * scale : 0.f; a.y = k.y ? a.y * scale : 0.f;
        a.z = k.z ? a.z * scale : 0.f; a.w = k.w ? a.w * scale : 0.f;
      } else if (ep.drop_p > 0.f) {
        const uint4 r = philox_keep4(ep.seed, (uint64_t)row * dm + c, with_epoch(ep.stream_id, ep.epoch));
        a.x = r.x >= thresh ? a.x * scale : 0.f; a.y = r.y >= thresh ? a.y * scale : 0.f;
        a.z = r.z >= thresh ? a.z * scale : 0.f; a.w = r.w >= thresh ? a.w * scale : 0.f;
      }
      if (ep.out_bf16) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 h;
        h[0] = (__bf16)a.x; h[1] = (__bf16)a.y; h[2] = (__bf16)a.z; h[3] = (__bf16)a.w;
        if (ep.out_gate) {
          const uint2 t = *reinterpret_cast<const uint2*>(ep.out_gate + (int64_t)row * ep.ld_gate + c);
          const float g0 = __uint_as_float(t.x << 16), g1 = __uint_as_float(t.x & 0xffff0000u);
          const float g2 = __uint_as_float(t.y << 16), g3 = __uint_as_float(t.y & 0xffff0000u);
          h[0] = (__bf16)(g0 > 0.f ? (float)h[0] * ep.out_gate_scale : 0.f);
          h[1] = (__bf16)(g1 > 0.f ? (float)h[1] * ep.out_gate_scale : 0.f);
          h[2] = (__bf16)(g2 > 0.f ? (float)h[2] * ep.out_gate_scale : 0.f);
          h[3] = (__bf16)(g3 > 0.f ? (float)h[3] * ep.out_gate_scale : 0.f);
        }
        if constexpr (Q == 2) {     // the two quads of a 16-byte piece leave as ONE 16-byte store (8 columns)
          if (q == 0) { lo = __builtin_bit_cast(uint2, h); continue; }
          const uint2 hi = __builtin_bit_cast(uint2, h);
          *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(out) + (int64_t)row * ldo + c - 4) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        } else {
          *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(out) + (int64_t)row * ldo + c) = h;
        }
      } else {
        *reinterpret_cast<float4*>(out + (int64_t)row * ldo + c) = a;
      }
    }
}

// ---- short rows (rowsplit.hip: rows of 0..8 entries of a packed table).  RPW = 64 / LS rows per wave: lane group g
// of LS lanes owns row desc[i0 + g] alone -- VS 16-byte pieces per lane cover the row -- so there is no cross-lane
// reduction and no per-row wave.  Descriptor and the row's (padded) indices arrive in one round trip, the weights
// travel beside the rows: two dependent memory levels per wave instead of four, every load unconditional.
template <typename T, int LS, int VS>
__device__ __forceinline__ void spmm_short_rows(int i, int n_short, const int32_t* __restrict__ desc,
                                                const int32_t* __restrict__ scol, const float* __restrict__ w,
                                                const T* __restrict__ x, int64_t ldx, float* __restrict__ out, int64_t ldo,
                                                const SpmmEpilogue& ep, int lig) {
  constexpr int Q = Piece<T>::Q, EPL = Piece<T>::EPL, VQ = VS * Q;
  constexpr int UB = 4;                                       // entries per pass (88 % of the bench graph's short rows
  const bool live = i < n_short;                              // have <= 4: most waves stop after the first)
  if (!__any(live)) return;                                   // (bounds not cut to the count: whole waves past the list)
  const int ic = live ? i : 0;                                // a live lane exists, so entry 0 of the table is valid
  const int4 dsc = reinterpret_cast<const int4*>(desc)[ic];
  const int4 c0 = reinterpret_cast<const int4*>(scol)[2 * ic], c1 = reinterpret_cast<const int4*>(scol)[2 * ic + 1];
  const int idx[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
  const int row = dsc.x, beg = dsc.y, len = live ? dsc.z : 0;
  float4 acc[VQ];
  zero4(acc);
#pragma unroll
  for (int b0 = 0; b0 < 8; b0 += UB) {
    if (b0 > 0 && !__any(b0 < len)) break;
    Piece<T> val[UB][VS];
#pragma unroll
    for (int u = 0; u < UB; ++u) {     // unconditional (pads repeat the row's last index: an L1 hit); exec-masking the
      const T* src = x + (int64_t)idx[b0 + u] * ldx + EPL * lig;   // pads was measured slower (54 -> 62 us)
#pragma unroll
      for (int v = 0; v < VS; ++v) val[u][v].load(src + EPL * v * LS);
    }
    float wt[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) wt[u] = (w && b0 + u < len) ? w[beg + b0 + u] : 1.0f;
#pragma unroll
    for (int u = 0; u < UB; ++u)
      if (b0 + u < len) {
#pragma unroll
        for (int v = 0; v < VS; ++v) val[u][v].fma(acc + v * Q, wt[u]);
      }
  }
  if (live) spmm_finish<T, LS, VS>(acc, row, lig, x, ldx, out, ldo, ep);
}

// one wave per row (no splitting): small or unskewed inputs
template <typename T, int LPR, int VPL>
__global__ __launch_bounds__(256) void spmm_csr_kernel(const int32_t* __restrict__ rowptr,
                                                       const int32_t* __restrict__ col, const float* __restrict__ w,
                                                       const T* __restrict__ x, int64_t ldx, float* __restrict__ out,
                                                       int64_t ldo, int n_rows, SpmmEpilogue ep) {
  const int lane = threadIdx.x & 63;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  row = __builtin_amdgcn_readfirstlane(row);
  const int grp = lane / LPR, lig = lane % LPR;
  const int beg = __builtin_amdgcn_readfirstlane(rowptr[row]);
  const int end = __builtin_amdgcn_readfirstlane(rowptr[row + 1]);
  float4 acc[VPL * Piece<T>::Q];
  zero4(acc);
  spmm_accumulate<T, LPR, VPL>(acc, beg, end, col, w, x, ldx, grp, lig);
  if (grp == 0) spmm_finish<T, LPR, VPL>(acc, row, lig, x, ldx, out, ldo, ep);
}

// First launch of a split table: one wave per work item (a whole row, or one <= chunk-entry slice of a long row)
template <typename T, int LPR, int VPL>
__global__ __launch_bounds__(256) void spmm_items_kernel(bkg_rowsplit rs, const int32_t* __restrict__ col,
                                                         const float* __restrict__ w, const T* __restrict__ x,
                                                         int64_t ldx, float* __restrict__ out, int64_t ldo,
                                                         float* __restrict__ partials, SpmmEpilogue ep) {
  constexpr int Q = Piece<T>::Q, EPL = Piece<T>::EPL;
  unsigned half;
  const int bi = (int)spmm_half_of<EPL * LPR * VPL>(ep, half);
  x += half * (EPL * LPR * VPL);
  partials += half * ep.part_stride;
  const int lane = threadIdx.x & 63;
  int item = bi * 4 + (threadIdx.x >> 6);
  if (item >= rs.totals[0]) return;
  item = __builtin_amdgcn_readfirstlane(item);
  const int grp = lane / LPR, lig = lane % LPR;
  const int row = __builtin_amdgcn_readfirstlane(rs.item_row[item]);
  const int beg = __builtin_amdgcn_readfirstlane(rs.item_beg[item]);
  const int end = __builtin_amdgcn_readfirstlane(rs.item_end[item]);
  const int slot = __builtin_amdgcn_readfirstlane(rs.item_slot[item]);
  float4 acc[VPL * Q];
  zero4(acc);
  spmm_accumulate<T, LPR, VPL>(acc, beg, end, col, w, x, ldx, grp, lig);
  if (grp != 0) return;
  if (slot < 0) {
    spmm_finish<T, LPR, VPL>(acc, row, lig, x, ldx, out, ldo, ep);
  } else {
    constexpr int d = EPL * LPR * VPL;
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
      for (int q = 0; q < Q; ++q)
        *reinterpret_cast<float4*>(partials + (int64_t)slot * d + EPL * (lig + v * LPR) + 4 * q) = acc[v * Q + q];
  }
}

// Second launch of a split table.  Blocks [0, n_long_max): long rows, one block per row.  Its 4 * G lane groups each
// add a strided subset of the row's partial slots (8 loads in flight), the group sums are combined through LDS in group
// order (a fixed tree: bitwise repeatable), then the epilogue.  (One lane group per row walked a hub row's ~150 slots
// serially: 20 us per launch for a handful of rows.)  Blocks from n_long_max on: the packed table's short rows,
// 256 / LS per block (spmm_short_rows).  (The two roles in ONE launch with the item waves, interleaved block by block,
// was measured and lost: the union of their register needs costs the item waves two of their seven slots per SIMD --
// 254 against 174 us on the bench graph's segments.)
template <typename T, int LPR, int VPL, int LS, int VS>
__global__ __launch_bounds__(256) void spmm_fixup_kernel(bkg_rowsplit rs, const float* __restrict__ w,
                                                         const T* __restrict__ x, int64_t ldx,
                                                         float* __restrict__ out, int64_t ldo,
                                                         float* __restrict__ partials, SpmmEpilogue ep) {
  constexpr int G = 64 / LPR, NG = 4 * G, Q = Piece<T>::Q, EPL = Piece<T>::EPL, VQ = VPL * Q, d = EPL * LPR * VPL;
  __shared__ __attribute__((aligned(16))) float red[NG * d];
  unsigned half;
  const unsigned bid = spmm_half_of<d>(ep, half);
  x += half * d;
  partials += half * ep.part_stride;
  if (bid >= (unsigned)rs.n_long_max) {      // block-uniform
    const int i = ((int)bid - (int)rs.n_long_max) * (256 / LS) + (int)threadIdx.x / LS;
    spmm_short_rows<T, LS, VS>(i, rs.totals[3], rs.short_desc, rs.short_col, w, x, ldx, out, ldo, ep, (int)threadIdx.x % LS);
    return;
  }
  const int l = bid;
  if (l >= rs.totals[2]) return;    // block-uniform
  const int lane = threadIdx.x & 63, lig = lane % LPR, gi = (threadIdx.x >> 6) * G + lane / LPR;
  const int row = rs.long_row[l], s0 = rs.long_slot0[l], n = rs.long_cnt[l];
  float4 acc[VQ];
  zero4(acc);
  constexpr int UF = 8;
  for (int c0 = gi; c0 < n; c0 += UF * NG) {
    float4 t[UF][VQ];
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const int c = min(c0 + u * NG, n - 1);
#pragma unroll
      for (int vq = 0; vq < VQ; ++vq)
        t[u][vq] = *reinterpret_cast<const float4*>(partials + (int64_t)(s0 + c) * d + EPL * (lig + (vq / Q) * LPR) + 4 * (vq % Q));
    }
#pragma unroll
    for (int u = 0; u < UF; ++u)
      if (c0 + u * NG < n) {
#pragma unroll
        for (int vq = 0; vq < VQ; ++vq) {
          acc[vq].x += t[u][vq].x; acc[vq].y += t[u][vq].y; acc[vq].z += t[u][vq].z; acc[vq].w += t[u][vq].w;
        }
      }
  }
#pragma unroll
  for (int vq = 0; vq < VQ; ++vq)
    *reinterpret_cast<float4*>(red + gi * d + EPL * (lig + (vq / Q) * LPR) + 4 * (vq % Q)) = acc[vq];
  __syncthreads();
  if (gi != 0) return;
  for (int g2 = 1; g2 < NG; ++g2)
#pragma unroll
    for (int vq = 0; vq < VQ; ++vq) {
      const float4 v = *reinterpret_cast<const float4*>(red + g2 * d + EPL * (lig + (vq / Q) * LPR) + 4 * (vq % Q));
      acc[vq].x += v.x; acc[vq].y += v.y; acc[vq].z += v.z; acc[vq].w += v.w;
    }
  spmm_finish<T, LPR, VPL>(acc, row, lig, x, ldx, out, ldo, ep);
}

// Any d (not a multiple of 4 / not a supported width), fp32 table only: lanes stride over the columns.
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
    if (ep.addend)
      a += ep.add_bf16 ? (float)reinterpret_cast<const __bf16*>(ep.addend)[(int64_t)row * ep.ld_add + c]
                       : ep.addend[(int64_t)row * ep.ld_add + c];
    if (ep.bias) a += ep.bias[c];
    if (ep.relu) a = fmaxf(a, 0.f);
    if (ep.keep) {
      a = ep.keep[(int64_t)row * d + c] ? a * scale : 0.f;
    } else if (ep.drop_p > 0.f) {
      const uint64_t eid = (uint64_t)row * d + c;
      const uint4 r = philox_keep4(ep.seed, eid, with_epoch(ep.stream_id, ep.epoch));
      const uint32_t rv = (eid & 3) == 0 ? r.x : (eid & 3) == 1 ? r.y : (eid & 3) == 2 ? r.z : r.w;
      a = rv >= thresh ? a * scale : 0.f;
    }
    out[(int64_t)row * ldo + c] = a;
  }
}

template <typename T>
struct SpmmArgs {
  const int32_t* rowptr; const int32_t* col; const float* w; const T* x; int64_t ldx; float* out; int64_t ldo;
  int n_rows; const bkg_rowsplit* split; float* partials; SpmmEpilogue ep; hipStream_t s;
};

// lanes per SHORT row for a row of LPR * VPL pieces: two pieces per lane where the width allows (16 lanes x 2 pieces for
// the d = 256 bf16 table: 4 rows per wave, all 8 entries of each in flight), never fewer than 4 lanes
template <int LPR, int VPL> struct ShortShape {
  static constexpr int P = LPR * VPL, C = P / 2;                        // candidate: two pieces per lane
  static constexpr bool OK = P % 2 == 0 && C >= 4 && C <= 64 && 64 % C == 0;
  static constexpr int LS = OK ? C : LPR, VS = OK ? 2 : VPL;
};

template <typename T, int LPR, int VPL>
static void launch_spmm(const SpmmArgs<T>& a) {
  if (a.split) {
    using SS = ShortShape<LPR, VPL>;
    static_assert(SS::LS * SS::VS == LPR * VPL && 64 % SS::LS == 0, "short-row shape");
    // column halves: two problems per launch, four blocks of each per group of eight (spmm_half_of)
    auto grid_of = [&](int64_t nb) { return (unsigned)(a.ep.halves > 1 ? ceil_div(nb, 8 / a.ep.halves) * 8 : nb); };
    const int nb_items = ceil_div(a.split->n_items_max, 4);
    const int64_t nb_short = a.split->short_desc ? ceil_div(a.split->n_short_max, 256 / SS::LS) : 0;
    if (nb_items > 0)
      hipLaunchKernelGGL((spmm_items_kernel<T, LPR, VPL>), dim3(grid_of(nb_items)), dim3(256), 0, a.s, *a.split, a.col, a.w,
                         a.x, a.ldx, a.out, a.ldo, a.partials, a.ep);
    if (a.split->n_long_max + nb_short > 0)
      hipLaunchKernelGGL((spmm_fixup_kernel<T, LPR, VPL, SS::LS, SS::VS>), dim3(grid_of(a.split->n_long_max + nb_short)),
                         dim3(256), 0, a.s, *a.split, a.w, a.x, a.ldx, a.out, a.ldo, a.partials, a.ep);
  } else {
    hipLaunchKernelGGL((spmm_csr_kernel<T, LPR, VPL>), dim3(ceil_div(a.n_rows, 4)), dim3(256), 0, a.s, a.rowptr, a.col,
                       a.w, a.x, a.ldx, a.out, a.ldo, a.n_rows, a.ep);
  }
}

// d -> (LPR, VPL) for 16-byte pieces of EPL columns
template <typename T>
static bool dispatch_spmm(int d, const SpmmArgs<T>& a) {
  constexpr int EPL = Piece<T>::EPL;
  switch (d / EPL) {  // pieces per row
    case 4: launch_spmm<T, 4, 1>(a); return true;
    case 8: launch_spmm<T, 8, 1>(a); return true;
    case 16: launch_spmm<T, 16, 1>(a); return true;
    case 32: launch_spmm<T, SPMM_L32, 32 / SPMM_L32>(a); return true;
    case 64: launch_spmm<T, 64, 1>(a); return true;
    case 96: launch_spmm<T, 32, 3>(a); return true;
    case 128: launch_spmm<T, 64, 2>(a); return true;
    case 192: launch_spmm<T, 64, 3>(a); return true;
    case 256: launch_spmm<T, 64, 4>(a); return true;
    default: return false;
  }
}

}  // namespace bkg

extern "C" int bkg_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* w, const void* x,
                                int32_t x_is_bf16, int64_t ldx, void* out_, int32_t out_is_bf16, int64_t ldo,
                                int64_t n_rows, int32_t d, const float* self_w, const void* addend, int32_t addend_is_bf16,
                                int64_t ld_add, const float* bias,
                                int32_t relu, const uint8_t* keep, float drop_p, uint64_t seed,
                                uint32_t stream_id, const void* out_gate, int64_t ld_gate, float out_gate_scale,
                                const bkg_rowsplit* split, float* partials, int32_t col_halves, void* stream) {
  using namespace bkg;
  BKG_ARG(!out_gate || (out_is_bf16 && (uintptr_t)out_gate % 8 == 0 && ld_gate % 4 == 0),
          "bkg_spmm_csr_f32: an output gate goes with a bf16 result; 8-byte aligned bf16 rows");
  BKG_ARG(n_rows >= 0 && d > 0, "bkg_spmm_csr_f32: bad shape");
  BKG_ARG(col_halves == 1 || ((col_halves == 2 || col_halves == 4) && split && d % (8 * col_halves) == 0),
          "bkg_spmm_csr_f32: col_halves = 2 | 4 needs a row table and d % (8 col_halves) == 0");
  BKG_ARG(drop_p >= 0.f && drop_p < 1.f, "bkg_spmm_csr_f32: drop_p must be in [0,1)");
  BKG_ARG(!split || partials, "bkg_spmm_csr_f32: row splitting needs a partials buffer");
  if (n_rows == 0) return 0;
  float* out = reinterpret_cast<float*>(out_);
  SpmmEpilogue ep{self_w, bias, reinterpret_cast<const float*>(addend), ld_add, addend_is_bf16 ? 1 : 0, relu, keep, drop_p, seed,
                  stream_id, out_is_bf16 ? 1 : 0, rng_epoch_ptr(), 1, 0, 0, 0,
                  reinterpret_cast<const __bf16*>(out_gate), ld_gate, out_gate_scale};
  if (col_halves > 1) {            // the kernels run at a part's width; the mask keeps the whole row's indexing
    ep.halves = col_halves;
    ep.d_full = d;
    d /= col_halves;
    ep.part_stride = (int64_t)split->n_slots_max * d;
  }
  const int epl = x_is_bf16 ? 8 : 4;
  const bool vec_ok = (d % epl == 0) && (ldx % epl == 0) && (ldo % 4 == 0) && (((uintptr_t)x | (uintptr_t)out) % 16 == 0) &&
                      (!bias || ((uintptr_t)bias % 16 == 0)) && (!keep || ((uintptr_t)keep % 4 == 0)) &&
                      (!addend || (((uintptr_t)addend % 16 == 0) && (ld_add % 4 == 0))) &&
                      (!partials || ((uintptr_t)partials % 16 == 0));
  hipStream_t s = (hipStream_t)stream;
  bool done = false;
  if (x_is_bf16) {
    BKG_ARG(vec_ok, "bkg_spmm_csr_f32: a bf16 table needs d % 8 == 0 and 16-byte aligned rows");
    SpmmArgs<__bf16> a{rowptr, col, w, (const __bf16*)x, ldx, out, ldo, (int)n_rows, split, partials, ep, s};
    done = dispatch_spmm<__bf16>(d, a);
    BKG_ARG(done, "bkg_spmm_csr_f32: unsupported width for a bf16 table (d/8 in {4,8,16,32,64,96,128,192,256})");
  } else {
    SpmmArgs<float> a{rowptr, col, w, (const float*)x, ldx, out, ldo, (int)n_rows, split, partials, ep, s};
    if (vec_ok) done = dispatch_spmm<float>(d, a);
    BKG_ARG(done || !out_is_bf16, "bkg_spmm_csr_f32: a bf16 result needs a vector-width d and aligned rows");   // (=> no gate either)
    if (!done)
      hipLaunchKernelGGL(spmm_csr_f32_generic_kernel, dim3(ceil_div(n_rows, 4)), dim3(256), 0, s, rowptr, col, w,
                         (const float*)x, ldx, out, ldo, (int)n_rows, d, ep);
  }
  return check_launch("bkg_spmm_csr_f32");
}

// ---- bf16 (RNE) gather table from an fp32 or bf16 source, optionally through the ReLU/dropout backward gate
//      (gate = the layer's saved fp32 or bf16 OUTPUT):  dst = bf16( gate ? (gate > 0 ? src * scale : 0) : src )
namespace bkg {
__device__ __forceinline__ void load8(const float* p, int64_t i, float (&v)[8]) {
  const float4 a = reinterpret_cast<const float4*>(p)[2 * i], b = reinterpret_cast<const float4*>(p)[2 * i + 1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const __bf16* p, int64_t i, float (&v)[8]) {
  const uint4 u = reinterpret_cast<const uint4*>(p)[i];
  v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
  v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
  v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
  v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}

template <typename ST, typename GT>
__global__ __launch_bounds__(256) void cast_bf16_kernel(const ST* __restrict__ src, const GT* __restrict__ gate,
                                                        float scale, __bf16* __restrict__ dst, int64_t n8) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  float a[8];
  load8(src, i, a);
  if (gate) {
    float g[8];
    load8(gate, i, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = g[e] > 0.f ? a[e] * scale : 0.f;
  }
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (__bf16)a[e];
  reinterpret_cast<bf16x8*>(dst)[i] = o;
}
}  // namespace bkg

extern "C" int bkg_gate_cast_bf16(const void* src, int32_t src_is_bf16, const void* gate, int32_t gate_is_bf16,
                                  float gate_scale, void* dst, int64_t n, void* stream) {
  using namespace bkg;
  BKG_ARG(n >= 0 && n % 8 == 0, "bkg_gate_cast_bf16: element count must be a multiple of 8");
  BKG_ARG(((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0) && (!gate || (uintptr_t)gate % 16 == 0),
          "bkg_gate_cast_bf16: 16-byte alignment");
  if (n == 0) return 0;
  dim3 grid(ceil_div(n / 8, 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define BKG_CAST(ST, GT) hipLaunchKernelGGL((cast_bf16_kernel<ST, GT>), grid, block, 0, s, (const ST*)src, (const GT*)gate, gate_scale, (__bf16*)dst, n / 8)
  if (src_is_bf16) { if (gate_is_bf16) BKG_CAST(__bf16, __bf16); else BKG_CAST(__bf16, float); }
  else { if (gate_is_bf16) BKG_CAST(float, __bf16); else BKG_CAST(float, float); }
#undef BKG_CAST
  return check_launch("bkg_gate_cast_bf16");
}

// ---- feature masking of the GCL augmentations (reference biomedkg/model/gcl.py:42,76: x * (rand >= p), no
//      rescale) fused with the bf16 table cast: dst = bf16(philox(seed, stream, i) >= p ? src[i] : 0); p = 0: plain cast
namespace bkg {
template <typename ST>
__global__ __launch_bounds__(256) void mask_cast_bf16_kernel(const ST* __restrict__ src, float p, uint64_t seed,
                                                             uint32_t stream_id, __bf16* __restrict__ dst, int64_t n8,
                                                             const uint32_t* __restrict__ epoch) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  stream_id = with_epoch(stream_id, epoch);
  float a[8];
  load8(src, i, a);
  if (p > 0.f) {
    const uint32_t thresh = (uint32_t)fminf(p * 4294967296.0f, 4294967295.0f);
    const uint4 r0 = philox_keep4(seed, (uint64_t)i * 8, stream_id), r1 = philox_keep4(seed, (uint64_t)i * 8 + 4, stream_id);
    a[0] = r0.x >= thresh ? a[0] : 0.f; a[1] = r0.y >= thresh ? a[1] : 0.f;
    a[2] = r0.z >= thresh ? a[2] : 0.f; a[3] = r0.w >= thresh ? a[3] : 0.f;
    a[4] = r1.x >= thresh ? a[4] : 0.f; a[5] = r1.y >= thresh ? a[5] : 0.f;
    a[6] = r1.z >= thresh ? a[6] : 0.f; a[7] = r1.w >= thresh ? a[7] : 0.f;
  }
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (__bf16)a[e];
  reinterpret_cast<bf16x8*>(dst)[i] = o;
}
}  // namespace bkg

extern "C" int bkg_mask_cast_bf16(const void* src, int32_t src_is_bf16, float p, uint64_t seed, uint32_t stream_id,
                                  void* dst, int64_t n, void* stream) {
  using namespace bkg;
  BKG_ARG(n >= 0 && n % 8 == 0 && p >= 0.f && p < 1.f, "bkg_mask_cast_bf16: n must be a multiple of 8, p in [0,1)");
  BKG_ARG(((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0), "bkg_mask_cast_bf16: 16-byte alignment");
  if (n == 0) return 0;
  if (src_is_bf16)      // (the cached bf16 copy of constant features: bf16(mask ? x : 0) = mask ? bf16(x) : 0, half the bytes read)
    hipLaunchKernelGGL(mask_cast_bf16_kernel<__bf16>, dim3(ceil_div(n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const __bf16*)src, p, seed, stream_id, (__bf16*)dst, n / 8, rng_epoch_ptr());
  else
    hipLaunchKernelGGL(mask_cast_bf16_kernel<float>, dim3(ceil_div(n / 8, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)src, p, seed, stream_id, (__bf16*)dst, n / 8, rng_epoch_ptr());
  return check_launch("bkg_mask_cast_bf16");
}

extern "C" int bkg_cast_f32_bf16(const float* src, const float* gate, float gate_scale, void* dst, int64_t n,
                                 void* stream) {
  return bkg_gate_cast_bf16(src, 0, gate, 0, gate_scale, dst, n, stream);
}
