// KGE scorer backward, node side: dz[i] = sum over the triplets incident to node i of the contribution of
// the OTHER endpoint.  NOT a scatter-add: the caller sorts the 2T (node, role) incidences by node once
// (graph.hip radix sort; entry e < T: i is the head of triplet e, else the tail of e - T) and this kernel
// does a per-node segment reduce with plain stores -- bitwise reproducible, and several times the
// float-atomic rate on gfx950.  Long incidence lists are split into <= 256-entry work items (rowsplit.hip).
//
//   DistMult (reference decoder.py:45-49):  dz_i += g_t * rel[r_t] (*) z[other]
//   TransE   (reference decoder.py:31-38):  with z^ = z / max(||z||_1, eps), u = h^ + r - t^,
//            Q_i = sum_t (-/+) g_t sign(u_t)  (head / tail role), then through the L1 normalisation ONCE per
//            node:  dz_i = (Q_i - sign(z_i) (z^_i . Q_i)) / ||z_i||_1   (Q_i / eps when ||z_i||_1 <= eps).
//
// The gathered node table z is fp32 or bf16 (rowpiece.h); g, rel_emb, l1 and dz are fp32.
#include "common.h"
#include "rowpiece.h"

namespace bkg {

enum { KN_DISTMULT = 0, KN_TRANSE = 1 };
constexpr float KN_EPS = 1e-12f;
#ifndef KS_RUNS
#define KS_RUNS 0             // 1: the relation row applied once per run of equal ids (pays only on relation-ordered lists;
                              // measured equal on the north-star graph, so the lists stay in entry order)
#endif
#ifndef KS_L32
#define KS_L32 16     // lanes per row of a 32-piece table (d = 256 bf16) in the stream form (swept: 8 0.775, 16 0.662, 32 0.676 ms per tagged call)
#endif
#ifndef KS_U2
#define KS_U2 4       // entries in flight per lane group at two pieces per lane (swept: docs/experiments.md 9-5)
#endif
#ifndef KS_ATTR
#define KS_ATTR
#endif
#ifndef KN_REL_LDS
#define KN_REL_LDS 1          // the relation table staged in LDS, persistent workgroups (0: rows from the caches)
#endif
#ifndef KN_PERSIST_BLOCKS
#define KN_PERSIST_BLOCKS 8192  // workgroups of the persistent form (swept on the bench graph: 1024 0.699, 2048 0.693, 4096 0.688, 8192 0.677, 16384 0.682, 32768 0.701 ms per tagged call)
#endif
#ifndef KN_U4
#define KN_U4 2      // incidences in flight per lane group at four accumulator quads (experiment knob)
#endif

__device__ __forceinline__ float kn_sgn(float v) { return (v > 0.f) - (v < 0.f); }
__device__ __forceinline__ float4 kn_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int LPR>
__device__ __forceinline__ float kn_group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// What the LAST launch over a dz buffer may do on its way out (the whole-graph training step: the loss's upstream gradient
// and the L2 term of reference kge_module.py:128-129, otherwise one more pass over [N, d] -- bkg_scale_add_rows_f32):
// dz = dz * gloss[0] + zreg * (gloss[0] * coef).  gloss == nullptr: off.
struct KnTail {
  const float* gloss;
  const float* zreg;      // nullable: scale only
  int64_t ldzr;
  float coef;
  // an accumulating launch adds to THESE rows instead of to dz's own (nullable): the positives' part of dz computed by
  // the forward's one-pass walk stays intact for a second backward
  const float* acc_src;
  int64_t ld_src;
};
__device__ __forceinline__ void kn_tail(float4& v, const KnTail& tl, int node, int col) {
  if (!tl.gloss) return;
  const float g = tl.gloss[0];
  v.x *= g; v.y *= g; v.z *= g; v.w *= g;
  if (tl.zreg) {
    const float c = g * tl.coef;
    const float4 s = *reinterpret_cast<const float4*>(tl.zreg + (int64_t)node * tl.ldzr + col);
    v.x = fmaf(s.x, c, v.x); v.y = fmaf(s.y, c, v.y); v.z = fmaf(s.z, c, v.z); v.w = fmaf(s.w, c, v.w);
  }
}

struct KnTriplets {
  const int64_t* head;
  const int64_t* tail;
  const int64_t* rel;
  int64_t T;
  const int4* rec;   // nullable: {head, tail, rel, bits(g)} per triplet (written by the fused forward pass); when given
                     // it replaces head / tail / rel / gscore -- one 16-byte gather per incidence instead of four
};

// z^_i (TransE) of lane `lig`'s columns
template <typename T, int LPR, int VPL>
__device__ __forceinline__ void kn_load_self(const T* __restrict__ z, int64_t ldz, int node, int lig, float inv_self,
                                             float4* self) {
  constexpr int Q = Piece<T>::Q, EPL = Piece<T>::EPL;
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    Piece<T> p;
    p.load(z + (int64_t)node * ldz + EPL * (lig + v * LPR));
    p.expand(self + v * Q);
  }
#pragma unroll
  for (int q = 0; q < VPL * Q; ++q) {
    self[q].x *= inv_self; self[q].y *= inv_self; self[q].z *= inv_self; self[q].w *= inv_self;
  }
}

// TransE normalisation backward (needs every lane of the group), then store / accumulate
template <typename T, int KIND, int LPR, int VPL>
__device__ __forceinline__ void kn_finish(float4* acc, const float4* self, float inv_self, bool above_eps, int node,
                                          int lig, float* __restrict__ dz, int64_t lddz, int accumulate, bool store,
                                          const KnTail& tl = KnTail{nullptr, nullptr, 0, 0.f, nullptr, 0}) {
  constexpr int VQ = VPL * Piece<T>::Q;
  if (KIND == KN_TRANSE) {
    float dot = 0.f;
#pragma unroll
    for (int q = 0; q < VQ; ++q)
      dot += self[q].x * acc[q].x + self[q].y * acc[q].y + self[q].z * acc[q].z + self[q].w * acc[q].w;
    dot = kn_group_sum<LPR>(dot);
    if (!above_eps) dot = 0.f;
#pragma unroll
    for (int q = 0; q < VQ; ++q) {
      acc[q].x = (acc[q].x - kn_sgn(self[q].x) * dot) * inv_self; acc[q].y = (acc[q].y - kn_sgn(self[q].y) * dot) * inv_self;
      acc[q].z = (acc[q].z - kn_sgn(self[q].z) * dot) * inv_self; acc[q].w = (acc[q].w - kn_sgn(self[q].w) * dot) * inv_self;
    }
  }
  if (!store) return;
#pragma unroll
  for (int q = 0; q < VQ; ++q) {
    float4* dst = reinterpret_cast<float4*>(dz + (int64_t)node * lddz + piece_col<T, LPR>(lig, q));
    if (accumulate) {
      const float4 o = tl.acc_src ? *reinterpret_cast<const float4*>(tl.acc_src + (int64_t)node * tl.ld_src + piece_col<T, LPR>(lig, q))
                                  : *dst;
      acc[q].x += o.x; acc[q].y += o.y; acc[q].z += o.z; acc[q].w += o.w;
    }
    kn_tail(acc[q], tl, node, piece_col<T, LPR>(lig, q));
    *dst = acc[q];
  }
}

// SPLIT: one wave per row-split item instead of one wave per node
// REL_LDS: the relation table (n_rel rows + a zero row for the pad relation id of padded steps, <= 64 KB) is staged in LDS
// once per workgroup, and a workgroup walks a strided share of the items instead of four.  Measured neutral at the bench
// size (0.737 ms per launch either way; with the relation factor taken out altogether -- KN_EXP_NOREL -- 0.60): the row
// comes out of LDS as slowly as out of the L1, what costs is the chain entry id -> record -> (row, relation row).
template <typename T, int KIND, int LPR, int VPL, bool SPLIT, bool REL_LDS>
__global__ __launch_bounds__(256) void kge_bwd_nodes_kernel(const T* __restrict__ z, const T* __restrict__ z_self,
                                                            int64_t ldz, int n_nodes,
                                                            KnTriplets tr, const float* __restrict__ rel_emb,
                                                            int64_t ldr, int n_rel, const float* __restrict__ gscore,
                                                            const float* __restrict__ l1, const float* __restrict__ l1_self,
                                                            const int32_t* __restrict__ inc_rowptr,
                                                            const int32_t* __restrict__ inc_perm,
                                                            float* __restrict__ dz, int64_t lddz, bkg_rowsplit rs,
                                                            float* __restrict__ partials, int accumulate) {
  constexpr int G = 64 / LPR, Q = Piece<T>::Q, EPL = Piece<T>::EPL, VQ = VPL * Q, D = EPL * LPR * VPL;
  extern __shared__ __attribute__((aligned(16))) float rel_lds[];
  const int lane = threadIdx.x & 63, grp = lane / LPR, lig = lane % LPR;
  if (REL_LDS) {
    for (int i = threadIdx.x; i < (n_rel + 1) * (D / 4); i += 256) {      // (+ a zero row: the pad relation id of padded steps)
      const int r = i / (D / 4), c = i % (D / 4);
      reinterpret_cast<float4*>(rel_lds)[i] = r < n_rel ? kn_ld4(rel_emb + (int64_t)r * ldr + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
  }
  const int n_work = SPLIT ? rs.totals[0] : n_nodes;
  for (int work = blockIdx.x * 4 + (threadIdx.x >> 6); work < n_work; work += gridDim.x * 4) {
  int node, start, end, slot = -1;
  if (SPLIT) {
    const int item = __builtin_amdgcn_readfirstlane(work);
    node = __builtin_amdgcn_readfirstlane(rs.item_row[item]);
    start = __builtin_amdgcn_readfirstlane(rs.item_beg[item]);
    end = __builtin_amdgcn_readfirstlane(rs.item_end[item]);
    slot = __builtin_amdgcn_readfirstlane(rs.item_slot[item]);
  } else {
    node = __builtin_amdgcn_readfirstlane(work);
    start = __builtin_amdgcn_readfirstlane(inc_rowptr[node]);
    end = __builtin_amdgcn_readfirstlane(inc_rowptr[node + 1]);
  }
  float4 self[VQ], acc[VQ];
  float inv_self = 0.f;
  zero4(acc);
  if (KIND == KN_TRANSE) {
    inv_self = 1.0f / fmaxf(l1_self[node], KN_EPS);
    kn_load_self<T, LPR, VPL>(z_self, ldz, node, lig, inv_self, self);
  }
  // Batches of U incidences per lane group through a three-stage software pipeline: with the rows of batch i the
  // triplet fields of batch i+1 and the entry ids of batch i+2 are in flight (issued before the rows, so the one
  // wait per iteration retires all three): the chain entry id -> triplet fields -> rows costs one memory latency
  // per batch.  Slots past the end are clamped and masked out.
  constexpr int U = (VQ == 1) ? 4 : (VQ >= 8 ? 1 : (VQ == 4 ? KN_U4 : 2));
  constexpr int STEP = U * G;
  int64_t other[U], ri[U];
  float g[U];
  bool is_head[U];
  int e_n[U];
  auto fields = [&](int e, int64_t& o, int64_t& r, float& gg, bool& hd) {
    hd = e < tr.T;
    const int64_t t = hd ? e : e - tr.T;
    if (tr.rec) {
      const int4 v = tr.rec[t];
      o = hd ? v.y : v.x;
      r = v.z;
      gg = __int_as_float(v.w);
    } else {
      o = hd ? tr.tail[t] : tr.head[t];
      r = tr.rel[t];
      gg = gscore[t];
    }
  };
  if (start < end) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e0 = inc_perm[min(start + u * G + grp, end - 1)];
      e_n[u] = inc_perm[min(start + STEP + u * G + grp, end - 1)];
      fields(e0, other[u], ri[u], g[u], is_head[u]);
    }
  }
  for (int k0 = start; k0 < end; k0 += STEP) {
    int64_t other_n[U], ri_n[U];
    float g_n[U], inv_o[U];
    bool is_head_n[U], ok[U];
    int e_nn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ok[u] = k0 + u * G + grp < end;
      fields(e_n[u], other_n[u], ri_n[u], g_n[u], is_head_n[u]);
      e_nn[u] = inc_perm[min(k0 + 2 * STEP + u * G + grp, end - 1)];
    }
    Piece<T> ob[U][VPL];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      inv_o[u] = (KIND == KN_TRANSE) ? 1.0f / fmaxf(l1[other[u]], KN_EPS) : 0.f;
#pragma unroll
      for (int v = 0; v < VPL; ++v) ob[u][v].load(z + other[u] * ldz + EPL * (lig + v * LPR));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;
      float4 o[VQ];
#pragma unroll
      for (int v = 0; v < VPL; ++v) ob[u][v].expand(o + v * Q);
      const float sg = is_head[u] ? -g[u] : g[u], io = inv_o[u];
#pragma unroll
      for (int q = 0; q < VQ; ++q) {
#if defined(KN_EXP_NOREL)
        const float4 r = make_float4(1.f, 1.f, 1.f, 1.f);
#else
        const float4 r = REL_LDS ? *reinterpret_cast<const float4*>(rel_lds + (int)ri[u] * D + piece_col<T, LPR>(lig, q))
                                 : kn_ld4(rel_emb + (n_rel > 0 ? min(ri[u], (int64_t)n_rel - 1) : ri[u]) * ldr + piece_col<T, LPR>(lig, q));   // (the pad relation id of a padded step: a finite row times g = 0)
#endif
        if (KIND == KN_DISTMULT) {
          acc[q].x = fmaf(g[u] * r.x, o[q].x, acc[q].x); acc[q].y = fmaf(g[u] * r.y, o[q].y, acc[q].y);
          acc[q].z = fmaf(g[u] * r.z, o[q].z, acc[q].z); acc[q].w = fmaf(g[u] * r.w, o[q].w, acc[q].w);
        } else {
          float4 d4;  // u = h^ + r - t^ ;  d score / d h^ = -sign(u), d score / d t^ = +sign(u)
          if (is_head[u]) d4 = make_float4(self[q].x + r.x - o[q].x * io, self[q].y + r.y - o[q].y * io,
                                           self[q].z + r.z - o[q].z * io, self[q].w + r.w - o[q].w * io);
          else d4 = make_float4(o[q].x * io + r.x - self[q].x, o[q].y * io + r.y - self[q].y,
                                o[q].z * io + r.z - self[q].z, o[q].w * io + r.w - self[q].w);
          acc[q].x = fmaf(sg, kn_sgn(d4.x), acc[q].x); acc[q].y = fmaf(sg, kn_sgn(d4.y), acc[q].y);
          acc[q].z = fmaf(sg, kn_sgn(d4.z), acc[q].z); acc[q].w = fmaf(sg, kn_sgn(d4.w), acc[q].w);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      other[u] = other_n[u]; ri[u] = ri_n[u]; g[u] = g_n[u]; is_head[u] = is_head_n[u]; e_n[u] = e_nn[u];
    }
  }
  if (G > 1) {
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
      for (int q = 0; q < VQ; ++q) {
        acc[q].x += __shfl_xor(acc[q].x, o, 64); acc[q].y += __shfl_xor(acc[q].y, o, 64);
        acc[q].z += __shfl_xor(acc[q].z, o, 64); acc[q].w += __shfl_xor(acc[q].w, o, 64);
      }
  }
  // after the cross-group combine every group holds the full sum: group 0 writes it
  if (SPLIT && slot >= 0) {
    if (grp == 0) {
#pragma unroll
      for (int q = 0; q < VQ; ++q)
        *reinterpret_cast<float4*>(partials + (int64_t)slot * D + piece_col<T, LPR>(lig, q)) = acc[q];
    }
    continue;
  }
  // (TransE: the normalisation's dot product runs over the lanes of a group, every group has the same values)
  kn_finish<T, KIND, LPR, VPL>(acc, self, inv_self, KIND == KN_TRANSE ? (l1_self[node] > KN_EPS) : true, node, lig, dz, lddz,
                               accumulate, grp == 0);
  }
}

// long incidence lists: add the node's partial slots in chunk order, then the TransE epilogue
template <typename T, int KIND, int LPR, int VPL>
__global__ __launch_bounds__(256) void kge_bwd_nodes_fixup_kernel(bkg_rowsplit rs, const T* __restrict__ z /* self rows */,
                                                                  int64_t ldz, const float* __restrict__ l1 /* of the self rows */,
                                                                  const float* __restrict__ partials,
                                                                  float* __restrict__ dz, int64_t lddz,
                                                                  int accumulate, KnTail tl) {
  constexpr int G = 64 / LPR, Q = Piece<T>::Q, EPL = Piece<T>::EPL, VQ = VPL * Q, d = EPL * LPR * VPL;
  const int lane = threadIdx.x & 63;
  const int l = (blockIdx.x * 4 + (threadIdx.x >> 6)) * G + lane / LPR;
  const bool live = l < rs.totals[2];
  const int lig = lane % LPR;
  const int node = live ? rs.long_row[l] : 0, s0 = live ? rs.long_slot0[l] : 0, n = live ? rs.long_cnt[l] : 0;
  float4 acc[VQ], self[VQ];
  float inv_self = 0.f;
  bool above = false;
  zero4(acc);
  if (KIND == KN_TRANSE) {
    const float nrm = l1[node];
    inv_self = 1.0f / fmaxf(nrm, KN_EPS);
    above = nrm > KN_EPS;
    kn_load_self<T, LPR, VPL>(z, ldz, node, lig, inv_self, self);
  }
  constexpr int UF = 8;  // partial slots in flight (a hub node has hundreds); still added in chunk order
  for (int c0 = 0; c0 < n; c0 += UF) {
    float4 t[UF][VQ];
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const int c = min(c0 + u, n - 1);
#pragma unroll
      for (int q = 0; q < VQ; ++q) t[u][q] = kn_ld4(partials + (int64_t)(s0 + c) * d + piece_col<T, LPR>(lig, q));
    }
#pragma unroll
    for (int u = 0; u < UF; ++u)
      if (c0 + u < n) {
#pragma unroll
        for (int q = 0; q < VQ; ++q) {
          acc[q].x += t[u][q].x; acc[q].y += t[u][q].y; acc[q].z += t[u][q].z; acc[q].w += t[u][q].w;
        }
      }
  }
  kn_finish<T, KIND, LPR, VPL>(acc, self, inv_self, above, node, lig, dz, lddz, accumulate, live, tl);
}

// ------------------------------------------------------------------------------------------------ DistMult, stream form
// The kernel above reaches a node row through a chain entry id -> triplet record -> row (three dependent loads per
// incidence); the SpMM kernels reduce the very same lists (a weighted sum of gathered 512-byte rows) 2.3 x faster (0.32
// against 0.74 ms at the bench size) because their per-entry data -- column, weight -- lie IN LIST ORDER and, a wave's row
// being wave-uniform, come through scalar loads a batch ahead of the rows.  So the incidence lists get streams in list
// order: other[e] (the row to gather), relid[e], g[e] -- the first two fixed per structure (built once for the positives,
// per step for fresh negatives), g gathered per step by one elementwise pass (bkg_kge_incidence_streams) -- and this kernel
// is the SpMM item loop with one more factor: dz_i += g_e (rel[relid_e] (*) z[other_e]), the relation table in LDS (+ a
// zero row for the pad relation id of padded steps).
__global__ __launch_bounds__(256) void kge_incidence_streams_kernel(const int32_t* __restrict__ perm, int64_t n_inc, int64_t T,
                                                                    const int64_t* __restrict__ head,
                                                                    const int64_t* __restrict__ tail,
                                                                    const int64_t* __restrict__ rel,
                                                                    const int4* __restrict__ rec,
                                                                    const float* __restrict__ gscore,
                                                                    int32_t* __restrict__ other, int32_t* __restrict__ relid,
                                                                    float* __restrict__ g) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_inc) return;
  const int64_t p = perm[e];
  const bool hd = p < T;
  const int64_t t = hd ? p : p - T;
  if (rec) {                      // {head, tail, rel, bits(g)} of the fused forward pass
    const int4 v = rec[t];
    if (other) { other[e] = hd ? v.y : v.x; relid[e] = v.z; }
    if (g) g[e] = __int_as_float(v.w);
  } else {
    if (other) { other[e] = (int32_t)(hd ? tail[t] : head[t]); relid[e] = (int32_t)rel[t]; }
    if (g) g[e] = gscore[t];
  }
}

// run / run_rel (KS_RUNS): the sum of g z[other] over the current run of equal relation ids in this lane group's share of
// the list; the relation row is applied when the id changes (kn_stream_flush).  Off by default (KS_RUNS).
template <typename T, int LPR, int VPL>
__device__ __forceinline__ void kn_stream_flush(float4* acc, float4* run, int run_rel, const float* rel_lds, int lig) {
  constexpr int Q = Piece<T>::Q, EPL = Piece<T>::EPL, VQ = VPL * Q, D = EPL * LPR * VPL;
  const float* rrow = rel_lds + run_rel * D;
#pragma unroll
  for (int q = 0; q < VQ; ++q) {
    const float4 r = *reinterpret_cast<const float4*>(rrow + piece_col<T, LPR>(lig, q));
    acc[q].x = fmaf(r.x, run[q].x, acc[q].x); acc[q].y = fmaf(r.y, run[q].y, acc[q].y);
    acc[q].z = fmaf(r.z, run[q].z, acc[q].z); acc[q].w = fmaf(r.w, run[q].w, acc[q].w);
    run[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

template <typename T, int LPR, int VPL, int UT>
__device__ __forceinline__ void kn_stream_batch(float4* acc, float4* run, int& run_rel, int e, int end, int clamp,
                                                const int32_t* __restrict__ other,
                                                const int32_t* __restrict__ relid, const float* __restrict__ g,
                                                const EntryIdx<64 / LPR, UT>& idx, const T* __restrict__ z, int64_t ldz,
                                                const float* rel_lds, int grp, int lig, bool predicated) {
  constexpr int G = 64 / LPR, Q = Piece<T>::Q, EPL = Piece<T>::EPL, VQ = VPL * Q, D = EPL * LPR * VPL;
  Piece<T> val[UT][VPL];
#pragma unroll
  for (int u = 0; u < UT; ++u) {
    const T* src = z + (int64_t)idx.mine(u, grp) * ldz + EPL * lig;
#pragma unroll
    for (int v = 0; v < VPL; ++v) val[u][v].load(src + EPL * v * LPR);
  }
  EntryIdx<G, UT> rl;
  rl.load(e, clamp, relid, grp);
  EntryWt<G, UT, true> wt;
  wt.load(e, clamp, g, grp);
#pragma unroll
  for (int u = 0; u < UT; ++u) {
    if (predicated && !(e + u * G + grp < end)) continue;
    const float gu = wt.mine(u, grp);
    if (KS_RUNS) {
      const int rid = rl.mine(u, grp);
      if (rid != run_rel) {
        if (run_rel >= 0) kn_stream_flush<T, LPR, VPL>(acc, run, run_rel, rel_lds, lig);
        run_rel = rid;
      }
#pragma unroll
      for (int v = 0; v < VPL; ++v) val[u][v].fma(run + v * Q, gu);
      continue;
    }
    const float* rrow = rel_lds + rl.mine(u, grp) * D;
    float4 o[VQ];
#pragma unroll
    for (int v = 0; v < VPL; ++v) val[u][v].expand(o + v * Q);
#pragma unroll
    for (int q = 0; q < VQ; ++q) {
      const float4 r = *reinterpret_cast<const float4*>(rrow + piece_col<T, LPR>(lig, q));
      acc[q].x = fmaf(gu * r.x, o[q].x, acc[q].x); acc[q].y = fmaf(gu * r.y, o[q].y, acc[q].y);
      acc[q].z = fmaf(gu * r.z, o[q].z, acc[q].z); acc[q].w = fmaf(gu * r.w, o[q].w, acc[q].w);
    }
  }
}

template <typename T, int LPR, int VPL, int UT>
__device__ __forceinline__ void kn_stream_tail(float4* acc, float4* run, int& run_rel, int e, int end, const int32_t* __restrict__ other,
                                               const int32_t* __restrict__ relid, const float* __restrict__ g,
                                               const T* __restrict__ z, int64_t ldz, const float* rel_lds, int grp, int lig) {
  EntryIdx<64 / LPR, UT> idx;
  idx.load(e, end - 1, other, grp);
  kn_stream_batch<T, LPR, VPL, UT>(acc, run, run_rel, e, end, end - 1, other, relid, g, idx, z, ldz, rel_lds, grp, lig, true);
}

template <typename T, int LPR, int VPL>
__global__ __launch_bounds__(256) KS_ATTR void kge_bwd_stream_kernel(const T* __restrict__ z, int64_t ldz,
                                                             const float* __restrict__ rel_emb, int64_t ldr, int n_rel,
                                                             const int32_t* __restrict__ other,
                                                             const int32_t* __restrict__ relid, const float* __restrict__ g,
                                                             float* __restrict__ dz, int64_t lddz, bkg_rowsplit rs,
                                                             float* __restrict__ partials, int accumulate, KnTail tl) {
  constexpr int G = 64 / LPR, Q = Piece<T>::Q, EPL = Piece<T>::EPL, VQ = VPL * Q, D = EPL * LPR * VPL;
  constexpr int U = (VPL == 1) ? 8 : KS_U2, NB = U * G;
  extern __shared__ __attribute__((aligned(16))) float rel_lds[];
  const int lane = threadIdx.x & 63, grp = lane / LPR, lig = lane % LPR;
  for (int i = threadIdx.x; i < (n_rel + 1) * (D / 4); i += 256) {
    const int r = i / (D / 4), c = i % (D / 4);
    reinterpret_cast<float4*>(rel_lds)[i] = r < n_rel ? kn_ld4(rel_emb + (int64_t)r * ldr + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  const int n_work = rs.totals[0];
  for (int work = blockIdx.x * 4 + (threadIdx.x >> 6); work < n_work; work += gridDim.x * 4) {
    const int item = __builtin_amdgcn_readfirstlane(work);
    const int node = __builtin_amdgcn_readfirstlane(rs.item_row[item]);
    const int beg = __builtin_amdgcn_readfirstlane(rs.item_beg[item]);
    const int end = __builtin_amdgcn_readfirstlane(rs.item_end[item]);
    const int slot = __builtin_amdgcn_readfirstlane(rs.item_slot[item]);
    float4 acc[VQ], run[VQ];
    int run_rel = -1;
    zero4(acc);
    zero4(run);
    int e = beg;
    if (e + NB <= end) {
      EntryIdx<G, U> idx;
      idx.load(e, NO_CLAMP, other, grp);
      for (;;) {
        EntryIdx<G, U> cur = idx;
        const int e0 = e;
        e += NB;
        const bool more = e + NB <= end;           // wave-uniform
        if (more) idx.load(e, NO_CLAMP, other, grp);   // the next batch's rows-to-gather, under this batch's row loads
        kn_stream_batch<T, LPR, VPL, U>(acc, run, run_rel, e0, end, NO_CLAMP, other, relid, g, cur, z, ldz, rel_lds, grp, lig, false);
        if (!more) break;
      }
    }
    const int rem = end - e;                       // wave-uniform, < NB
    if (rem > 0) {
      if (rem > 4 * G) kn_stream_tail<T, LPR, VPL, (U > 4 ? U : 4)>(acc, run, run_rel, e, end, other, relid, g, z, ldz, rel_lds, grp, lig);
      else if (rem > 2 * G) kn_stream_tail<T, LPR, VPL, 4>(acc, run, run_rel, e, end, other, relid, g, z, ldz, rel_lds, grp, lig);
      else if (rem > G) kn_stream_tail<T, LPR, VPL, 2>(acc, run, run_rel, e, end, other, relid, g, z, ldz, rel_lds, grp, lig);
      else kn_stream_tail<T, LPR, VPL, 1>(acc, run, run_rel, e, end, other, relid, g, z, ldz, rel_lds, grp, lig);
    }
    if (KS_RUNS && run_rel >= 0) kn_stream_flush<T, LPR, VPL>(acc, run, run_rel, rel_lds, lig);
    if (G > 1) {
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
        for (int q = 0; q < VQ; ++q) {
          acc[q].x += __shfl_xor(acc[q].x, o, 64); acc[q].y += __shfl_xor(acc[q].y, o, 64);
          acc[q].z += __shfl_xor(acc[q].z, o, 64); acc[q].w += __shfl_xor(acc[q].w, o, 64);
        }
    }
    if (grp != 0) continue;
    if (slot >= 0) {
#pragma unroll
      for (int q = 0; q < VQ; ++q)
        *reinterpret_cast<float4*>(partials + (int64_t)slot * D + piece_col<T, LPR>(lig, q)) = acc[q];
      continue;
    }
#pragma unroll
    for (int q = 0; q < VQ; ++q) {
      float4* dst = reinterpret_cast<float4*>(dz + (int64_t)node * lddz + piece_col<T, LPR>(lig, q));
      if (accumulate) {
        const float4 o = tl.acc_src ? *reinterpret_cast<const float4*>(tl.acc_src + (int64_t)node * tl.ld_src + piece_col<T, LPR>(lig, q))
                                    : *dst;
        acc[q].x += o.x; acc[q].y += o.y; acc[q].z += o.z; acc[q].w += o.w;
      }
      kn_tail(acc[q], tl, node, piece_col<T, LPR>(lig, q));
      *dst = acc[q];
    }
  }
}


// ------------------------------------------------------------------------------------------------ DistMult, ONE pass (positives)
// A graph's positive triplets are the same every step, and on a knowledge graph a node's edges of one relation come in
// RUNS (the bench graph: 19 entries per (head, relation) run): with a node's incidence list sorted by (role, relation) the
// list walk can do the scorer's work too.  z_n stays in registers; per run zr = z_n (*) rel_r; per entry the OTHER
// endpoint's row o gives s = zr . o (DistMult is symmetric: the same score from either end), g = (sigmoid(s) - y) / total
// and M += g o; at the end of a run dz_n += rel_r (*) M -- and, where n is the head, one ROW z_n (*) M of d rel_emb's sum
// leaves for a record buffer (207 k rows on the bench graph, added per relation by the caller in a fixed order: no atomics
// anywhere, bitwise repeatable).  Every triplet's rows are read once per incidence -- 2 T row reads where the two-pass
// form (relation-ordered scorer + node-ordered gradient) reads 4 T -- and g never travels.  The lists, their runs and
// the batches the kernel walks (<= NB entries of ONE run each, so that the role / relation of a batch is wave-uniform) are
// built once per graph (biomedkg_amd/ops.py OnePassLists).  Negatives are fresh every step and uniformly random -- their
// runs are single entries -- and keep the two-pass form.
#ifndef KO_U
#define KO_U 2          // entries in flight per lane group (the batch table is cut for NB = KO_U * 64 / lanes-per-row)
#endif
#ifndef KO_L32
#define KO_L32 16       // lanes per row of a 32-piece table (d = 256 bf16)
#endif

// The sum over a lane group of 2..16 lanes, in every lane of it, by DPP partners inside the 16-lane row (row_ror:8,
// row_half_mirror, quad_perm) -- VALU instructions, no LDS crossbar round trip (the __shfl_xor butterfly is four
// ds_bpermute + lgkmcnt waits per score); wider groups finish with the shuffles.
template <int CTRL>
__device__ __forceinline__ float ko_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int LPR>
__device__ __forceinline__ float ko_group_sum(float v) {
  if constexpr (LPR > 16) {
#pragma unroll
    for (int o = LPR / 2; o >= 16; o >>= 1) v += __shfl_xor(v, o, 64);
  }
  if constexpr (LPR >= 16) v += ko_dpp<0x128>(v);      // row_ror:8        i <-> i ^ 8
  if constexpr (LPR >= 8) v += ko_dpp<0x141>(v);       // row_half_mirror  i <-> i ^ 7 (the other quad of the eight)
  if constexpr (LPR >= 4) v += ko_dpp<0x4E>(v);        // quad_perm [2,3,0,1]
  if constexpr (LPR >= 2) v += ko_dpp<0xB1>(v);        // quad_perm [1,0,3,2]
  return v;
}

template <typename T, int LPR, int VPL>
__global__ __launch_bounds__(256) void kge_onepass_kernel(const T* __restrict__ z, int64_t ldz,
                                                          const float* __restrict__ rel_emb, int64_t ldr, int n_rel,
                                                          const int32_t* __restrict__ other, const int32_t* __restrict__ tid,
                                                          const int4* __restrict__ batches, const int4* __restrict__ items,
                                                          int n_items, float label, float inv_t, float* __restrict__ score,
                                                          float* __restrict__ loss_part, float* __restrict__ records,
                                                          float* __restrict__ dz, int64_t lddz, float* __restrict__ partials) {
  constexpr int G = 64 / LPR, Q = Piece<T>::Q, EPL = Piece<T>::EPL, VQ = VPL * Q, D = EPL * LPR * VPL, U = KO_U;
  extern __shared__ __attribute__((aligned(16))) float ko_rel[];
  const int lane = threadIdx.x & 63, grp = lane / LPR, lig = lane % LPR;
  float* scratch = ko_rel + n_rel * D + (threadIdx.x >> 6) * ((G > 1 ? G - 1 : 1) * D);      // this wave's (G - 1) rows
  for (int i = threadIdx.x; i < n_rel * (D / 4); i += 256) {
    const int r = i / (D / 4), c = i % (D / 4);
    reinterpret_cast<float4*>(ko_rel)[i] = kn_ld4(rel_emb + (int64_t)r * ldr + 4 * c);
  }
  __syncthreads();
  float loss_acc = 0.f;
  for (int work = blockIdx.x * 4 + (threadIdx.x >> 6); work < n_items; work += gridDim.x * 4) {
    const int item = __builtin_amdgcn_readfirstlane(work);
    const int4 it = items[item];                      // {node, first batch, end batch, partial slot or -1}
    const int node = __builtin_amdgcn_readfirstlane(it.x), b0 = __builtin_amdgcn_readfirstlane(it.y);
    const int b1 = __builtin_amdgcn_readfirstlane(it.z), slot = __builtin_amdgcn_readfirstlane(it.w);
    float4 acc[VQ], zn[VQ], zr[VQ], M[VQ];
    zero4(acc);
    zero4(M);
    zero4(zr);
    {
      const T* src = z + (int64_t)node * ldz + EPL * lig;
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        Piece<T> pc;
        pc.load(src + EPL * v * LPR);
        pc.expand(zn + v * Q);
      }
    }
    // Software pipeline over the item's batches: the rows of batch b + 1, the indices of batch b + 2 and the record of
    // batch b + 3 are in flight while batch b is computed (the chain record -> indices -> rows would otherwise cost three
    // memory latencies per batch: 1.2 ms per launch on the bench graph).  Past the item's last batch the last one is
    // re-read (cache hits, never used).
    auto rec_at = [&](int bb) { return batches[min(bb, b1 - 1)]; };
    auto idx_of = [&](const int4& br_, int (&oi_)[U], int (&ti_)[U]) {
      const int e0_ = __builtin_amdgcn_readfirstlane(br_.x), cnt_ = __builtin_amdgcn_readfirstlane(br_.y) & 0xff;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int e = e0_ + min(u * G + grp, cnt_ - 1);
        oi_[u] = other[e];
        ti_[u] = tid[e];
      }
    };
    auto rows_of = [&](const int (&oi_)[U], Piece<T> (&val_)[U][VPL]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const T* src = z + (int64_t)oi_[u] * ldz + EPL * lig;
#pragma unroll
        for (int v = 0; v < VPL; ++v) val_[u][v].load(src + EPL * v * LPR);
      }
    };
    int4 r0 = make_int4(0, 0, 0, 0), r1 = r0, r2 = r0;
    int oi0[U], ti0[U], oi1[U], ti1[U];
    Piece<T> val[U][VPL], valn[U][VPL];
    if (b0 < b1) {
      r0 = rec_at(b0); r1 = rec_at(b0 + 1); r2 = rec_at(b0 + 2);
      idx_of(r0, oi0, ti0);
      idx_of(r1, oi1, ti1);
      rows_of(oi0, val);
    }
    for (int b = b0; b < b1; ++b) {
      const int4 br = r0;                             // {first entry, count | flags << 8, relation + n_rel * role, record row}
      int ti[U];
#pragma unroll
      for (int u = 0; u < U; ++u) ti[u] = ti0[u];
      rows_of(oi1, valn);                             // batch b + 1
      int oi2[U], ti2[U];
      idx_of(r2, oi2, ti2);                           // batch b + 2
      const int4 r3 = rec_at(b + 3);
      const int cf = __builtin_amdgcn_readfirstlane(br.y);
      const int relrole = __builtin_amdgcn_readfirstlane(br.z), rec = __builtin_amdgcn_readfirstlane(br.w);
      const int cnt = cf & 0xff;
      const bool first = (cf & 0x100) != 0, last = (cf & 0x200) != 0, head = relrole < n_rel;
      const float* rrow = ko_rel + (head ? relrole : relrole - n_rel) * D;
      if (first) {
#pragma unroll
        for (int q = 0; q < VQ; ++q) {
          const float4 r = *reinterpret_cast<const float4*>(rrow + piece_col<T, LPR>(lig, q));
          zr[q] = make_float4(zn[q].x * r.x, zn[q].y * r.y, zn[q].z * r.z, zn[q].w * r.w);
          M[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (u > 0 && u * G >= cnt) break;             // (wave-uniform: a short batch's empty rounds cost nothing)
        const bool live = u * G + grp < cnt;
        float4 o[VQ];
#pragma unroll
        for (int v = 0; v < VPL; ++v) val[u][v].expand(o + v * Q);
        float sp = 0.f;
#pragma unroll
        for (int q = 0; q < VQ; ++q) {
          sp = fmaf(zr[q].x, o[q].x, sp); sp = fmaf(zr[q].y, o[q].y, sp);
          sp = fmaf(zr[q].z, o[q].z, sp); sp = fmaf(zr[q].w, o[q].w, sp);
        }
        const float sc = ko_group_sum<LPR>(sp);
        // BCE with logits (torch formula): max(s,0) - s*y + log1p(exp(-|s|));  d/ds = sigmoid(s) - y.  Every triplet is
        // scored from both of its ends here: the hardware exp2 / rcp / log2 (1 ulp) instead of the library forms
        const float ex = __builtin_amdgcn_exp2f(-1.4426950408889634f * fabsf(sc));
        const float rc = __builtin_amdgcn_rcpf(1.0f + ex);
        const float sig = sc >= 0.f ? rc : ex * rc;
        const float g = live ? (sig - label) * inv_t : 0.f;
        if (head && live && lig == 0) {               // (head: wave-uniform)
          if (score) score[ti[u]] = sc;
          loss_acc += fmaxf(sc, 0.f) - sc * label + 0.6931471805599453f * __builtin_amdgcn_logf(1.0f + ex);
        }
#pragma unroll
        for (int q = 0; q < VQ; ++q) {
          M[q].x = fmaf(g, o[q].x, M[q].x); M[q].y = fmaf(g, o[q].y, M[q].y);
          M[q].z = fmaf(g, o[q].z, M[q].z); M[q].w = fmaf(g, o[q].w, M[q].w);
        }
      }
      if (last) {
        // the groups' sums of the run meet in group 0 through this wave's LDS scratch (G - 1 rows written, read back by
        // group 0: 16-byte LDS accesses, no cross-lane instruction; a wave's LDS operations complete in order)
        if (G > 1) {
          float* mine = scratch + (grp > 0 ? grp - 1 : 0) * D;
          if (grp > 0) {
#pragma unroll
            for (int q = 0; q < VQ; ++q) *reinterpret_cast<float4*>(mine + piece_col<T, LPR>(lig, q)) = M[q];
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          if (grp == 0) {
#pragma unroll
            for (int o = 0; o < G - 1; ++o)
#pragma unroll
              for (int q = 0; q < VQ; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(scratch + o * D + piece_col<T, LPR>(lig, q));
                M[q].x += v.x; M[q].y += v.y; M[q].z += v.z; M[q].w += v.w;
              }
          }
        }
        if (grp == 0) {
#pragma unroll
          for (int q = 0; q < VQ; ++q) {
            const float4 r = *reinterpret_cast<const float4*>(rrow + piece_col<T, LPR>(lig, q));
            acc[q].x = fmaf(r.x, M[q].x, acc[q].x); acc[q].y = fmaf(r.y, M[q].y, acc[q].y);
            acc[q].z = fmaf(r.z, M[q].z, acc[q].z); acc[q].w = fmaf(r.w, M[q].w, acc[q].w);
            if (head)
              *reinterpret_cast<float4*>(records + (int64_t)rec * D + piece_col<T, LPR>(lig, q)) =
                  make_float4(zn[q].x * M[q].x, zn[q].y * M[q].y, zn[q].z * M[q].z, zn[q].w * M[q].w);
          }
        }
      }
      r0 = r1; r1 = r2; r2 = r3;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        oi0[u] = oi1[u]; ti0[u] = ti1[u]; oi1[u] = oi2[u]; ti1[u] = ti2[u];
#pragma unroll
        for (int v = 0; v < VPL; ++v) val[u][v] = valn[u][v];
      }
    }
    if (grp != 0) continue;
    if (slot >= 0) {
#pragma unroll
      for (int q = 0; q < VQ; ++q)
        *reinterpret_cast<float4*>(partials + (int64_t)slot * D + piece_col<T, LPR>(lig, q)) = acc[q];
      continue;
    }
#pragma unroll
    for (int q = 0; q < VQ; ++q)
      *reinterpret_cast<float4*>(dz + (int64_t)node * lddz + piece_col<T, LPR>(lig, q)) = acc[q];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) loss_acc += __shfl_xor(loss_acc, o, 64);
  if (lane == 0) loss_part[blockIdx.x * 4 + (threadIdx.x >> 6)] = loss_acc;
}

}  // namespace bkg

using namespace bkg;

extern "C" int bkg_kge_bwd_nodes_f32(int32_t kind, const void* z, int32_t z_is_bf16, int64_t ldz, int64_t n_nodes,
                                     int32_t d, const int64_t* head, const int64_t* tail, const int64_t* rel,
                                     int64_t T, const float* rel_emb, int64_t ldr, int64_t n_rel, const float* gscore,
                                     const void* rec, const float* l1, const int32_t* inc_rowptr, const int32_t* inc_perm, float* dz,
                                     int64_t lddz, const bkg_rowsplit* split, float* partials, int32_t accumulate,
                                     int64_t self_row0, void* stream) {
  BKG_ARG(kind == KN_DISTMULT || kind == KN_TRANSE, "bkg_kge_bwd_nodes_f32: bad kind");
  BKG_ARG(self_row0 >= 0, "bkg_kge_bwd_nodes_f32: bad self_row0");
  // incidence row i is node `self_row0 + i` of the table z / l1 (0 for a whole graph; a node-partitioned rank reduces the
  // lists of ITS nodes only, whose rows of the gathered table start at its first slot): TransE reads the node's own row
  const float* l1_self = l1 ? l1 + self_row0 : nullptr;
  BKG_ARG(kind == KN_DISTMULT || l1 != nullptr, "bkg_kge_bwd_nodes_f32: TransE needs the row L1 norms");
  const int epl = z_is_bf16 ? 8 : 4;
  BKG_ARG(((uintptr_t)z % 16 == 0) && (ldz % epl == 0) && (d % epl == 0) && ((uintptr_t)rel_emb % 16 == 0) &&
              (ldr % 4 == 0) && ((uintptr_t)dz % 16 == 0) && (lddz % 4 == 0), "bkg_kge_bwd_nodes_f32: alignment");
  BKG_ARG(!split || (partials && ((uintptr_t)partials % 16 == 0)), "bkg_kge_bwd_nodes_f32: splitting needs partials");
  if (n_nodes == 0) return 0;
  KnTriplets tr{head, tail, rel, T, reinterpret_cast<const int4*>(rec)};
  hipStream_t s = (hipStream_t)stream;
  dim3 block(256);
  bkg_rowsplit none{};
  /* relation table in LDS (and a strided walk over the items) when it fits 64 KB and there is enough work to amortise   \
     the staging; else one block per four items, rows from the caches */                                               \
  const bool rel_lds = KN_REL_LDS && n_rel > 0 && (int64_t)(n_rel + 1) * d * 4 <= 65536;                                     \
  const size_t smem = rel_lds ? (size_t)(n_rel + 1) * d * 4 : 0;
#define KN_GRID(work) dim3((unsigned)std::max<int64_t>(1, rel_lds ? std::min<int64_t>(ceil_div(work, 4), KN_PERSIST_BLOCKS) \
                                                                   : ceil_div(work, 4)))
#define KN_LAUNCH2(TT, K, SP, RL, work, RS)                                                                            \
  hipLaunchKernelGGL((kge_bwd_nodes_kernel<TT, K, LPR, VPL, SP, RL>), KN_GRID(work), block, RL ? smem : 0, s,          \
                     (const TT*)z, (const TT*)z + self_row0 * ldz, ldz, (int)n_nodes, tr, rel_emb, ldr, (int)n_rel, gscore, l1, \
                     l1_self, inc_rowptr, inc_perm,                                                                    \
                     dz, lddz, RS, partials, accumulate)
#define KN_LAUNCH(TT, K)                                                                                               \
  if (split) {                                                                                                         \
    if (split->n_items_max > 0) {                                                                                      \
      if (rel_lds) KN_LAUNCH2(TT, K, true, true, split->n_items_max, *split);                                          \
      else KN_LAUNCH2(TT, K, true, false, split->n_items_max, *split);                                                 \
    }                                                                                                                  \
    if (split->n_long_max > 0)   /* (bounds cut to the table's actual counts may be zero) */                          \
    hipLaunchKernelGGL((kge_bwd_nodes_fixup_kernel<TT, K, LPR, VPL>), dim3(ceil_div(split->n_long_max, 4 * (64 / LPR))), \
                       block, 0, s, *split, (const TT*)z + self_row0 * ldz, ldz, l1_self, partials, dz, lddz, accumulate, \
                       KnTail{nullptr, nullptr, 0, 0.f, nullptr, 0});                                                              \
  } else {                                                                                                             \
    if (rel_lds) KN_LAUNCH2(TT, K, false, true, n_nodes, none);                                                        \
    else KN_LAUNCH2(TT, K, false, false, n_nodes, none);                                                               \
  }
  BKG_PIECE_DISPATCH_(d / epl, 16, 16, {
    if (z_is_bf16) { if (kind == KN_DISTMULT) { KN_LAUNCH(__bf16, KN_DISTMULT) } else { KN_LAUNCH(__bf16, KN_TRANSE) } }
    else { if (kind == KN_DISTMULT) { KN_LAUNCH(float, KN_DISTMULT) } else { KN_LAUNCH(float, KN_TRANSE) } }
  })
#undef KN_LAUNCH
#undef KN_LAUNCH2
#undef KN_GRID
  return check_launch("bkg_kge_bwd_nodes_f32");
}

// The per-incidence streams of bkg_kge_bwd_stream_f32 in list order (perm: the incidence permutation of the 2T entries
// cat(head, tail)): other[e] = the other endpoint, relid[e], g[e] = d score.  other / relid nullable (a cached structure
// whose g alone changes), g nullable; rec (nullable): the fused pass's records instead of head / tail / rel / gscore.
extern "C" int bkg_kge_incidence_streams(const int32_t* perm, int64_t n_inc, int64_t T, const int64_t* head,
                                         const int64_t* tail, const int64_t* rel, const void* rec, const float* gscore,
                                         int32_t* other, int32_t* relid, float* g, void* stream) {
  BKG_ARG(n_inc >= 0 && T >= 0 && perm && (rec || (head && tail && rel)) && (!other == !relid) && (other || g) &&
              (!g || rec || gscore), "bkg_kge_incidence_streams: bad arguments");
  if (n_inc == 0) return 0;
  hipLaunchKernelGGL(kge_incidence_streams_kernel, dim3(ceil_div(n_inc, 256)), dim3(256), 0, (hipStream_t)stream, perm, n_inc, T,
                     head, tail, rel, reinterpret_cast<const int4*>(rec), gscore, other, relid, g);
  return check_launch("bkg_kge_incidence_streams");
}

// DistMult's node-side backward over those streams: dz[i] (+)= sum over the incidence list of node i of
// g[e] * rel_emb[relid[e]] (*) z[other[e]] (split: the row table of the lists; relation ids <= n_rel, id n_rel = zero row).
extern "C" int bkg_kge_bwd_stream_f32(const void* z, int32_t z_is_bf16, int64_t ldz, int64_t n_nodes, int32_t d,
                                      const float* rel_emb, int64_t ldr, int64_t n_rel, const int32_t* other,
                                      const int32_t* relid, const float* g, float* dz, int64_t lddz,
                                      const bkg_rowsplit* split, float* partials, int32_t accumulate, const float* tail_gloss,
                                      const float* tail_z, int64_t ld_tail_z, float tail_coef, const float* acc_src,
                                      int64_t ld_acc_src, void* stream) {
  const int epl = z_is_bf16 ? 8 : 4;
  BKG_ARG(!tail_z || (tail_gloss && (uintptr_t)tail_z % 16 == 0 && ld_tail_z % 4 == 0),
          "bkg_kge_bwd_stream_f32: the tail's rows need its scale and 16-byte aligned rows");
  BKG_ARG(!acc_src || (accumulate && (uintptr_t)acc_src % 16 == 0 && ld_acc_src % 4 == 0),
          "bkg_kge_bwd_stream_f32: acc_src goes with accumulate, 16-byte aligned rows");
  const KnTail tl{tail_gloss, tail_z, ld_tail_z, tail_coef, acc_src, ld_acc_src};
  BKG_ARG(split && partials && other && relid && g && n_rel > 0, "bkg_kge_bwd_stream_f32: needs the row table, its partials and the streams");
  BKG_ARG(((uintptr_t)z % 16 == 0) && (ldz % epl == 0) && (d % epl == 0) && ((uintptr_t)rel_emb % 16 == 0) && (ldr % 4 == 0) &&
              ((uintptr_t)dz % 16 == 0) && (lddz % 4 == 0) && ((uintptr_t)partials % 16 == 0), "bkg_kge_bwd_stream_f32: alignment");
  BKG_ARG((n_rel + 1) * (int64_t)d * 4 <= 65536, "bkg_kge_bwd_stream_f32: the relation table must fit 64 KB of LDS");
  if (n_nodes == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  dim3 block(256);
  const size_t smem = (size_t)(n_rel + 1) * d * 4;
#define KS_LAUNCH(TT)                                                                                                  \
  if (split->n_items_max > 0)                                                                                          \
    hipLaunchKernelGGL((kge_bwd_stream_kernel<TT, LPR, VPL>),                                                          \
                       dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(split->n_items_max, 4), KN_PERSIST_BLOCKS))), \
                       block, smem, s, (const TT*)z, ldz, rel_emb, ldr, (int)n_rel, other, relid, g, dz, lddz, *split,  \
                       partials, accumulate, tl);                                                                      \
  if (split->n_long_max > 0)                                                                                           \
    hipLaunchKernelGGL((kge_bwd_nodes_fixup_kernel<TT, KN_DISTMULT, LPR, VPL>),                                        \
                       dim3(ceil_div(split->n_long_max, 4 * (64 / LPR))), block, 0, s, *split, (const TT*)z, ldz,      \
                       (const float*)nullptr, partials, dz, lddz, accumulate, tl);
  BKG_PIECE_DISPATCH_(d / epl, KS_L32, 64, {
    if (z_is_bf16) { KS_LAUNCH(__bf16) } else { KS_LAUNCH(float) }
  })
#undef KS_LAUNCH
  return check_launch("bkg_kge_bwd_stream_f32");
}

// The whole DistMult TRAINING pass of a graph's positive triplets in ONE walk over (role, relation)-sorted incidence lists
// (kge_onepass_kernel): scores, BCE terms, dz, and one record row z_n (*) M per (head, relation) run -- the caller adds the
// records of each relation (a fixed-order segment reduce) into d rel_emb.  items [n_items] = {node, first batch, end batch,
// partial slot}; batches = {first entry, count | first << 8 | last << 9, relation + n_rel * role, record row}; the batches
// of an item hold <= bkg_kge_onepass_batch(d, z_is_bf16) entries of one run each.  split: the row table the items were
// cut by (its long rows' partial slots are added by the fix-up launch); dz is written, not added to.
// lanes per row as BKG_PIECE_DISPATCH_(d / epl, KO_L32, 64) picks them; 0: a width the walk has no instance for
static int onepass_lanes_per_row(int32_t d, int32_t z_is_bf16) {
  const int epl = z_is_bf16 ? 8 : 4;
  if (d <= 0 || d % epl) return 0;
  switch (d / epl) {
    case 2: return 2;   case 4: return 4;   case 6: return 2;   case 8: return 8;    case 12: return 4;   case 16: return 16;
    case 24: return 8;  case 32: return KO_L32;                 case 64: return 64;  case 96: return 32;
    case 128: case 192: return 64;
    default: return 0;
  }
}
extern "C" int32_t bkg_kge_onepass_batch(int32_t d, int32_t z_is_bf16) {
  const int lpr = onepass_lanes_per_row(d, z_is_bf16);
  return lpr ? KO_U * (64 / lpr) : 0;                 // 0: unsupported width (the caller keeps the two-pass form)
}
// Dynamic LDS of one workgroup of the walk: the relation table + every wave's scratch ((lane groups - 1) rows of d floats,
// four waves); 0 for a width without an instance.  The caller takes the one-pass form only when this fits 64 KB (ADVICE
// r05: narrow tables have many lane groups -- d = 64 bf16: 28 scratch rows -- so n_rel * d * 4 alone does not decide).
extern "C" int64_t bkg_kge_onepass_smem_bytes(int32_t d, int32_t z_is_bf16, int64_t n_rel) {
  const int lpr = onepass_lanes_per_row(d, z_is_bf16);
  if (!lpr || n_rel <= 0) return 0;
  return ((int64_t)n_rel + 4 * (int64_t)std::max(64 / lpr - 1, 1)) * d * 4;
}
#ifndef KO_BLOCKS
#define KO_BLOCKS KN_PERSIST_BLOCKS
#endif
extern "C" int32_t bkg_kge_onepass_blocks(int64_t n_items) {
  return (int32_t)std::max<int64_t>(1, std::min<int64_t>(ceil_div(std::max<int64_t>(n_items, 1), 4), KO_BLOCKS));
}
extern "C" int bkg_kge_onepass_f32(const void* z, int32_t z_is_bf16, int64_t ldz, int64_t n_nodes, int32_t d, const float* rel_emb,
                                   int64_t ldr, int64_t n_rel, const int32_t* other, const int32_t* tid, const void* batches,
                                   const void* items, int64_t n_items, float label, float inv_total, float* score,
                                   float* loss_part, float* records, float* dz, int64_t lddz, const bkg_rowsplit* split,
                                   float* partials, void* stream) {
  const int epl = z_is_bf16 ? 8 : 4;
  BKG_ARG(other && tid && batches && items && loss_part && records && dz && split && partials && n_rel > 0 && n_items >= 0,
          "bkg_kge_onepass_f32: needs the lists, their batch / item tables, the row table and the output buffers");
  BKG_ARG(((uintptr_t)z % 16 == 0) && (ldz % epl == 0) && (d % epl == 0) && ((uintptr_t)rel_emb % 16 == 0) && (ldr % 4 == 0) &&
              ((uintptr_t)dz % 16 == 0) && (lddz % 4 == 0) && ((uintptr_t)partials % 16 == 0) && ((uintptr_t)records % 16 == 0) &&
              ((uintptr_t)batches % 16 == 0) && ((uintptr_t)items % 16 == 0), "bkg_kge_onepass_f32: alignment");
  {
    const int64_t smem_need = bkg_kge_onepass_smem_bytes(d, z_is_bf16, n_rel);
    BKG_ARG(smem_need > 0 && smem_need <= 65536,
            "bkg_kge_onepass_f32: the relation table + the waves' scratch rows must fit 64 KB of LDS (bkg_kge_onepass_smem_bytes)");
  }
  if (n_nodes == 0 || n_items == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const unsigned blocks = (unsigned)bkg_kge_onepass_blocks(n_items);
#define KO_LAUNCH(TT)                                                                                                  \
  {                                                                                                                    \
    /* the relation table + every wave's scratch: (lane groups - 1) rows of d floats */                                \
    const size_t smem = ((size_t)n_rel + 4 * (size_t)std::max(64 / LPR - 1, 1)) * d * 4; /* = bkg_kge_onepass_smem_bytes */ \
    hipLaunchKernelGGL((kge_onepass_kernel<TT, LPR, VPL>), dim3(blocks), dim3(256), smem, s, (const TT*)z, ldz, rel_emb, ldr, \
                       (int)n_rel, other, tid, (const int4*)batches, (const int4*)items, (int)n_items, label, inv_total, score, \
                       loss_part, records, dz, lddz, partials);                                                        \
    if (split->n_long_max > 0)                                                                                         \
      hipLaunchKernelGGL((kge_bwd_nodes_fixup_kernel<TT, KN_DISTMULT, LPR, VPL>),                                      \
                         dim3(ceil_div(split->n_long_max, 4 * (64 / LPR))), dim3(256), 0, s, *split, (const TT*)z, ldz, \
                         (const float*)nullptr, partials, dz, lddz, 0, KnTail{nullptr, nullptr, 0, 0.f, nullptr, 0});              \
  }
  BKG_PIECE_DISPATCH_(d / epl, KO_L32, 64, {
    if (z_is_bf16) KO_LAUNCH(__bf16) else KO_LAUNCH(float)
  })
#undef KO_LAUNCH
  return check_launch("bkg_kge_onepass_f32");
}
