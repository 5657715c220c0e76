// KGE triplet scorers (DistMult / TransE) -- forward and a non-atomic backward.
//
// Reference arithmetic: /root/reference/biomedkg/model/decoder.py:31-38 (TransE: L1-normalise head
// and tail with eps 1e-12, score = -||h^ + r - t^||_1) and :45-49 (DistMult: sum_k h_k r_k t_k).
// The reference materialises three [T,d] gathers per call; here each triplet's two rows are read
// once per pass straight into registers (rel_emb is a few KB and stays cache-resident).
//
// forward   : LPR lanes x VPL float4 cover one d-wide row; the 64/LPR lane groups of a wave take
//             different triplets; in-group shuffle reduction -> one score store per triplet.
// backward dz: NOT a scatter-add.  The caller sorts the 2T (node, role) incidences by node once
//             (csrc/graph.hip radix sort) and this kernel does a per-node segment reduce with
//             plain stores: dz[i] = sum over incident triplets of the contribution of the OTHER
//             endpoint.  Bitwise reproducible, and ~4-5x the float-atomic rate on gfx950.
// backward drel: triplet-parallel; per-block [R,d] accumulator in LDS (ds_add_f32), one flush of
//             float atomics per block into the (zeroed) [R,d] result -- R*d*4 bytes per block,
//             negligible against the 1.3 TB/s atomic budget.
#include "common.h"

namespace bkg {

enum { KGE_DISTMULT = 0, KGE_TRANSE = 1 };
constexpr float L1_EPS = 1e-12f;

struct Triplets {
  const int64_t* head;
  const int64_t* tail;
  const int64_t* rel;
  int64_t T;
};

__device__ __forceinline__ float sgn(float v) { return (v > 0.f) - (v < 0.f); }

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ------------------------------------------------------------------------------------ forward
template <int KIND, int LPR, int VPL>
__global__ __launch_bounds__(256) void kge_score_kernel(const float* __restrict__ z, int64_t ldz, Triplets tr,
                                                        const float* __restrict__ rel_emb, int64_t ldr,
                                                        float* __restrict__ score) {
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63, grp = lane / LPR, lig = lane % LPR;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  for (int64_t base = wave * G; base < tr.T; base += n_waves * G) {
    const int64_t t = base + grp;
    const bool live = t < tr.T;
    const int64_t hi = live ? tr.head[t] : 0, ti = live ? tr.tail[t] : 0, ri = live ? tr.rel[t] : 0;
    float4 h[VPL], tl[VPL], r[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = 4 * (lig + v * LPR);
      h[v] = ld4(z + hi * ldz + c);
      tl[v] = ld4(z + ti * ldz + c);
      r[v] = ld4(rel_emb + ri * ldr + c);
    }
    float s = 0.f;
    if (KIND == KGE_DISTMULT) {
#pragma unroll
      for (int v = 0; v < VPL; ++v)
        s += h[v].x * r[v].x * tl[v].x + h[v].y * r[v].y * tl[v].y + h[v].z * r[v].z * tl[v].z + h[v].w * r[v].w * tl[v].w;
      s = group_sum<LPR>(s);
    } else {
      float nh = 0.f, nt = 0.f;
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        nh += fabsf(h[v].x) + fabsf(h[v].y) + fabsf(h[v].z) + fabsf(h[v].w);
        nt += fabsf(tl[v].x) + fabsf(tl[v].y) + fabsf(tl[v].z) + fabsf(tl[v].w);
      }
      nh = group_sum<LPR>(nh);
      nt = group_sum<LPR>(nt);
      const float ih = 1.0f / fmaxf(nh, L1_EPS), it = 1.0f / fmaxf(nt, L1_EPS);
#pragma unroll
      for (int v = 0; v < VPL; ++v)
        s += fabsf(h[v].x * ih + r[v].x - tl[v].x * it) + fabsf(h[v].y * ih + r[v].y - tl[v].y * it) +
             fabsf(h[v].z * ih + r[v].z - tl[v].z * it) + fabsf(h[v].w * ih + r[v].w - tl[v].w * it);
      s = -group_sum<LPR>(s);
    }
    if (live && lig == 0) score[t] = s;
  }
}

// ------------------------------------------------------------------------------------ row L1 norms
template <int LPR, int VPL>
__global__ __launch_bounds__(256) void row_l1_kernel(const float* __restrict__ z, int64_t ldz, int64_t n,
                                                     float* __restrict__ l1) {
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63, grp = lane / LPR, lig = lane % LPR;
  const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * G + grp;
  float s = 0.f;
  if (row < n) {
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      float4 a = ld4(z + row * ldz + 4 * (lig + v * LPR));
      s += fabsf(a.x) + fabsf(a.y) + fabsf(a.z) + fabsf(a.w);
    }
  }
  s = group_sum<LPR>(s);
  if (row < n && lig == 0) l1[row] = s;
}

// ------------------------------------------------------------------------------------ backward: dz
// inc_perm[k] in [0, 2T): entry e < T -> node is the HEAD of triplet e; e >= T -> TAIL of e - T.
// TransE: acc = Q = d loss / d z^_i.  Through z^ = z / max(||z||_1, eps):
//   ||z||_1 > eps : dz = (Q - sign(z) (z^ . Q)) / ||z||_1     else : dz = Q / eps
template <int KIND, int LPR, int VPL>
__device__ __forceinline__ void kge_nodes_finish(float4 (&acc)[VPL], const float4 (&self)[VPL], float inv_self,
                                                 bool above_eps, int node, int lig, float* __restrict__ dz,
                                                 int64_t lddz, int accumulate = 0) {
  if (KIND == KGE_TRANSE) {
    float dot = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v)
      dot += self[v].x * acc[v].x + self[v].y * acc[v].y + self[v].z * acc[v].z + self[v].w * acc[v].w;
    dot = group_sum<LPR>(dot);
    if (!above_eps) dot = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      acc[v].x = (acc[v].x - sgn(self[v].x) * dot) * inv_self; acc[v].y = (acc[v].y - sgn(self[v].y) * dot) * inv_self;
      acc[v].z = (acc[v].z - sgn(self[v].z) * dot) * inv_self; acc[v].w = (acc[v].w - sgn(self[v].w) * dot) * inv_self;
    }
  }
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    float4* dst = reinterpret_cast<float4*>(dz + (int64_t)node * lddz + 4 * (lig + v * LPR));
    if (accumulate) {
      const float4 o = *dst;
      acc[v].x += o.x; acc[v].y += o.y; acc[v].z += o.z; acc[v].w += o.w;
    }
    *dst = acc[v];
  }
}

// long incidence lists: add the node's partial slots in chunk order, then the TransE epilogue
template <int KIND, int LPR, int VPL>
__global__ __launch_bounds__(256) void kge_bwd_nodes_fixup_kernel(bkg_rowsplit rs, const float* __restrict__ z,
                                                                  int64_t ldz, const float* __restrict__ l1,
                                                                  const float* __restrict__ partials,
                                                                  float* __restrict__ dz, int64_t lddz,
                                                                  int accumulate) {
  constexpr int G = 64 / LPR, d = 4 * LPR * VPL;
  const int lane = threadIdx.x & 63;
  const int l = (blockIdx.x * 4 + (threadIdx.x >> 6)) * G + lane / LPR;
  const bool live = l < rs.totals[2];
  const int lig = lane % LPR;
  const int node = live ? rs.long_row[l] : 0, s0 = live ? rs.long_slot0[l] : 0, n = live ? rs.long_cnt[l] : 0;
  float4 acc[VPL], self[VPL];
  float inv_self = 0.f;
  bool above = false;
#pragma unroll
  for (int v = 0; v < VPL; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (KIND == KGE_TRANSE) {
    const float nrm = l1[node];
    inv_self = 1.0f / fmaxf(nrm, L1_EPS);
    above = nrm > L1_EPS;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const float4 a = ld4(z + (int64_t)node * ldz + 4 * (lig + v * LPR));
      self[v] = make_float4(a.x * inv_self, a.y * inv_self, a.z * inv_self, a.w * inv_self);
    }
  }
  for (int c = 0; c < n; ++c) {
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const float4 t = ld4(partials + (int64_t)(s0 + c) * d + 4 * (lig + v * LPR));
      acc[v].x += t.x; acc[v].y += t.y; acc[v].z += t.z; acc[v].w += t.w;
    }
  }
  if (KIND == KGE_TRANSE) {  // the in-group reduction below needs every lane of the group
    float dot = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v)
      dot += self[v].x * acc[v].x + self[v].y * acc[v].y + self[v].z * acc[v].z + self[v].w * acc[v].w;
    dot = group_sum<LPR>(dot);
    if (!above) dot = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      acc[v].x = (acc[v].x - sgn(self[v].x) * dot) * inv_self; acc[v].y = (acc[v].y - sgn(self[v].y) * dot) * inv_self;
      acc[v].z = (acc[v].z - sgn(self[v].z) * dot) * inv_self; acc[v].w = (acc[v].w - sgn(self[v].w) * dot) * inv_self;
    }
  }
  if (live) {
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      float4* dst = reinterpret_cast<float4*>(dz + (int64_t)node * lddz + 4 * (lig + v * LPR));
      if (accumulate) {
        const float4 o = *dst;
        acc[v].x += o.x; acc[v].y += o.y; acc[v].z += o.z; acc[v].w += o.w;
      }
      *dst = acc[v];
    }
  }
}

// SPLIT: one wave per row-split item (rowsplit.hip) instead of one wave per node
template <int KIND, int LPR, int VPL, bool SPLIT>
__global__ __launch_bounds__(256) void kge_bwd_nodes_kernel(const float* __restrict__ z, int64_t ldz, int n_nodes,
                                                            Triplets tr, const float* __restrict__ rel_emb,
                                                            int64_t ldr, const float* __restrict__ gscore,
                                                            const float* __restrict__ l1,
                                                            const int32_t* __restrict__ inc_rowptr,
                                                            const int32_t* __restrict__ inc_perm,
                                                            float* __restrict__ dz, int64_t lddz, bkg_rowsplit rs,
                                                            float* __restrict__ partials, int accumulate) {
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63, grp = lane / LPR, lig = lane % LPR;
  int node, start, end, slot = -1;
  if (SPLIT) {
    int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= rs.totals[0]) return;
    item = __builtin_amdgcn_readfirstlane(item);
    node = __builtin_amdgcn_readfirstlane(rs.item_row[item]);
    start = __builtin_amdgcn_readfirstlane(rs.item_beg[item]);
    end = __builtin_amdgcn_readfirstlane(rs.item_end[item]);
    slot = __builtin_amdgcn_readfirstlane(rs.item_slot[item]);
  } else {
    node = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (node >= n_nodes) return;
    node = __builtin_amdgcn_readfirstlane(node);
    start = __builtin_amdgcn_readfirstlane(inc_rowptr[node]);
    end = __builtin_amdgcn_readfirstlane(inc_rowptr[node + 1]);
  }

  float4 self[VPL], acc[VPL];
  float inv_self = 0.f;
#pragma unroll
  for (int v = 0; v < VPL; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (KIND == KGE_TRANSE) {
    inv_self = 1.0f / fmaxf(l1[node], L1_EPS);
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      float4 a = ld4(z + (int64_t)node * ldz + 4 * (lig + v * LPR));
      self[v] = make_float4(a.x * inv_self, a.y * inv_self, a.z * inv_self, a.w * inv_self);  // z^_i
    }
  }
  // Batches of U incidences per lane group with the three dependent load stages (entry id -> triplet fields
  // -> rows) each issued for the whole batch, so a batch pays the latency chain once instead of per entry.
  // Slots past the end are clamped to a valid entry and masked out of the accumulation.
  constexpr int U = (VPL == 1) ? 4 : 2;
  for (int k0 = start; k0 < end; k0 += U * G) {
    int64_t e[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + u * G + grp;
      ok[u] = k < end;
      e[u] = inc_perm[min(k, end - 1)];
    }
    int64_t other[U], ri[U];
    float g[U], inv_o[U];
    bool is_head[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      is_head[u] = e[u] < tr.T;
      const int64_t t = is_head[u] ? e[u] : e[u] - tr.T;
      other[u] = is_head[u] ? tr.tail[t] : tr.head[t];
      ri[u] = tr.rel[t];
      g[u] = gscore[t];
    }
    float4 o[U][VPL], r[U][VPL];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      inv_o[u] = (KIND == KGE_TRANSE) ? 1.0f / fmaxf(l1[other[u]], L1_EPS) : 0.f;
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int c = 4 * (lig + v * LPR);
        o[u][v] = ld4(z + other[u] * ldz + c);
        r[u][v] = ld4(rel_emb + ri[u] * ldr + c);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        if (KIND == KGE_DISTMULT) {
          acc[v].x = fmaf(g[u] * r[u][v].x, o[u][v].x, acc[v].x); acc[v].y = fmaf(g[u] * r[u][v].y, o[u][v].y, acc[v].y);
          acc[v].z = fmaf(g[u] * r[u][v].z, o[u][v].z, acc[v].z); acc[v].w = fmaf(g[u] * r[u][v].w, o[u][v].w, acc[v].w);
        } else {
          // u = h^ + r - t^ ;  d score / d h^ = -sign(u), d score / d t^ = +sign(u)
          const float sg = is_head[u] ? -g[u] : g[u];
          const float io = inv_o[u];
          float4 d4;
          if (is_head[u]) d4 = make_float4(self[v].x + r[u][v].x - o[u][v].x * io, self[v].y + r[u][v].y - o[u][v].y * io,
                                           self[v].z + r[u][v].z - o[u][v].z * io, self[v].w + r[u][v].w - o[u][v].w * io);
          else d4 = make_float4(o[u][v].x * io + r[u][v].x - self[v].x, o[u][v].y * io + r[u][v].y - self[v].y,
                                o[u][v].z * io + r[u][v].z - self[v].z, o[u][v].w * io + r[u][v].w - self[v].w);
          acc[v].x = fmaf(sg, sgn(d4.x), acc[v].x); acc[v].y = fmaf(sg, sgn(d4.y), acc[v].y);
          acc[v].z = fmaf(sg, sgn(d4.z), acc[v].z); acc[v].w = fmaf(sg, sgn(d4.w), acc[v].w);
        }
      }
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
  if (grp != 0) return;  // after the cross-group combine every group holds the full sum
  if (SPLIT && slot >= 0) {
    constexpr int d = 4 * LPR * VPL;
#pragma unroll
    for (int v = 0; v < VPL; ++v)
      *reinterpret_cast<float4*>(partials + (int64_t)slot * d + 4 * (lig + v * LPR)) = acc[v];
    return;
  }
  kge_nodes_finish<KIND, LPR, VPL>(acc, self, inv_self, KIND == KGE_TRANSE ? (l1[node] > L1_EPS) : true, node, lig, dz,
                                   lddz, accumulate);
}

// ------------------------------------------------------------------------------------ backward: drel
template <int KIND, int LPR, int VPL>
__global__ __launch_bounds__(256) void kge_bwd_rel_kernel(const float* __restrict__ z, int64_t ldz, Triplets tr,
                                                          const float* __restrict__ rel_emb, int64_t ldr,
                                                          const float* __restrict__ gscore, int n_rel,
                                                          float* __restrict__ drel, int64_t lddr, int use_lds) {
  extern __shared__ float s_acc[];  // [n_rel][d] when use_lds
  constexpr int G = 64 / LPR, D = 4 * LPR * VPL;
  const int lane = threadIdx.x & 63, grp = lane / LPR, lig = lane % LPR;
  if (use_lds) {
    for (int i = threadIdx.x; i < n_rel * D; i += blockDim.x) s_acc[i] = 0.f;
    __syncthreads();
  }
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  for (int64_t base = wave * G; base < tr.T; base += n_waves * G) {
    const int64_t t = base + grp;
    const bool live = t < tr.T;
    const int64_t hi = live ? tr.head[t] : 0, ti = live ? tr.tail[t] : 0, ri = live ? tr.rel[t] : 0;
    const float g = live ? gscore[t] : 0.f;
    float4 h[VPL], tl[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = 4 * (lig + v * LPR);
      h[v] = ld4(z + hi * ldz + c);
      tl[v] = ld4(z + ti * ldz + c);
    }
    float ih = 0.f, it = 0.f;
    if (KIND == KGE_TRANSE) {
      float nh = 0.f, nt = 0.f;
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        nh += fabsf(h[v].x) + fabsf(h[v].y) + fabsf(h[v].z) + fabsf(h[v].w);
        nt += fabsf(tl[v].x) + fabsf(tl[v].y) + fabsf(tl[v].z) + fabsf(tl[v].w);
      }
      ih = 1.0f / fmaxf(group_sum<LPR>(nh), L1_EPS);
      it = 1.0f / fmaxf(group_sum<LPR>(nt), L1_EPS);
    }
    if (!live) continue;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = 4 * (lig + v * LPR);
      float4 q;
      if (KIND == KGE_DISTMULT) {
        q = make_float4(g * h[v].x * tl[v].x, g * h[v].y * tl[v].y, g * h[v].z * tl[v].z, g * h[v].w * tl[v].w);
      } else {
        const float4 r = ld4(rel_emb + ri * ldr + c);
        q = make_float4(-g * sgn(h[v].x * ih + r.x - tl[v].x * it), -g * sgn(h[v].y * ih + r.y - tl[v].y * it),
                        -g * sgn(h[v].z * ih + r.z - tl[v].z * it), -g * sgn(h[v].w * ih + r.w - tl[v].w * it));
      }
      float* dst = use_lds ? (s_acc + ri * D + c) : (drel + ri * lddr + c);
      atomicAdd(dst + 0, q.x); atomicAdd(dst + 1, q.y); atomicAdd(dst + 2, q.z); atomicAdd(dst + 3, q.w);
    }
  }
  if (use_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < n_rel * D; i += blockDim.x) {
      float v = s_acc[i];
      if (v != 0.f) atomicAdd(drel + (int64_t)(i / D) * lddr + (i % D), v);
    }
  }
}

// Relation-sorted variant: order[] lists triplet ids grouped by relation (rel_ptr[r] .. rel_ptr[r+1]).
// One wave owns a <= CH-triplet slice of ONE relation, accumulates it in registers and flushes once
// with 16 B/lane-shaped float atomics (T/CH flushes of d floats: far below the atomic budget).
template <int KIND, int LPR, int VPL>
__global__ __launch_bounds__(256) void kge_bwd_rel_sorted_kernel(const float* __restrict__ z, int64_t ldz, Triplets tr,
                                                                 const float* __restrict__ rel_emb, int64_t ldr,
                                                                 const float* __restrict__ gscore, int n_rel,
                                                                 float* __restrict__ drel, int64_t lddr,
                                                                 const int32_t* __restrict__ order,
                                                                 const int32_t* __restrict__ rel_ptr, int chunk) {
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63, grp = lane / LPR, lig = lane % LPR;
  int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  w = __builtin_amdgcn_readfirstlane(w);
  int r = 0, beg = 0, end = 0;
  for (; r < n_rel; ++r) {  // wave-uniform scan of <= n_rel scalars
    const int a = rel_ptr[r], b = rel_ptr[r + 1];
    const int nc = (b - a + chunk - 1) / chunk;
    if (w < nc) { beg = a + w * chunk; end = min(beg + chunk, b); break; }
    w -= nc;
  }
  if (r >= n_rel) return;
  float4 acc[VPL], rr[VPL];
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    rr[v] = ld4(rel_emb + (int64_t)r * ldr + 4 * (lig + v * LPR));
  }
  for (int k0 = beg; k0 < end; k0 += G) {
    const int k = k0 + grp;
    const bool live = k < end;
    const int64_t t = live ? order[k] : order[beg];
    const int64_t hi = tr.head[t], ti = tr.tail[t];
    const float g = live ? gscore[t] : 0.f;
    float4 h[VPL], tl[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = 4 * (lig + v * LPR);
      h[v] = ld4(z + hi * ldz + c);
      tl[v] = ld4(z + ti * ldz + c);
    }
    if (KIND == KGE_DISTMULT) {
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        acc[v].x = fmaf(g * h[v].x, tl[v].x, acc[v].x); acc[v].y = fmaf(g * h[v].y, tl[v].y, acc[v].y);
        acc[v].z = fmaf(g * h[v].z, tl[v].z, acc[v].z); acc[v].w = fmaf(g * h[v].w, tl[v].w, acc[v].w);
      }
    } else {
      float nh = 0.f, nt = 0.f;
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        nh += fabsf(h[v].x) + fabsf(h[v].y) + fabsf(h[v].z) + fabsf(h[v].w);
        nt += fabsf(tl[v].x) + fabsf(tl[v].y) + fabsf(tl[v].z) + fabsf(tl[v].w);
      }
      const float ih = 1.0f / fmaxf(group_sum<LPR>(nh), L1_EPS), it = 1.0f / fmaxf(group_sum<LPR>(nt), L1_EPS);
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        acc[v].x -= g * sgn(h[v].x * ih + rr[v].x - tl[v].x * it); acc[v].y -= g * sgn(h[v].y * ih + rr[v].y - tl[v].y * it);
        acc[v].z -= g * sgn(h[v].z * ih + rr[v].z - tl[v].z * it); acc[v].w -= g * sgn(h[v].w * ih + rr[v].w - tl[v].w * it);
      }
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
  if (grp != 0) return;
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    float* dst = drel + (int64_t)r * lddr + 4 * (lig + v * LPR);
    atomicAdd(dst + 0, acc[v].x); atomicAdd(dst + 1, acc[v].y); atomicAdd(dst + 2, acc[v].z); atomicAdd(dst + 3, acc[v].w);
  }
}

// ------------------------------------------------------------------------------------ launchers
#define BKG_KGE_DISPATCH(D, ...)                                    \
  switch (D) {                                                      \
    case 8: { constexpr int LPR = 2, VPL = 1; __VA_ARGS__; } break;        \
    case 16: { constexpr int LPR = 4, VPL = 1; __VA_ARGS__; } break;       \
    case 24: { constexpr int LPR = 2, VPL = 3; __VA_ARGS__; } break;       \
    case 48: { constexpr int LPR = 4, VPL = 3; __VA_ARGS__; } break;       \
    case 96: { constexpr int LPR = 8, VPL = 3; __VA_ARGS__; } break;       \
    case 32: { constexpr int LPR = 8, VPL = 1; __VA_ARGS__; } break;       \
    case 64: { constexpr int LPR = 16, VPL = 1; __VA_ARGS__; } break;      \
    case 128: { constexpr int LPR = 32, VPL = 1; __VA_ARGS__; } break;     \
    case 256: { constexpr int LPR = 64, VPL = 1; __VA_ARGS__; } break;     \
    case 512: { constexpr int LPR = 64, VPL = 2; __VA_ARGS__; } break;     \
    case 768: { constexpr int LPR = 64, VPL = 3; __VA_ARGS__; } break;     \
    default: bkg::set_error_msg("kge: embedding width must be one of 8,16,24,32,48,64,96,128,256,512,768"); return -1; \
  }

static bool aligned16(const void* p, int64_t ld) { return ((uintptr_t)p % 16 == 0) && (ld % 4 == 0); }

}  // namespace bkg

using namespace bkg;

extern "C" int bkg_kge_score_f32(int32_t kind, const float* z, int64_t ldz, int32_t d, const int64_t* head,
                                 const int64_t* tail, const int64_t* rel, int64_t T, const float* rel_emb,
                                 int64_t ldr, float* score, void* stream) {
  BKG_ARG(kind == KGE_DISTMULT || kind == KGE_TRANSE, "bkg_kge_score_f32: bad kind");
  BKG_ARG(aligned16(z, ldz) && aligned16(rel_emb, ldr), "bkg_kge_score_f32: rows must be 16-byte aligned");
  if (T == 0) return 0;
  Triplets tr{head, tail, rel, T};
  hipStream_t s = (hipStream_t)stream;
  BKG_KGE_DISPATCH(d, {
    constexpr int G = 64 / LPR;
    int blocks = (int)std::min<int64_t>((T + 4 * G - 1) / (4 * G), 256 * 16);
    if (kind == KGE_DISTMULT) hipLaunchKernelGGL((kge_score_kernel<KGE_DISTMULT, LPR, VPL>), dim3(blocks), dim3(256), 0, s, z, ldz, tr, rel_emb, ldr, score);
    else hipLaunchKernelGGL((kge_score_kernel<KGE_TRANSE, LPR, VPL>), dim3(blocks), dim3(256), 0, s, z, ldz, tr, rel_emb, ldr, score);
  })
  return check_launch("bkg_kge_score_f32");
}

extern "C" int bkg_row_l1norm_f32(const float* z, int64_t ldz, int64_t n, int32_t d, float* l1, void* stream) {
  BKG_ARG(aligned16(z, ldz), "bkg_row_l1norm_f32: rows must be 16-byte aligned");
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  BKG_KGE_DISPATCH(d, {
    constexpr int G = 64 / LPR;
    hipLaunchKernelGGL((row_l1_kernel<LPR, VPL>), dim3(ceil_div(n, 4 * G)), dim3(256), 0, s, z, ldz, n, l1);
  })
  return check_launch("bkg_row_l1norm_f32");
}

extern "C" int bkg_kge_bwd_nodes_f32(int32_t kind, const float* z, int64_t ldz, int64_t n_nodes, int32_t d,
                                     const int64_t* head, const int64_t* tail, const int64_t* rel, int64_t T,
                                     const float* rel_emb, int64_t ldr, const float* gscore, const float* l1,
                                     const int32_t* inc_rowptr, const int32_t* inc_perm, float* dz, int64_t lddz,
                                     const bkg_rowsplit* split, float* partials, int32_t accumulate, void* stream) {
  BKG_ARG(kind == KGE_DISTMULT || kind == KGE_TRANSE, "bkg_kge_bwd_nodes_f32: bad kind");
  BKG_ARG(kind == KGE_DISTMULT || l1 != nullptr, "bkg_kge_bwd_nodes_f32: TransE needs the row L1 norms");
  BKG_ARG(aligned16(z, ldz) && aligned16(rel_emb, ldr) && aligned16(dz, lddz), "bkg_kge_bwd_nodes_f32: alignment");
  BKG_ARG(!split || (partials && ((uintptr_t)partials % 16 == 0)), "bkg_kge_bwd_nodes_f32: splitting needs partials");
  if (n_nodes == 0) return 0;
  Triplets tr{head, tail, rel, T};
  hipStream_t s = (hipStream_t)stream;
  dim3 block(256);
  bkg_rowsplit none{};
  BKG_KGE_DISPATCH(d, {
    if (split) {
      constexpr int G = 64 / LPR;
      dim3 grid(ceil_div(split->n_items_max, 4)), fgrid(ceil_div(split->n_long_max, 4 * G));
      if (kind == KGE_DISTMULT) {
        hipLaunchKernelGGL((kge_bwd_nodes_kernel<KGE_DISTMULT, LPR, VPL, true>), grid, block, 0, s, z, ldz, (int)n_nodes, tr, rel_emb, ldr, gscore, l1, inc_rowptr, inc_perm, dz, lddz, *split, partials, accumulate);
        hipLaunchKernelGGL((kge_bwd_nodes_fixup_kernel<KGE_DISTMULT, LPR, VPL>), fgrid, block, 0, s, *split, z, ldz, l1, partials, dz, lddz, accumulate);
      } else {
        hipLaunchKernelGGL((kge_bwd_nodes_kernel<KGE_TRANSE, LPR, VPL, true>), grid, block, 0, s, z, ldz, (int)n_nodes, tr, rel_emb, ldr, gscore, l1, inc_rowptr, inc_perm, dz, lddz, *split, partials, accumulate);
        hipLaunchKernelGGL((kge_bwd_nodes_fixup_kernel<KGE_TRANSE, LPR, VPL>), fgrid, block, 0, s, *split, z, ldz, l1, partials, dz, lddz, accumulate);
      }
    } else {
      dim3 grid(ceil_div(n_nodes, 4));
      if (kind == KGE_DISTMULT) hipLaunchKernelGGL((kge_bwd_nodes_kernel<KGE_DISTMULT, LPR, VPL, false>), grid, block, 0, s, z, ldz, (int)n_nodes, tr, rel_emb, ldr, gscore, l1, inc_rowptr, inc_perm, dz, lddz, none, partials, accumulate);
      else hipLaunchKernelGGL((kge_bwd_nodes_kernel<KGE_TRANSE, LPR, VPL, false>), grid, block, 0, s, z, ldz, (int)n_nodes, tr, rel_emb, ldr, gscore, l1, inc_rowptr, inc_perm, dz, lddz, none, partials, accumulate);
    }
  })
  return check_launch("bkg_kge_bwd_nodes_f32");
}

extern "C" int bkg_kge_bwd_rel_f32(int32_t kind, const float* z, int64_t ldz, int32_t d, const int64_t* head,
                                   const int64_t* tail, const int64_t* rel, int64_t T, const float* rel_emb,
                                   int64_t ldr, const float* gscore, int32_t n_rel, float* drel, int64_t lddr,
                                   const int32_t* order, const int32_t* rel_ptr, void* stream) {
  BKG_ARG(kind == KGE_DISTMULT || kind == KGE_TRANSE, "bkg_kge_bwd_rel_f32: bad kind");
  BKG_ARG(aligned16(z, ldz) && aligned16(rel_emb, ldr), "bkg_kge_bwd_rel_f32: alignment");
  BKG_ARG((order == nullptr) == (rel_ptr == nullptr), "bkg_kge_bwd_rel_f32: order and rel_ptr come together");
  if (T == 0) return 0;
  Triplets tr{head, tail, rel, T};
  hipStream_t s = (hipStream_t)stream;
  if (order) {
    constexpr int CH = 256;
    dim3 grid(ceil_div(ceil_div(T, CH) + n_rel, 4)), block(256);
    BKG_KGE_DISPATCH(d, {
      if (kind == KGE_DISTMULT) hipLaunchKernelGGL((kge_bwd_rel_sorted_kernel<KGE_DISTMULT, LPR, VPL>), grid, block, 0, s, z, ldz, tr, rel_emb, ldr, gscore, n_rel, drel, lddr, order, rel_ptr, CH);
      else hipLaunchKernelGGL((kge_bwd_rel_sorted_kernel<KGE_TRANSE, LPR, VPL>), grid, block, 0, s, z, ldz, tr, rel_emb, ldr, gscore, n_rel, drel, lddr, order, rel_ptr, CH);
    })
    return check_launch("bkg_kge_bwd_rel_f32");
  }
  size_t lds = (size_t)n_rel * d * sizeof(float);
  int use_lds = lds <= 64 * 1024;
  BKG_KGE_DISPATCH(d, {
    constexpr int G = 64 / LPR;
    int blocks = (int)std::min<int64_t>((T + 4 * G - 1) / (4 * G), 1024);
    if (kind == KGE_DISTMULT) hipLaunchKernelGGL((kge_bwd_rel_kernel<KGE_DISTMULT, LPR, VPL>), dim3(blocks), dim3(256), use_lds ? lds : 0, s, z, ldz, tr, rel_emb, ldr, gscore, n_rel, drel, lddr, use_lds);
    else hipLaunchKernelGGL((kge_bwd_rel_kernel<KGE_TRANSE, LPR, VPL>), dim3(blocks), dim3(256), use_lds ? lds : 0, s, z, ldz, tr, rel_emb, ldr, gscore, n_rel, drel, lddr, use_lds);
  })
  return check_launch("bkg_kge_bwd_rel_f32");
}
