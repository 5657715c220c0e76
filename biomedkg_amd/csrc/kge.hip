// KGE triplet scorers (DistMult / TransE): the standalone forward (decode), the row L1 norms TransE's
// backward needs, and the relation-side gradient of the UNFUSED autograd route (model.decode(...) used on
// its own, e.g. in validation / test steps).  The training step runs the fused pass (kge_fused.hip) and
// both routes share the node-side backward (kge_bwd_nodes.hip).
//
// Reference arithmetic: /root/reference/biomedkg/model/decoder.py:31-38 (TransE: L1-normalise head
// and tail with eps 1e-12, score = -||h^ + r - t^||_1) and :45-49 (DistMult: sum_k h_k r_k t_k).
// The reference materialises three [T,d] gathers per call; here each triplet's two rows are read
// once per pass straight into registers (rel_emb is a few KB and stays cache-resident).
//
// forward : LPR lanes x VPL float4 cover one d-wide row; the 64/LPR lane groups of a wave take
//           different triplets; in-group shuffle reduction -> one score store per triplet.
// drel    : relation-sorted form (one wave per <= 256-triplet slice of ONE relation, register accumulate,
//           one 1-KiB atomic flush) for large T; for small T a per-block [R,d] LDS accumulator.
#include "common.h"
#include "rowpiece.h"

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
__device__ __forceinline__ float abs4(const float4& a) { return fabsf(a.x) + fabsf(a.y) + fabsf(a.z) + fabsf(a.w); }

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
      for (int v = 0; v < VPL; ++v) { nh += abs4(h[v]); nt += abs4(tl[v]); }
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
    for (int v = 0; v < VPL; ++v) s += abs4(ld4(z + row * ldz + 4 * (lig + v * LPR)));
  }
  s = group_sum<LPR>(s);
  if (row < n && lig == 0) l1[row] = s;
}

// per-triplet relation-side contribution q (one float4 chunk): DistMult g h (*) t, TransE -g sign(u)
template <int KIND>
__device__ __forceinline__ float4 rel_term(float g, const float4& h, const float4& t, const float4& r, float ih, float it) {
  if (KIND == KGE_DISTMULT) return make_float4(g * h.x * t.x, g * h.y * t.y, g * h.z * t.z, g * h.w * t.w);
  return make_float4(-g * sgn(h.x * ih + r.x - t.x * it), -g * sgn(h.y * ih + r.y - t.y * it),
                     -g * sgn(h.z * ih + r.z - t.z * it), -g * sgn(h.w * ih + r.w - t.w * it));
}

// ------------------------------------------------------------------------------------ backward: drel (small T)
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
      for (int v = 0; v < VPL; ++v) { nh += abs4(h[v]); nt += abs4(tl[v]); }
      ih = 1.0f / fmaxf(group_sum<LPR>(nh), L1_EPS);
      it = 1.0f / fmaxf(group_sum<LPR>(nt), L1_EPS);
    }
    if (!live) continue;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = 4 * (lig + v * LPR);
      const float4 r = (KIND == KGE_TRANSE) ? ld4(rel_emb + ri * ldr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 q = rel_term<KIND>(g, h[v], tl[v], r, ih, it);
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

// ------------------------------------------------------------------------------------ backward: drel (sorted)
// order[] lists triplet ids grouped by relation (rel_ptr[r] .. rel_ptr[r+1]).  One wave owns a <= chunk-triplet
// slice of ONE relation, accumulates it in registers and flushes once with float atomics.
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
  zero4(acc);
#pragma unroll
  for (int v = 0; v < VPL; ++v) rr[v] = ld4(rel_emb + (int64_t)r * ldr + 4 * (lig + v * LPR));
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
    float ih = 0.f, it = 0.f;
    if (KIND == KGE_TRANSE) {
      float nh = 0.f, nt = 0.f;
#pragma unroll
      for (int v = 0; v < VPL; ++v) { nh += abs4(h[v]); nt += abs4(tl[v]); }
      ih = 1.0f / fmaxf(group_sum<LPR>(nh), L1_EPS);
      it = 1.0f / fmaxf(group_sum<LPR>(nt), L1_EPS);
    }
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const float4 q = rel_term<KIND>(g, h[v], tl[v], rr[v], ih, it);
      acc[v].x += q.x; acc[v].y += q.y; acc[v].z += q.z; acc[v].w += q.w;
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

static bool aligned16(const void* p, int64_t ld) { return ((uintptr_t)p % 16 == 0) && (ld % 4 == 0); }

}  // namespace bkg

using namespace bkg;

extern "C" int bkg_kge_score_f32(int32_t kind, const float* z, int64_t ldz, int32_t d, const int64_t* head,
                                 const int64_t* tail, const int64_t* rel, int64_t T, const float* rel_emb,
                                 int64_t ldr, float* score, void* stream) {
  BKG_ARG(kind == KGE_DISTMULT || kind == KGE_TRANSE, "bkg_kge_score_f32: bad kind");
  BKG_ARG(aligned16(z, ldz) && aligned16(rel_emb, ldr) && d % 4 == 0, "bkg_kge_score_f32: rows must be 16-byte aligned");
  if (T == 0) return 0;
  Triplets tr{head, tail, rel, T};
  hipStream_t s = (hipStream_t)stream;
  BKG_PIECE_DISPATCH(d / 4, {
    constexpr int G = 64 / LPR;
    int blocks = (int)std::min<int64_t>((T + 4 * G - 1) / (4 * G), 256 * 16);
    if (kind == KGE_DISTMULT) hipLaunchKernelGGL((kge_score_kernel<KGE_DISTMULT, LPR, VPL>), dim3(blocks), dim3(256), 0, s, z, ldz, tr, rel_emb, ldr, score);
    else hipLaunchKernelGGL((kge_score_kernel<KGE_TRANSE, LPR, VPL>), dim3(blocks), dim3(256), 0, s, z, ldz, tr, rel_emb, ldr, score);
  })
  return check_launch("bkg_kge_score_f32");
}

extern "C" int bkg_row_l1norm_f32(const float* z, int64_t ldz, int64_t n, int32_t d, float* l1, void* stream) {
  BKG_ARG(aligned16(z, ldz) && d % 4 == 0, "bkg_row_l1norm_f32: rows must be 16-byte aligned");
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  BKG_PIECE_DISPATCH(d / 4, {
    constexpr int G = 64 / LPR;
    hipLaunchKernelGGL((row_l1_kernel<LPR, VPL>), dim3(ceil_div(n, 4 * G)), dim3(256), 0, s, z, ldz, n, l1);
  })
  return check_launch("bkg_row_l1norm_f32");
}

extern "C" int bkg_kge_bwd_rel_f32(int32_t kind, const float* z, int64_t ldz, int32_t d, const int64_t* head,
                                   const int64_t* tail, const int64_t* rel, int64_t T, const float* rel_emb,
                                   int64_t ldr, const float* gscore, int32_t n_rel, float* drel, int64_t lddr,
                                   const int32_t* order, const int32_t* rel_ptr, void* stream) {
  BKG_ARG(kind == KGE_DISTMULT || kind == KGE_TRANSE, "bkg_kge_bwd_rel_f32: bad kind");
  BKG_ARG(aligned16(z, ldz) && aligned16(rel_emb, ldr) && d % 4 == 0, "bkg_kge_bwd_rel_f32: alignment");
  BKG_ARG((order == nullptr) == (rel_ptr == nullptr), "bkg_kge_bwd_rel_f32: order and rel_ptr come together");
  if (T == 0) return 0;
  Triplets tr{head, tail, rel, T};
  hipStream_t s = (hipStream_t)stream;
  if (order) {
    constexpr int CH = 256;
    dim3 grid(ceil_div(ceil_div(T, CH) + n_rel, 4)), block(256);
    BKG_PIECE_DISPATCH(d / 4, {
      if (kind == KGE_DISTMULT) hipLaunchKernelGGL((kge_bwd_rel_sorted_kernel<KGE_DISTMULT, LPR, VPL>), grid, block, 0, s, z, ldz, tr, rel_emb, ldr, gscore, n_rel, drel, lddr, order, rel_ptr, CH);
      else hipLaunchKernelGGL((kge_bwd_rel_sorted_kernel<KGE_TRANSE, LPR, VPL>), grid, block, 0, s, z, ldz, tr, rel_emb, ldr, gscore, n_rel, drel, lddr, order, rel_ptr, CH);
    })
    return check_launch("bkg_kge_bwd_rel_f32");
  }
  size_t lds = (size_t)n_rel * d * sizeof(float);
  int use_lds = lds <= 64 * 1024;
  BKG_PIECE_DISPATCH(d / 4, {
    constexpr int G = 64 / LPR;
    int blocks = (int)std::min<int64_t>((T + 4 * G - 1) / (4 * G), 1024);
    if (kind == KGE_DISTMULT) hipLaunchKernelGGL((kge_bwd_rel_kernel<KGE_DISTMULT, LPR, VPL>), dim3(blocks), dim3(256), use_lds ? lds : 0, s, z, ldz, tr, rel_emb, ldr, gscore, n_rel, drel, lddr, use_lds);
    else hipLaunchKernelGGL((kge_bwd_rel_kernel<KGE_TRANSE, LPR, VPL>), dim3(blocks), dim3(256), use_lds ? lds : 0, s, z, ldz, tr, rel_emb, ldr, gscore, n_rel, drel, lddr, use_lds);
  })
  return check_launch("bkg_kge_bwd_rel_f32");
}
