// Fused KGE training pass: triplet scores + BCE-with-logits terms + d loss / d score + d rel_emb in ONE
// sweep over the triplets' head / tail rows.
//
// Reference: KGEModule.training_step (/root/reference/biomedkg/kge_module.py:121-128) runs
// decode(pos), decode(neg), cat, ones/zeros, binary_cross_entropy_with_logits -- and autograd then walks
// the [T,d] gathers again.  Unfused, this build needs three passes over the 2T rows (score, node-side
// gradient, relation-side gradient); because BCE's gradient g = sigmoid(s) - y only needs the score and
// the label (triplet id < n_pos), the relation-side gradient can be accumulated in the scoring pass
// itself, leaving one more pass for the node side (kge.hip kge_bwd_nodes_kernel, fed with the g written
// here).  Triplets are visited in relation-sorted order so that a wave owns a slice of ONE relation and
// keeps its d rel_emb partial in registers (one 1-KiB atomic flush per <= 256 triplets).
//
// One launch per label (positives y = 1, negatives y = 0).  Outputs: score[T] (original triplet order),
// g[T] = (sigmoid(s) - y) * inv_total, loss_part[n_waves] (sum it, multiply by inv_total),
// drel[R,d] (zeroed by the caller).  The caller multiplies by the upstream gradient of the loss.
#include "common.h"

namespace bkg {

enum { KF_DISTMULT = 0, KF_TRANSE = 1 };
constexpr float KF_EPS = 1e-12f;

__device__ __forceinline__ float kf_sgn(float v) { return (v > 0.f) - (v < 0.f); }
__device__ __forceinline__ float4 kf_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int LPR>
__device__ __forceinline__ float kf_group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int KIND, int LPR, int VPL>
__global__ __launch_bounds__(256) void kge_fused_bce_kernel(const float* __restrict__ z, int64_t ldz,
                                                            const int64_t* __restrict__ head,
                                                            const int64_t* __restrict__ tail, int64_t T, float label, float inv_t,
                                                            const float* __restrict__ rel_emb, int64_t ldr, int n_rel,
                                                            const int32_t* __restrict__ order,
                                                            const int32_t* __restrict__ rel_ptr, int chunk,
                                                            float* __restrict__ score, float* __restrict__ g_out,
                                                            float* __restrict__ loss_part, float* __restrict__ drel,
                                                            int64_t lddr) {
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63, grp = lane / LPR, lig = lane % LPR;
  const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
  int w = __builtin_amdgcn_readfirstlane(wave_id);
  int r = 0, beg = 0, end = 0;
  for (; r < n_rel; ++r) {
    const int a = rel_ptr[r], b = rel_ptr[r + 1];
    const int nc = (b - a + chunk - 1) / chunk;
    if (w < nc) { beg = a + w * chunk; end = min(beg + chunk, b); break; }
    w -= nc;
  }
  float loss_acc = 0.f;
  if (r < n_rel) {
    float4 acc[VPL], rr[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
      rr[v] = kf_ld4(rel_emb + (int64_t)r * ldr + 4 * (lig + v * LPR));
    }
    // batches of U triplets per lane group: triplet ids, then endpoints, then rows are each loaded for the
    // whole batch so the three-deep dependent-load chain is paid once per batch, not once per triplet
    constexpr int U = (VPL == 1) ? 4 : 2;
    for (int k0 = beg; k0 < end; k0 += U * G) {
      int64_t tt[U], hh[U], ta[U];
      bool lv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 + u * G + grp;
        lv[u] = k < end;
        tt[u] = order[min(k, end - 1)];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { hh[u] = head[tt[u]]; ta[u] = tail[tt[u]]; }
      float4 hb[U][VPL], tb[U][VPL];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const int c = 4 * (lig + v * LPR);
          hb[u][v] = kf_ld4(z + hh[u] * ldz + c);
          tb[u][v] = kf_ld4(z + ta[u] * ldz + c);
        }
#pragma unroll
      for (int u = 0; u < U; ++u) {
      const bool live = lv[u];
      const int64_t t = tt[u];
      const float4* h = hb[u];
      const float4* tl = tb[u];
      float s = 0.f, ih = 0.f, it = 0.f;
      if (KIND == KF_DISTMULT) {
#pragma unroll
        for (int v = 0; v < VPL; ++v)
          s += h[v].x * rr[v].x * tl[v].x + h[v].y * rr[v].y * tl[v].y + h[v].z * rr[v].z * tl[v].z + h[v].w * rr[v].w * tl[v].w;
        s = kf_group_sum<LPR>(s);
      } else {
        float nh = 0.f, nt = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          nh += fabsf(h[v].x) + fabsf(h[v].y) + fabsf(h[v].z) + fabsf(h[v].w);
          nt += fabsf(tl[v].x) + fabsf(tl[v].y) + fabsf(tl[v].z) + fabsf(tl[v].w);
        }
        ih = 1.0f / fmaxf(kf_group_sum<LPR>(nh), KF_EPS);
        it = 1.0f / fmaxf(kf_group_sum<LPR>(nt), KF_EPS);
#pragma unroll
        for (int v = 0; v < VPL; ++v)
          s += fabsf(h[v].x * ih + rr[v].x - tl[v].x * it) + fabsf(h[v].y * ih + rr[v].y - tl[v].y * it) +
               fabsf(h[v].z * ih + rr[v].z - tl[v].z * it) + fabsf(h[v].w * ih + rr[v].w - tl[v].w * it);
        s = -kf_group_sum<LPR>(s);
      }
      // BCE with logits (torch formula): max(s,0) - s*y + log1p(exp(-|s|));  d/ds = sigmoid(s) - y
      const float y = label;
      const float e = expf(-fabsf(s));
      const float sig = s >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
      const float g = live ? (sig - y) * inv_t : 0.f;
      if (live && lig == 0) {
        score[t] = s;
        g_out[t] = g;
        loss_acc += fmaxf(s, 0.f) - s * y + log1pf(e);
      }
      if (KIND == KF_DISTMULT) {
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          acc[v].x = fmaf(g * h[v].x, tl[v].x, acc[v].x); acc[v].y = fmaf(g * h[v].y, tl[v].y, acc[v].y);
          acc[v].z = fmaf(g * h[v].z, tl[v].z, acc[v].z); acc[v].w = fmaf(g * h[v].w, tl[v].w, acc[v].w);
        }
      } else {
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          acc[v].x -= g * kf_sgn(h[v].x * ih + rr[v].x - tl[v].x * it); acc[v].y -= g * kf_sgn(h[v].y * ih + rr[v].y - tl[v].y * it);
          acc[v].z -= g * kf_sgn(h[v].z * ih + rr[v].z - tl[v].z * it); acc[v].w -= g * kf_sgn(h[v].w * ih + rr[v].w - tl[v].w * it);
        }
      }
      }  // u
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
    if (grp == 0) {
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        float* dst = drel + (int64_t)r * lddr + 4 * (lig + v * LPR);
        atomicAdd(dst + 0, acc[v].x); atomicAdd(dst + 1, acc[v].y); atomicAdd(dst + 2, acc[v].z); atomicAdd(dst + 3, acc[v].w);
      }
    }
  }
  // per-wave loss partial (fixed-order sum by the caller): only lig == 0 lanes hold terms
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) loss_acc += __shfl_xor(loss_acc, o, 64);
  if (lane == 0) loss_part[wave_id] = loss_acc;
}

static bool kf_aligned16(const void* p, int64_t ld) { return ((uintptr_t)p % 16 == 0) && (ld % 4 == 0); }

}  // namespace bkg

using namespace bkg;

extern "C" int64_t bkg_kge_fused_bce_waves(int64_t T, int32_t n_rel) {
  return ((T + 255) / 256 + n_rel + 3) / 4 * 4;
}

extern "C" int bkg_kge_fused_bce_f32(int32_t kind, const float* z, int64_t ldz, int32_t d, const int64_t* head,
                                     const int64_t* tail, int64_t T, float label, float inv_total, const float* rel_emb,
                                     int64_t ldr,
                                     int32_t n_rel, const int32_t* order, const int32_t* rel_ptr, float* score,
                                     float* g, float* loss_part, float* drel, int64_t lddr, void* stream) {
  BKG_ARG(kind == KF_DISTMULT || kind == KF_TRANSE, "bkg_kge_fused_bce_f32: bad kind");
  BKG_ARG(kf_aligned16(z, ldz) && kf_aligned16(rel_emb, ldr) && kf_aligned16(drel, lddr), "bkg_kge_fused_bce_f32: alignment");
  BKG_ARG(T > 0 && n_rel > 0 && (label == 0.f || label == 1.f), "bkg_kge_fused_bce_f32: bad sizes / label");
  hipStream_t s = (hipStream_t)stream;
  constexpr int CH = 256;
  dim3 grid((unsigned)(bkg_kge_fused_bce_waves(T, n_rel) / 4)), block(256);
#define KF_LAUNCH(LPR_, VPL_)                                                                                          \
  if (kind == KF_DISTMULT) hipLaunchKernelGGL((kge_fused_bce_kernel<KF_DISTMULT, LPR_, VPL_>), grid, block, 0, s, z, ldz, head, tail, T, label, inv_total, rel_emb, ldr, n_rel, order, rel_ptr, CH, score, g, loss_part, drel, lddr); \
  else hipLaunchKernelGGL((kge_fused_bce_kernel<KF_TRANSE, LPR_, VPL_>), grid, block, 0, s, z, ldz, head, tail, T, label, inv_total, rel_emb, ldr, n_rel, order, rel_ptr, CH, score, g, loss_part, drel, lddr)
  switch (d) {
    case 8: KF_LAUNCH(2, 1); break;
    case 16: KF_LAUNCH(4, 1); break;
    case 24: KF_LAUNCH(2, 3); break;
    case 32: KF_LAUNCH(8, 1); break;
    case 48: KF_LAUNCH(4, 3); break;
    case 64: KF_LAUNCH(16, 1); break;
    case 96: KF_LAUNCH(8, 3); break;
    case 128: KF_LAUNCH(32, 1); break;
    case 256: KF_LAUNCH(64, 1); break;
    case 512: KF_LAUNCH(64, 2); break;
    case 768: KF_LAUNCH(64, 3); break;
    default: set_error_msg("bkg_kge_fused_bce_f32: embedding width must be one of 8,16,24,32,48,64,96,128,256,512,768"); return -1;
  }
#undef KF_LAUNCH
  return check_launch("bkg_kge_fused_bce_f32");
}
