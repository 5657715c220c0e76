// Modality-fusion epilogues: what follows the projection GEMMs of AttentionFusion / ReDAF.
//
// AttentionFusion (/root/reference/biomedkg/utils/fusion.py:17-31): q,k,v = Linear(x) on
// x [n, M, D] -> F.scaled_dot_product_attention treats it as batch n, sequence M (= 2 modalities),
// ONE head of width D -> mean over the M outputs.  The three Linears run as one [n*M, D] x [D, 3D]
// MFMA GEMM (csrc/gemm_f32.hip); this kernel then does, per node and in one pass over the 3*M
// projected rows, the M x M dot products, the row softmax, P.V and the modality mean:
//     y = sum_b ( 1/M sum_a softmax_b(q_a.k_b / sqrt(D)) ) v_b
// ReDAF (fusion.py:49-90, sub_type_ids=None): t = relu(Linear_T(x)) comes out of the GEMM epilogue;
// this kernel applies  y = 1/M sum_m relu(dropout_0.1(t[m] * modal_w[m] * zeta))  (zeta =
// sigmoid(Linear_R(0.2)) is a [D] vector computed on the host side of the boundary).
#include "common.h"

namespace bkg {

template <int M>
__global__ __launch_bounds__(256) void attn_fuse_kernel(const float* __restrict__ qkv, int64_t ld, int n, int D,
                                                        float* __restrict__ out, int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int node = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (node >= n) return;
  const float* base = qkv + (int64_t)node * M * ld;  // rows node*M + m, columns [q | k | v]
  float s[M][M];
#pragma unroll
  for (int a = 0; a < M; ++a)
#pragma unroll
    for (int b = 0; b < M; ++b) s[a][b] = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    float4 q[M], k[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      q[m] = *reinterpret_cast<const float4*>(base + m * ld + c);
      k[m] = *reinterpret_cast<const float4*>(base + m * ld + D + c);
    }
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
      for (int b = 0; b < M; ++b)
        s[a][b] += q[a].x * k[b].x + q[a].y * k[b].y + q[a].z * k[b].z + q[a].w * k[b].w;
  }
  const float scale = 1.0f / sqrtf((float)D);
  float coef[M];
#pragma unroll
  for (int b = 0; b < M; ++b) coef[b] = 0.f;
#pragma unroll
  for (int a = 0; a < M; ++a) {
    float mx = -INFINITY;
#pragma unroll
    for (int b = 0; b < M; ++b) {
      s[a][b] = wave_sum(s[a][b]) * scale;
      mx = fmaxf(mx, s[a][b]);
    }
    float den = 0.f;
#pragma unroll
    for (int b = 0; b < M; ++b) {
      s[a][b] = expf(s[a][b] - mx);
      den += s[a][b];
    }
#pragma unroll
    for (int b = 0; b < M; ++b) coef[b] += s[a][b] / den * (1.0f / M);
  }
  for (int c = lane * 4; c < D; c += 256) {
    float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int b = 0; b < M; ++b) {
      float4 v = *reinterpret_cast<const float4*>(base + b * ld + 2 * D + c);
      y.x = fmaf(coef[b], v.x, y.x); y.y = fmaf(coef[b], v.y, y.y);
      y.z = fmaf(coef[b], v.z, y.z); y.w = fmaf(coef[b], v.w, y.w);
    }
    *reinterpret_cast<float4*>(out + (int64_t)node * ldo + c) = y;
  }
}

template <int M>
__global__ __launch_bounds__(256) void redaf_combine_kernel(const float* __restrict__ t, int64_t ld, int n, int D,
                                                            const float* __restrict__ scale,  // [M, D]
                                                            const uint8_t* __restrict__ keep, float drop_p,
                                                            uint64_t seed, uint32_t stream_id, int64_t row0,
                                                            float* __restrict__ out, int64_t ldo,
                                                            const uint32_t* __restrict__ epoch) {
  stream_id = with_epoch(stream_id, epoch);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one float4 of one node
  const int chunks = D / 4;
  if (i >= (int64_t)n * chunks) return;
  const int node = (int)(i / chunks), c = (int)(i % chunks) * 4;
  const float ks = (keep || drop_p > 0.f) ? 1.0f / (1.0f - drop_p) : 1.0f;
  const uint32_t thresh = (uint32_t)fminf(drop_p * 4294967296.0f, 4294967295.0f);
  float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int64_t row = (int64_t)node * M + m;
    float4 v = *reinterpret_cast<const float4*>(t + row * ld + c);
    const float4 sc = *reinterpret_cast<const float4*>(scale + (int64_t)m * D + c);
    v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w;
    if (keep) {
      uchar4 k = *reinterpret_cast<const uchar4*>(keep + row * D + c);
      v.x = k.x ? v.x * ks : 0.f; v.y = k.y ? v.y * ks : 0.f; v.z = k.z ? v.z * ks : 0.f; v.w = k.w ? v.w * ks : 0.f;
    } else if (drop_p > 0.f) {
      uint4 r = philox_keep4(seed, (uint64_t)(row0 + row) * D + c, stream_id);   // row0: this call's first row of the whole input
      v.x = r.x >= thresh ? v.x * ks : 0.f; v.y = r.y >= thresh ? v.y * ks : 0.f;
      v.z = r.z >= thresh ? v.z * ks : 0.f; v.w = r.w >= thresh ? v.w * ks : 0.f;
    }
    y.x += fmaxf(v.x, 0.f); y.y += fmaxf(v.y, 0.f); y.z += fmaxf(v.z, 0.f); y.w += fmaxf(v.w, 0.f);
  }
  const float inv = 1.0f / M;
  y.x *= inv; y.y *= inv; y.z *= inv; y.w *= inv;
  *reinterpret_cast<float4*>(out + (int64_t)node * ldo + c) = y;
}

}  // namespace bkg

using namespace bkg;

extern "C" int bkg_attn_fuse_f32(const float* qkv, int64_t ld, int64_t n, int32_t M, int32_t D, float* out,
                                 int64_t ldo, void* stream) {
  BKG_ARG(M >= 1 && M <= 4, "bkg_attn_fuse_f32: 1 <= modalities <= 4");
  BKG_ARG(D % 4 == 0 && ld % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)qkv % 16 == 0) && ((uintptr_t)out % 16 == 0),
          "bkg_attn_fuse_f32: D and strides must be multiples of 4 floats, 16-byte aligned");
  if (n == 0) return 0;
  dim3 grid(ceil_div(n, 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (M) {
    case 1: hipLaunchKernelGGL(attn_fuse_kernel<1>, grid, block, 0, s, qkv, ld, (int)n, D, out, ldo); break;
    case 2: hipLaunchKernelGGL(attn_fuse_kernel<2>, grid, block, 0, s, qkv, ld, (int)n, D, out, ldo); break;
    case 3: hipLaunchKernelGGL(attn_fuse_kernel<3>, grid, block, 0, s, qkv, ld, (int)n, D, out, ldo); break;
    default: hipLaunchKernelGGL(attn_fuse_kernel<4>, grid, block, 0, s, qkv, ld, (int)n, D, out, ldo); break;
  }
  return check_launch("bkg_attn_fuse_f32");
}

extern "C" int bkg_redaf_combine_f32(const float* t, int64_t ld, int64_t n, int32_t M, int32_t D, const float* scale,
                                     const uint8_t* keep, float drop_p, uint64_t seed, uint32_t stream_id, int64_t row0,
                                     float* out, int64_t ldo, void* stream) {
  BKG_ARG(M >= 1 && M <= 4, "bkg_redaf_combine_f32: 1 <= modalities <= 4");
  BKG_ARG(D % 4 == 0 && ld % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)t % 16 == 0) && ((uintptr_t)out % 16 == 0) &&
              ((uintptr_t)scale % 16 == 0), "bkg_redaf_combine_f32: alignment");
  BKG_ARG(drop_p >= 0.f && drop_p < 1.f, "bkg_redaf_combine_f32: drop_p in [0,1)");
  if (n == 0) return 0;
  dim3 grid(ceil_div(n * (D / 4), 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (M) {
    case 1: hipLaunchKernelGGL(redaf_combine_kernel<1>, grid, block, 0, s, t, ld, (int)n, D, scale, keep, drop_p, seed, stream_id, row0, out, ldo, rng_epoch_ptr()); break;
    case 2: hipLaunchKernelGGL(redaf_combine_kernel<2>, grid, block, 0, s, t, ld, (int)n, D, scale, keep, drop_p, seed, stream_id, row0, out, ldo, rng_epoch_ptr()); break;
    case 3: hipLaunchKernelGGL(redaf_combine_kernel<3>, grid, block, 0, s, t, ld, (int)n, D, scale, keep, drop_p, seed, stream_id, row0, out, ldo, rng_epoch_ptr()); break;
    default: hipLaunchKernelGGL(redaf_combine_kernel<4>, grid, block, 0, s, t, ld, (int)n, D, scale, keep, drop_p, seed, stream_id, row0, out, ldo, rng_epoch_ptr()); break;
  }
  return check_launch("bkg_redaf_combine_f32");
}
