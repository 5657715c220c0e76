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
      uint4 r = hash_keep4(seed, (uint64_t)(row0 + row) * D + c, stream_id);   // row0: this call's first row of the whole input
      v.x = r.x >= thresh ? v.x * ks : 0.f; v.y = r.y >= thresh ? v.y * ks : 0.f;
      v.z = r.z >= thresh ? v.z * ks : 0.f; v.w = r.w >= thresh ? v.w * ks : 0.f;
    }
    y.x += fmaxf(v.x, 0.f); y.y += fmaxf(v.y, 0.f); y.z += fmaxf(v.z, 0.f); y.w += fmaxf(v.w, 0.f);
  }
  const float inv = 1.0f / M;
  y.x *= inv; y.y *= inv; y.z *= inv; y.w *= inv;
  *reinterpret_cast<float4*>(out + (int64_t)node * ldo + c) = y;
}


// ---- AttentionFusion in projected form (bf16 mode, two modalities; biomedkg_amd/ops.py attention_fusion):
//   s_ab = q_a . k_b = x_a^T (Wq^T Wk) x_b + (Wq^T bk) . x_a + (Wk^T bq) . x_b + bq . bk          (the fuser is frozen:
//   y    = sum_b c_b v_b = Wv (sum_b c_b x_b) + bv,   c_b = 1/2 sum_a softmax_b(s_ab / sqrt(D)), sum_b c_b = 1   Wq^T Wk is a
// constant of the run) -- so the three [2n, D] x [D, D] projections of reference fusion.py:22-26 become ONE [2n, D] product
// T = X (Wq^T Wk) and one [n, D] product with Wv over the COMBINED rows.  This kernel sits between them: per node the four
// scores from T and X (two rows each, bf16), the 2 x 2 softmax, the modality mean's coefficients, and the combined row
// xc = c_0 x_0 + c_1 x_1 in bf16.  One wave per node, 16-byte pieces, wave-shuffle reductions.
__device__ __forceinline__ void bf16x8_to_f32(const uint4 v, float (&f)[8]) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

// Eight partial sums per lane -> the eight wave totals in every lane: a reduce-scatter over the lane bits 5, 4, 3 (each step
// sends the half of the values the partner keeps: 4 + 2 + 1 shuffles), three more steps over bits 2, 1, 0 on the one value
// left, then eight readlanes -- 10 shuffles where eight separate wave sums take 48.  Value k ends in lanes with
// (lane >> 3) == k.
__device__ __forceinline__ void attn_reduce8(float (&v)[8], float (&tot)[8]) {
  const int lane = threadIdx.x & 63;
  const bool hi5 = (lane & 32) != 0, hi4 = (lane & 16) != 0, hi3 = (lane & 8) != 0;
  float a[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {      // keep values [0, 4) in the low half of the wave, [4, 8) in the high half
    const float send = hi5 ? v[i] : v[4 + i];
    a[i] = (hi5 ? v[4 + i] : v[i]) + __shfl_xor(send, 32, 64);
  }
  float b[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float send = hi4 ? a[i] : a[2 + i];
    b[i] = (hi4 ? a[2 + i] : a[i]) + __shfl_xor(send, 16, 64);
  }
  float c = (hi3 ? b[1] : b[0]) + __shfl_xor(hi3 ? b[0] : b[1], 8, 64);
  c += __shfl_xor(c, 4, 64);
  c += __shfl_xor(c, 2, 64);
  c += __shfl_xor(c, 1, 64);
  // lane l now holds the total of value 4 * bit5 + 2 * bit4 + bit3 = l >> 3
#pragma unroll
  for (int k = 0; k < 8; ++k) tot[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 8 * k));
}

template <int PASSES>      // D = PASSES * 512 at most (64 lanes x 8 elements per pass)
__global__ __launch_bounds__(256) void attn_scores_combine_kernel(const __bf16* __restrict__ t, int64_t ldt,
                                                                  const __bf16* __restrict__ x, int64_t ldx, int n, int D,
                                                                  const float* __restrict__ u /* Wk^T bq */,
                                                                  const float* __restrict__ w /* Wq^T bk */, float c0,
                                                                  __bf16* __restrict__ out, int64_t ldo) {
  // A wave walks nodes wave, wave + stride, ...: the two bias vectors stay in its registers (read per node they were as
  // many L1 bytes as the node's four rows), and the eight reductions of a node are one joint butterfly (attn_reduce8).
  const int lane = threadIdx.x & 63;
  float fu[PASSES][8], fw[PASSES][8];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int c = (p * 64 + lane) * 8;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int cc = c < D ? c : 0;
      float4 a = *reinterpret_cast<const float4*>(u + cc + 4 * h), b = *reinterpret_cast<const float4*>(w + cc + 4 * h);
      if (c >= D) a = b = make_float4(0.f, 0.f, 0.f, 0.f);
      fu[p][4 * h] = a.x; fu[p][4 * h + 1] = a.y; fu[p][4 * h + 2] = a.z; fu[p][4 * h + 3] = a.w;
      fw[p][4 * h] = b.x; fw[p][4 * h + 1] = b.y; fw[p][4 * h + 2] = b.z; fw[p][4 * h + 3] = b.w;
    }
  }
  const float scale = 1.0f / sqrtf((float)D);
  const int stride = gridDim.x * 4;
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  auto fetch = [&](int node_, uint4 (&xr_)[PASSES][2], uint4 (&tr_)[PASSES][2]) {
    const int nd = min(node_, n - 1);                 // (past the end: a valid node, never used)
    const __bf16* t0 = t + (int64_t)(2 * nd) * ldt;
    const __bf16* x0 = x + (int64_t)(2 * nd) * ldx;
    // (every load unconditional -- lanes past D read column 0 and drop it: a load under a lane condition becomes a branch
    // of its own with its own wait, and the eight loads of a node went out one at a time)
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int c = (p * 64 + lane) * 8;
      const bool ok = c < D;
      const int cc = ok ? c : 0;
      const uint4 a0 = *reinterpret_cast<const uint4*>(x0 + cc), a1 = *reinterpret_cast<const uint4*>(x0 + ldx + cc);
      const uint4 b0 = *reinterpret_cast<const uint4*>(t0 + cc), b1 = *reinterpret_cast<const uint4*>(t0 + ldt + cc);
      xr_[p][0] = ok ? a0 : z; xr_[p][1] = ok ? a1 : z;
      tr_[p][0] = ok ? b0 : z; tr_[p][1] = ok ? b1 : z;
    }
  };
  uint4 xn[PASSES][2], tn[PASSES][2];
  int node = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (node < n) fetch(node, xn, tn);
  for (; node < n; node += stride) {
    uint4 xr[PASSES][2], tr[PASSES][2];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) { xr[p][0] = xn[p][0]; xr[p][1] = xn[p][1]; tr[p][0] = tn[p][0]; tr[p][1] = tn[p][1]; }
    fetch(node + stride, xn, tn);                     // the next node's rows travel under this node's arithmetic
    // v: s[0][0], s[0][1], s[1][0], s[1][1], lin_u[0], lin_u[1], lin_w[0], lin_w[1]
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      float fx[2][8], ft[2][8];
      bf16x8_to_f32(xr[p][0], fx[0]); bf16x8_to_f32(xr[p][1], fx[1]);
      bf16x8_to_f32(tr[p][0], ft[0]); bf16x8_to_f32(tr[p][1], ft[1]);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
          for (int b = 0; b < 2; ++b) v[2 * a + b] = fmaf(ft[a][e], fx[b][e], v[2 * a + b]);
          v[4 + a] = fmaf(fu[p][e], fx[a][e], v[4 + a]);
          v[6 + a] = fmaf(fw[p][e], fx[a][e], v[6 + a]);
        }
      }
    }
    float tot[8];
    attn_reduce8(v, tot);
    float coef[2] = {0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      float sc[2];
#pragma unroll
      for (int b = 0; b < 2; ++b) sc[b] = (tot[2 * a + b] + tot[6 + a] + tot[4 + b] + c0) * scale;
      const float mx = fmaxf(sc[0], sc[1]);
      const float e0 = expf(sc[0] - mx), e1 = expf(sc[1] - mx);
      const float inv = 0.5f / (e0 + e1);
      coef[0] += e0 * inv;
      coef[1] += e1 * inv;
    }
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int c = (p * 64 + lane) * 8;
      if (c >= D) continue;
      float f0[8], f1[8];
      bf16x8_to_f32(xr[p][0], f0); bf16x8_to_f32(xr[p][1], f1);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (__bf16)fmaf(coef[1], f1[e], coef[0] * f0[e]);
      *reinterpret_cast<bf16x8*>(out + (int64_t)node * ldo + c) = o;
    }
  }
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

// The node-wise middle of the projected AttentionFusion (see attn_scores_combine_kernel): t = X (Wq^T Wk) and x as
// [2 n, D] bf16 tables (rows 2 i, 2 i + 1 = the two modality tokens of node i), u = Wk^T bq, w = Wq^T bk, c0 = bq . bk ->
// out[i] = c_0 x_{i,0} + c_1 x_{i,1} (bf16), the rows the V projection then runs over.
extern "C" int bkg_attn_scores_combine_bf16(const void* t, int64_t ldt, const void* x, int64_t ldx, int64_t n, int32_t D,
                                            const float* u, const float* w, float c0, void* out, int64_t ldo, void* stream) {
  BKG_ARG(D % 8 == 0 && D <= 2048 && ldt % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)t % 16 == 0) &&
              ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)u % 16 == 0) && ((uintptr_t)w % 16 == 0),
          "bkg_attn_scores_combine_bf16: D a multiple of 8 up to 2048, 16-byte aligned rows");
  if (n == 0) return 0;
  // waves walk the nodes, six blocks per CU (swept at 130k nodes: 768 blocks 200 us, 1536 185, one node per wave 188)
  dim3 grid((unsigned)std::min<int64_t>(ceil_div(n, 4), 256 * 6)), block(256);
  hipStream_t s = (hipStream_t)stream;
#define BKG_ASC(P) hipLaunchKernelGGL(attn_scores_combine_kernel<P>, grid, block, 0, s, (const __bf16*)t, ldt, (const __bf16*)x, \
                                      ldx, (int)n, D, u, w, c0, (__bf16*)out, ldo)
  if (D <= 512) BKG_ASC(1); else if (D <= 1024) BKG_ASC(2); else if (D <= 1536) BKG_ASC(3); else BKG_ASC(4);
#undef BKG_ASC
  return check_launch("bkg_attn_scores_combine_bf16");
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
