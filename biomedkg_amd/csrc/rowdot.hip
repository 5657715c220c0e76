// A Linear whose outputs are summed over the features -- the head of GGD, reference biomedkg/model/gcl.py:86-90:
// `mlp[-1](h).sum(1)` -- is a dot product per row:  sum_j (W h + b)_j = h . (sum_j W[j, :]) + sum_j b_j.
// Forward: one pass over h (no [rows, hidden] product, no fp32 result table, no row-sum pass).  Backward: with the
// per-row scalar g = d loss / d s, d h = g (x) wsum (one table write) and d wsum = sum_i g_i h_i (column sums with
// row weights, colsum.hip); autograd spreads d wsum over the rows of W and sum(g) over b.
#include "common.h"
#include "rowpiece.h"

namespace bkg {

// LPR lanes per row, VPL 16-byte pieces per lane; 64 / LPR rows per wave, U of them in flight per lane group
template <typename T, int LPR, int VPL>
__global__ __launch_bounds__(256) void rowdot_kernel(const T* __restrict__ x, int64_t ldx, int64_t n,
                                                     const float* __restrict__ w, float* __restrict__ out) {
  constexpr int Q = Piece<T>::Q, EPL = Piece<T>::EPL, RPW = 64 / LPR, U = 4;
  const int lane = threadIdx.x & 63, lig = lane % LPR, sub = lane / LPR;
  float4 wv[VPL * Q];
#pragma unroll
  for (int vq = 0; vq < VPL * Q; ++vq) wv[vq] = *reinterpret_cast<const float4*>(w + piece_col<T, LPR>(lig, vq));
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), waves = (int64_t)gridDim.x * 4;
  for (int64_t r0 = wave * (RPW * U); r0 < n; r0 += waves * (RPW * U)) {
    Piece<T> p[U][VPL];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = min(r0 + u * RPW + sub, n - 1);
#pragma unroll
      for (int v = 0; v < VPL; ++v) p[u][v].load(x + r * ldx + EPL * (lig + v * LPR));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float s = 0.f;
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        float4 e[Q];
        p[u][v].expand(e);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const float4 c = wv[v * Q + q];
          s = fmaf(e[q].x, c.x, s); s = fmaf(e[q].y, c.y, s); s = fmaf(e[q].z, c.z, s); s = fmaf(e[q].w, c.w, s);
        }
      }
#pragma unroll
      for (int off = LPR / 2; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
      const int64_t r = r0 + u * RPW + sub;
      if (lig == 0 && r < n) out[r] = s;
    }
  }
}

// out[i, :] = g[i] * w  (one 16-byte piece per thread)
template <typename OT>
__global__ __launch_bounds__(256) void scaled_rows_kernel(const float* __restrict__ g, const float* __restrict__ w, int64_t n,
                                                          int d, OT* __restrict__ out, int64_t ldo) {
  constexpr int EPL = 16 / sizeof(OT);
  const int ppr = d / EPL;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * ppr) return;
  const int64_t r = t / ppr;
  const int c = (int)(t % ppr) * EPL;
  const float gi = g[r];
  const float4 a = *reinterpret_cast<const float4*>(w + c);
  if constexpr (sizeof(OT) == 4) {
    *reinterpret_cast<float4*>(out + r * ldo + c) = make_float4(gi * a.x, gi * a.y, gi * a.z, gi * a.w);
  } else {
    const float4 b = *reinterpret_cast<const float4*>(w + c + 4);
    typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
    bf8 o;
    o[0] = (__bf16)(gi * a.x); o[1] = (__bf16)(gi * a.y); o[2] = (__bf16)(gi * a.z); o[3] = (__bf16)(gi * a.w);
    o[4] = (__bf16)(gi * b.x); o[5] = (__bf16)(gi * b.y); o[6] = (__bf16)(gi * b.z); o[7] = (__bf16)(gi * b.w);
    *reinterpret_cast<bf8*>(out + r * ldo + c) = o;
  }
}

}  // namespace bkg

using namespace bkg;

extern "C" int bkg_rowdot_f32(const void* x, int32_t x_is_bf16, int64_t ldx, int64_t n, int32_t d, const float* w, float* out,
                              void* stream) {
  const int epl = x_is_bf16 ? 8 : 4;
  BKG_ARG(n >= 0 && d > 0 && d % epl == 0, "bkg_rowdot_f32: d must be a multiple of the 16-byte piece");
  BKG_ARG((uintptr_t)x % 16 == 0 && ldx % epl == 0 && (uintptr_t)w % 16 == 0, "bkg_rowdot_f32: alignment");
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int pieces = d / epl;
#define BKG_ROWDOT(T_)                                                                                                  \
  BKG_PIECE_DISPATCH(pieces, {                                                                                          \
    const unsigned grid = (unsigned)std::min<int64_t>(4096, ceil_div(n, (int64_t)4 * (64 / LPR) * 4));                 \
    hipLaunchKernelGGL((rowdot_kernel<T_, LPR, VPL>), dim3(grid), dim3(256), 0, s, (const T_*)x, ldx, n, w, out);       \
  })
  if (x_is_bf16) { BKG_ROWDOT(__bf16); } else { BKG_ROWDOT(float); }
#undef BKG_ROWDOT
  return check_launch("bkg_rowdot_f32");
}

extern "C" int bkg_scaled_rows(const float* g, const float* w, int64_t n, int32_t d, void* out, int32_t out_is_bf16, int64_t ldo,
                               void* stream) {
  const int epl = out_is_bf16 ? 8 : 4;
  BKG_ARG(n >= 0 && d > 0 && d % epl == 0, "bkg_scaled_rows: d must be a multiple of the 16-byte piece");
  BKG_ARG((uintptr_t)out % 16 == 0 && ldo % epl == 0 && (uintptr_t)w % 16 == 0, "bkg_scaled_rows: alignment");
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int64_t threads = n * (d / epl);
  if (out_is_bf16)
    hipLaunchKernelGGL(scaled_rows_kernel<__bf16>, dim3(ceil_div(threads, 256)), dim3(256), 0, s, g, w, n, d, (__bf16*)out, ldo);
  else
    hipLaunchKernelGGL(scaled_rows_kernel<float>, dim3(ceil_div(threads, 256)), dim3(256), 0, s, g, w, n, d, (float*)out, ldo);
  return check_launch("bkg_scaled_rows");
}
