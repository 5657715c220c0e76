// Shared epilogue of the MFMA GEMM tiles (gemm_f32.hip, gemm_tile_bf16.h).
//
// A 32x32 accumulator has its column on the lane and 16 rows in registers, so storing it directly
// costs 16 dword stores per lane per tile, each wave-instruction touching two 128-byte row segments:
// the tile ends store-ISSUE bound (cdna_hip_programming.md T21).  Instead each wave transposes 16 rows
// x WN columns at a time through a small LDS patch and writes them as 16 B/lane row-contiguous stores
// (16 lanes cover 256 contiguous bytes of one row); bias / activation / gate / beta are applied on the
// float4.  The split-K (atomic) form keeps the direct path: one dword per lane in two 128-B row segments
// is the shape global float atomics run at full rate in (MI355X_MICROARCH.md "Global float atomics").
#pragma once
#include "common.h"

namespace bkg {

typedef float acc16_t __attribute__((ext_vector_type(16)));

struct GemmEpilogue {
  float alpha;
  float beta;         // C = ... + beta * C_old (0 -> C_old never read)
  const float* bias;  // [N] or null
  int act;            // 0 none, 1 relu, 2 elu(alpha=1)
  const float* gate;  // [M, ldg] or null: C *= (gate > 0 ? gate_scale : 0)   (ReLU+dropout backward)
  int64_t ldg;
  float gate_scale;
  int atomic;         // 1: atomicAdd into C (split-K)
};

__device__ __forceinline__ float epi_scalar(float v, const GemmEpilogue& ep, float bias, int64_t row, int col) {
  v = ep.alpha * v + bias;
  if (ep.act == 1) v = fmaxf(v, 0.f);
  else if (ep.act == 2) v = v > 0.f ? v : expm1f(v);
  if (ep.gate) v = ep.gate[row * ep.ldg + col] > 0.f ? v * ep.gate_scale : 0.f;
  return v;
}

// acc[TM][TN] of one wave (rows m_base + i*32 + ..., cols n_base + j*32 + ...) -> C.
// `patch` is this wave's LDS scratch of 16 * (TN*32 + 4) floats.
template <int TM, int TN>
__device__ __forceinline__ void store_acc_tile(const acc16_t (&acc)[TM][TN], float* __restrict__ C, int64_t ldc, int M,
                                               int N, int m_base, int n_base, const GemmEpilogue& ep,
                                               float* __restrict__ patch) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
  if (ep.atomic) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n_base + j * 32 + l31;
        if (col >= N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < M) atomicAdd(C + (int64_t)row * ldc + col, ep.alpha * acc[i][j][r]);
        }
      }
    return;
  }
  constexpr int WN = TN * 32, LDP = WN + 4;
  const bool vec = ((reinterpret_cast<uintptr_t>(C) & 15) == 0) && (ldc % 4 == 0) &&
                   (!ep.bias || (reinterpret_cast<uintptr_t>(ep.bias) & 15) == 0) &&
                   (!ep.gate || ((reinterpret_cast<uintptr_t>(ep.gate) & 15) == 0 && ep.ldg % 4 == 0));
#pragma unroll
  for (int c = 0; c < TM * 2; ++c) {  // 16-row chunks of the wave tile
    const int i = c >> 1, r0 = (c & 1) * 8;
    // registers r0..r0+7 of tile i hold chunk rows (r&3) + 8*((r>>2)&1) + 4*lh
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int pr = (r & 3) + 8 * (r >> 2) + 4 * lh;
        patch[pr * LDP + j * 32 + l31] = acc[i][j][r0 + r];
      }
    // same wave wrote and reads: LDS ops of one wave complete in order, no barrier needed
    constexpr int F4_PER_ROW = WN / 4;
#pragma unroll
    for (int q = 0; q < (16 * F4_PER_ROW) / 64; ++q) {
      const int f = q * 64 + lane;
      const int pr = f / F4_PER_ROW, c4 = (f % F4_PER_ROW) * 4;
      const int row = m_base + c * 16 + pr, col = n_base + c4;
      if (row >= M || col >= N) continue;
      float4 v = *reinterpret_cast<const float4*>(&patch[pr * LDP + c4]);
      float* dst = C + (int64_t)row * ldc + col;
      if (vec && col + 3 < N) {
        float4 b = ep.bias ? *reinterpret_cast<const float4*>(ep.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        v.x = ep.alpha * v.x + b.x; v.y = ep.alpha * v.y + b.y; v.z = ep.alpha * v.z + b.z; v.w = ep.alpha * v.w + b.w;
        if (ep.act == 1) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        } else if (ep.act == 2) {
          v.x = v.x > 0.f ? v.x : expm1f(v.x); v.y = v.y > 0.f ? v.y : expm1f(v.y);
          v.z = v.z > 0.f ? v.z : expm1f(v.z); v.w = v.w > 0.f ? v.w : expm1f(v.w);
        }
        if (ep.gate) {
          const float4 g = *reinterpret_cast<const float4*>(ep.gate + (int64_t)row * ep.ldg + col);
          v.x = g.x > 0.f ? v.x * ep.gate_scale : 0.f; v.y = g.y > 0.f ? v.y * ep.gate_scale : 0.f;
          v.z = g.z > 0.f ? v.z * ep.gate_scale : 0.f; v.w = g.w > 0.f ? v.w * ep.gate_scale : 0.f;
        }
        if (ep.beta != 0.f) {
          const float4 o = *reinterpret_cast<const float4*>(dst);
          v.x = fmaf(ep.beta, o.x, v.x); v.y = fmaf(ep.beta, o.y, v.y);
          v.z = fmaf(ep.beta, o.z, v.z); v.w = fmaf(ep.beta, o.w, v.w);
        }
        *reinterpret_cast<float4*>(dst) = v;
      } else {
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (col + e >= N) break;
          float o = epi_scalar(vv[e], ep, ep.bias ? ep.bias[col + e] : 0.f, row, col + e);
          if (ep.beta != 0.f) o = fmaf(ep.beta, dst[e], o);
          dst[e] = o;
        }
      }
    }
  }
}

}  // namespace bkg
