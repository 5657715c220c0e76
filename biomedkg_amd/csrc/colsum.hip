// Column sums of a tall activation / gradient table: the bias gradients `dOut.sum(0)` of every Linear / GCNConv /
// RGCNConv backward (autograd of the `+ bias` in reference biomedkg/model/encoder.py:50-59,153-162, gcl.py:86-90).
// One streaming pass at HBM rate over an fp32 or bf16 table (16 B per lane), per-block partial rows, then a second
// tiny launch adds the partial rows in block order (plain stores: bitwise repeatable, no atomics).
#include "common.h"
#include "rowpiece.h"

namespace bkg {

#ifndef CS_MAX_BLOCKS_
#define CS_MAX_BLOCKS_ 512
#endif
constexpr int CS_MAX_BLOCKS = CS_MAX_BLOCKS_;

// block = 256 threads = RPB row slots x PPR pieces per row; thread (slot, piece) walks rows slot, slot + RPB, ...
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, int64_t ldx, int64_t n, int d,
                                                             int64_t rows_per_block, const float* __restrict__ row_w,
                                                             float* __restrict__ partial) {
  constexpr int Q = Piece<T>::Q, EPL = Piece<T>::EPL;
  __shared__ float4 red[256 * Q];
  const int ppr = d / EPL, rpb = 256 / ppr;             // pieces per row (<= 256), row slots per block
  const int piece = threadIdx.x % ppr, slot = threadIdx.x / ppr;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, n);
  float4 acc[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (slot < rpb) {
    constexpr int U = 8;   // rows in flight per thread
    int64_t r = r0 + slot;
    for (; r + (U - 1) * rpb < r1; r += U * rpb) {
      Piece<T> p[U];
      float w[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        p[u].load(x + (r + (int64_t)u * rpb) * ldx + piece * EPL);
        w[u] = row_w ? row_w[r + (int64_t)u * rpb] : 1.0f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) p[u].fma(acc, w[u]);
    }
    for (; r < r1; r += rpb) {
      Piece<T> p;
      p.load(x + r * ldx + piece * EPL);
      p.fma(acc, row_w ? row_w[r] : 1.0f);
    }
  }
  // add the row slots in slot order
#pragma unroll
  for (int q = 0; q < Q; ++q) red[threadIdx.x * Q + q] = acc[q];
  __syncthreads();
  if (slot == 0) {
    for (int s = 1; s < rpb; ++s)
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const float4 v = red[(s * ppr + piece) * Q + q];
        acc[q].x += v.x; acc[q].y += v.y; acc[q].z += v.z; acc[q].w += v.w;
      }
#pragma unroll
    for (int q = 0; q < Q; ++q)
      *reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * d + piece * EPL + 4 * q) = acc[q];
  }
}

// The gated bf16 table of a layer's output gradient (spmm.hip cast_bf16_kernel: bf16(gate > 0 ? s * x : 0)) AND its
// column sums (the bias gradient) in one pass: same blocking and addition order as colsum_partial_kernel<__bf16> over the
// finished table, so the sums are bit-identical to the two-pass form -- minus one read of the table.
template <typename T> struct Load8;
template <> struct Load8<float> {
  static __device__ __forceinline__ void get(const float* p, float (&a)[8]) {
    const float4 u = *reinterpret_cast<const float4*>(p), v = *reinterpret_cast<const float4*>(p + 4);
    a[0] = u.x; a[1] = u.y; a[2] = u.z; a[3] = u.w; a[4] = v.x; a[5] = v.y; a[6] = v.z; a[7] = v.w;
  }
};
template <> struct Load8<__bf16> {
  static __device__ __forceinline__ void get(const __bf16* p, float (&a)[8]) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    a[0] = __uint_as_float(v.x << 16); a[1] = __uint_as_float(v.x & 0xffff0000u);
    a[2] = __uint_as_float(v.y << 16); a[3] = __uint_as_float(v.y & 0xffff0000u);
    a[4] = __uint_as_float(v.z << 16); a[5] = __uint_as_float(v.z & 0xffff0000u);
    a[6] = __uint_as_float(v.w << 16); a[7] = __uint_as_float(v.w & 0xffff0000u);
  }
};

template <typename ST, typename GT>
__global__ __launch_bounds__(256) void gate_cast_colsum_kernel(const ST* __restrict__ src, const GT* __restrict__ gate,
                                                               float scale, __bf16* __restrict__ dst, int64_t n, int d,
                                                               int64_t rows_per_block, float* __restrict__ partial) {
  __shared__ float4 red[256 * 2];
  const int ppr = d / 8, rpb = 256 / ppr;
  const int piece = threadIdx.x % ppr, slot = threadIdx.x / ppr;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, n);
  float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  if (slot < rpb) {
    constexpr int U = 4;
    auto one = [&](const float (&x)[8], const float (&g)[8], int64_t r) {
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
      bf16x8 o;
      float back[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = gate ? (g[e] > 0.f ? x[e] * scale : 0.f) : x[e];
        o[e] = (__bf16)v;
        back[e] = (float)o[e];
      }
      *reinterpret_cast<bf16x8*>(dst + r * d + piece * 8) = o;
      acc[0].x += back[0]; acc[0].y += back[1]; acc[0].z += back[2]; acc[0].w += back[3];
      acc[1].x += back[4]; acc[1].y += back[5]; acc[1].z += back[6]; acc[1].w += back[7];
    };
    int64_t r = r0 + slot;
    for (; r + (U - 1) * rpb < r1; r += U * rpb) {
      float x[U][8], g[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        Load8<ST>::get(src + (r + (int64_t)u * rpb) * d + piece * 8, x[u]);
        if (gate) Load8<GT>::get(gate + (r + (int64_t)u * rpb) * d + piece * 8, g[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) one(x[u], g[u], r + (int64_t)u * rpb);
    }
    for (; r < r1; r += rpb) {
      float x[8], g[8];
      Load8<ST>::get(src + r * d + piece * 8, x);
      if (gate) Load8<GT>::get(gate + r * d + piece * 8, g);
      one(x, g, r);
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) red[threadIdx.x * 2 + q] = acc[q];
  __syncthreads();
  if (slot == 0) {
    for (int s = 1; s < rpb; ++s)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float4 v = red[(s * ppr + piece) * 2 + q];
        acc[q].x += v.x; acc[q].y += v.y; acc[q].z += v.z; acc[q].w += v.w;
      }
#pragma unroll
    for (int q = 0; q < 2; ++q)
      *reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * d + piece * 8 + 4 * q) = acc[q];
  }
}

// The gradient of a row dot product s[r] = x[r, :] . w -- dx[r, :] = g[r] w, an outer product -- written straight through the
// ReLU / dropout backward gate of the layer whose output x is, with that layer's bias gradient:
// dst[r, c] = bf16(x[r, c] > 0 ? bf16(g[r] w[c]) * scale : 0), partial[block][c] = sum of the block's rows of dst.  The values
// of bkg_scaled_rows (bf16 out) followed by bkg_gate_cast_colsum_bf16, same blocking and addition order as that kernel:
// two passes over [n, d] less (GGD's head behind its folded last conv layer: model/gcl.py).
__global__ __launch_bounds__(256) void outer_gate_colsum_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                                const __bf16* __restrict__ gate, float scale,
                                                                __bf16* __restrict__ dst, int64_t n, int d,
                                                                int64_t rows_per_block, float* __restrict__ partial) {
  __shared__ float4 red[256 * 2];
  const int ppr = d / 8, rpb = 256 / ppr;
  const int piece = threadIdx.x % ppr, slot = threadIdx.x / ppr;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, n);
  float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  if (slot < rpb) {
    float wv[8];
    Load8<float>::get(w + piece * 8, wv);
    constexpr int U = 8;      // one input stream: twice the rows in flight of gate_cast_colsum_kernel
    auto one = [&](float gr, const float (&gt)[8], int64_t r) {
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
      bf16x8 o;
      float back[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = (float)(__bf16)(gr * wv[e]);
        o[e] = (__bf16)(gt[e] > 0.f ? x * scale : 0.f);
        back[e] = (float)o[e];
      }
      *reinterpret_cast<bf16x8*>(dst + r * d + piece * 8) = o;
      acc[0].x += back[0]; acc[0].y += back[1]; acc[0].z += back[2]; acc[0].w += back[3];
      acc[1].x += back[4]; acc[1].y += back[5]; acc[1].z += back[6]; acc[1].w += back[7];
    };
    int64_t r = r0 + slot;
    for (; r + (U - 1) * rpb < r1; r += U * rpb) {
      float gt[U][8], gr[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        Load8<__bf16>::get(gate + (r + (int64_t)u * rpb) * d + piece * 8, gt[u]);
        gr[u] = g[r + (int64_t)u * rpb];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) one(gr[u], gt[u], r + (int64_t)u * rpb);
    }
    for (; r < r1; r += rpb) {
      float gt[8];
      Load8<__bf16>::get(gate + r * d + piece * 8, gt);
      one(g[r], gt, r);
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) red[threadIdx.x * 2 + q] = acc[q];
  __syncthreads();
  if (slot == 0) {
    for (int s = 1; s < rpb; ++s)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float4 v = red[(s * ppr + piece) * 2 + q];
        acc[q].x += v.x; acc[q].y += v.y; acc[q].z += v.z; acc[q].w += v.w;
      }
#pragma unroll
    for (int q = 0; q < 2; ++q)
      *reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * d + piece * 8 + 4 * q) = acc[q];
  }
}

// block = 8 row groups x 32 columns; group g adds partial rows g, g + 8, ... (8 loads in flight), then the groups are
// added in group order through LDS
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int n_blocks, int d,
                                                           float* __restrict__ out) {
  __shared__ float red[8][32];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), g = threadIdx.x >> 5;
  float s = 0.f;
  if (c < d) {
    int b = g;
    for (; b + 56 < n_blocks; b += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(b + 8 * u) * d + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < n_blocks; b += 8) s += partial[(int64_t)b * d + c];
  }
  red[g][threadIdx.x & 31] = s;
  __syncthreads();
  if (g == 0 && c < d) {
    float t = red[0][threadIdx.x];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += red[k][threadIdx.x];
    out[c] = t;
  }
}

// (for launches of other translation units whose kernels leave per-block partial rows: gemm_xw.hip's gated product)
void colsum_final_launch(const float* partial, int n_blocks, int d, float* out, hipStream_t s) {
  hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(d, 32)), dim3(256), 0, s, partial, n_blocks, d, out);
}

// few rows (a mini-batch): ONE launch -- block = 32 row groups x 32 columns over ALL rows (4 rows in flight per
// thread), groups added in group order
template <typename T>
__global__ __launch_bounds__(1024) void colsum_small_kernel(const T* __restrict__ x, int64_t ldx, int n, int d,
                                                            const float* __restrict__ row_w, float* __restrict__ out) {
  __shared__ float red[32][33];
  const int c = blockIdx.x * 32 + (threadIdx.x & 31), g = threadIdx.x >> 5;
  float s = 0.f;
  if (c < d) {
    int r = g;
    for (; r + 96 < n; r += 128) {
      const float v0 = (float)x[(int64_t)r * ldx + c], v1 = (float)x[(int64_t)(r + 32) * ldx + c];
      const float v2 = (float)x[(int64_t)(r + 64) * ldx + c], v3 = (float)x[(int64_t)(r + 96) * ldx + c];
      if (row_w) {
        s = fmaf(row_w[r], v0, s); s = fmaf(row_w[r + 32], v1, s); s = fmaf(row_w[r + 64], v2, s); s = fmaf(row_w[r + 96], v3, s);
      } else {
        s += v0; s += v1; s += v2; s += v3;
      }
    }
    for (; r < n; r += 32) s = row_w ? fmaf(row_w[r], (float)x[(int64_t)r * ldx + c], s) : s + (float)x[(int64_t)r * ldx + c];
  }
  red[g][threadIdx.x & 31] = s;
  __syncthreads();
  if (g == 0 && c < d) {
    float t = red[0][threadIdx.x];
#pragma unroll
    for (int k = 1; k < 32; ++k) t += red[k][threadIdx.x];
    out[c] = t;
  }
}

// out = dout where gate > 0 (times scale), 0 elsewhere: the backward of dropout(relu(.)) read off the saved output
__global__ __launch_bounds__(256) void gate_f32_kernel(const float* __restrict__ dout, const float* __restrict__ gate,
                                                       float scale, int64_t n4, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 a = reinterpret_cast<const float4*>(dout)[i], g = reinterpret_cast<const float4*>(gate)[i];
  float4 r;
  r.x = g.x > 0.f ? a.x * scale : 0.f; r.y = g.y > 0.f ? a.y * scale : 0.f;
  r.z = g.z > 0.f ? a.z * scale : 0.f; r.w = g.w > 0.f ? a.w * scale : 0.f;
  reinterpret_cast<float4*>(out)[i] = r;
}

}  // namespace bkg

using namespace bkg;

extern "C" int bkg_gate_f32(const float* dout, const float* gate, float scale, int64_t numel, float* out, void* stream) {
  BKG_ARG(numel >= 0 && numel % 4 == 0 && (uintptr_t)dout % 16 == 0 && (uintptr_t)gate % 16 == 0 && (uintptr_t)out % 16 == 0,
          "bkg_gate_f32: contiguous 16-byte aligned tables whose size is a multiple of 4");
  if (numel == 0) return 0;
  hipLaunchKernelGGL(gate_f32_kernel, dim3(ceil_div(numel / 4, 256)), dim3(256), 0, (hipStream_t)stream, dout, gate, scale,
                     numel / 4, out);
  return check_launch("bkg_gate_f32");
}

extern "C" int64_t bkg_colsum_workspace_floats(int64_t n, int32_t d) {
  return (int64_t)std::min<int64_t>(CS_MAX_BLOCKS, std::max<int64_t>(1, (n + 63) / 64)) * d;
}

static int colsum_launch(const void* x, int32_t x_is_bf16, int64_t ldx, int64_t n, int32_t d, const float* row_w,
                         float* workspace, float* out, void* stream, const char* who) {
  const int epl = x_is_bf16 ? 8 : 4;
  BKG_ARG(n >= 0 && d > 0 && d % epl == 0 && d / epl <= 256, "column sums: d must be a multiple of the 16-byte piece, <= 256 pieces");
  BKG_ARG((uintptr_t)x % 16 == 0 && ldx % epl == 0 && (uintptr_t)workspace % 16 == 0, "column sums: alignment");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    fill_u32(out, 0u, d, s);
    return check_launch(who);
  }
  if (n <= 4096) {
    if (x_is_bf16) hipLaunchKernelGGL(colsum_small_kernel<__bf16>, dim3(ceil_div(d, 32)), dim3(1024), 0, s, (const __bf16*)x, ldx, (int)n, d, row_w, out);
    else hipLaunchKernelGGL(colsum_small_kernel<float>, dim3(ceil_div(d, 32)), dim3(1024), 0, s, (const float*)x, ldx, (int)n, d, row_w, out);
    return check_launch(who);
  }
  const int n_blocks = (int)(bkg_colsum_workspace_floats(n, d) / d);
  const int64_t rpb = (n + n_blocks - 1) / n_blocks;
  if (x_is_bf16)
    hipLaunchKernelGGL(colsum_partial_kernel<__bf16>, dim3(n_blocks), dim3(256), 0, s, (const __bf16*)x, ldx, n, d, rpb, row_w, workspace);
  else
    hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3(n_blocks), dim3(256), 0, s, (const float*)x, ldx, n, d, rpb, row_w, workspace);
  hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(d, 32)), dim3(256), 0, s, workspace, n_blocks, d, out);
  return check_launch(who);
}

extern "C" int bkg_colsum_f32(const void* x, int32_t x_is_bf16, int64_t ldx, int64_t n, int32_t d, float* workspace,
                              float* out, void* stream) {
  return colsum_launch(x, x_is_bf16, ldx, n, d, nullptr, workspace, out, stream, "bkg_colsum_f32");
}

// out[c] = sum_i row_w[i] * x[i, c]: the weight gradient of a Linear whose outputs are summed over the features
extern "C" int bkg_colsum_weighted_f32(const void* x, int32_t x_is_bf16, int64_t ldx, int64_t n, int32_t d, const float* row_w,
                                       float* workspace, float* out, void* stream) {
  BKG_ARG(row_w != nullptr, "bkg_colsum_weighted_f32: row weights missing");
  return colsum_launch(x, x_is_bf16, ldx, n, d, row_w, workspace, out, stream, "bkg_colsum_weighted_f32");
}

// dst = bf16(gate > 0 ? src * gate_scale : 0) (gate may be NULL: a plain cast) over a contiguous [n, d] table, and
// out[c] = sum_i dst[i, c] -- the bias gradient of the layer whose output gradient this is.  n > 4096 rows (fewer: the
// two-pass form); workspace as for bkg_colsum_f32.
extern "C" int bkg_gate_cast_colsum_bf16(const void* src, int32_t src_is_bf16, const void* gate, int32_t gate_is_bf16,
                                         float gate_scale, void* dst, int64_t n, int32_t d, float* workspace, float* out,
                                         void* stream) {
  BKG_ARG(n > 4096 && d > 0 && d % 8 == 0 && d / 8 <= 256, "bkg_gate_cast_colsum_bf16: n > 4096 rows, d a multiple of 8, <= 2048");
  BKG_ARG(((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0) && (!gate || (uintptr_t)gate % 16 == 0) &&
              (uintptr_t)workspace % 16 == 0,
          "bkg_gate_cast_colsum_bf16: 16-byte alignment");
  hipStream_t s = (hipStream_t)stream;
  const int n_blocks = (int)(bkg_colsum_workspace_floats(n, d) / d);
  const int64_t rpb = (n + n_blocks - 1) / n_blocks;
#define BKG_GCC(ST, GT)                                                                                               \
  hipLaunchKernelGGL((gate_cast_colsum_kernel<ST, GT>), dim3(n_blocks), dim3(256), 0, s, (const ST*)src, (const GT*)gate, \
                     gate_scale, (__bf16*)dst, n, d, rpb, workspace)
  if (src_is_bf16) { if (gate_is_bf16) BKG_GCC(__bf16, __bf16); else BKG_GCC(__bf16, float); }
  else { if (gate_is_bf16) BKG_GCC(float, __bf16); else BKG_GCC(float, float); }
#undef BKG_GCC
  hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(d, 32)), dim3(256), 0, s, workspace, n_blocks, d, out);
  return check_launch("bkg_gate_cast_colsum_bf16");
}


// dst = bf16(gate > 0 ? bf16(g[r] w[c]) * gate_scale : 0) over a contiguous [n, d] bf16 gate table, out[c] = sum_r dst[r, c]:
// bkg_scaled_rows + bkg_gate_cast_colsum_bf16 in one pass (same values, same sums).  n > 4096, d % 8 == 0, d <= 2048.
extern "C" int bkg_outer_gate_colsum_bf16(const float* g, const float* w, const void* gate, float gate_scale, void* dst,
                                          int64_t n, int32_t d, float* workspace, float* out, void* stream) {
  BKG_ARG(n > 4096 && d > 0 && d % 8 == 0 && d / 8 <= 256, "bkg_outer_gate_colsum_bf16: n > 4096 rows, d a multiple of 8, <= 2048");
  BKG_ARG(g && w && gate && dst && workspace && out && ((uintptr_t)w % 16 == 0) && ((uintptr_t)dst % 16 == 0) &&
              ((uintptr_t)gate % 16 == 0) && (uintptr_t)workspace % 16 == 0,
          "bkg_outer_gate_colsum_bf16: arguments / 16-byte alignment");
  hipStream_t s = (hipStream_t)stream;
  const int n_blocks = (int)(bkg_colsum_workspace_floats(n, d) / d);
  const int64_t rpb = (n + n_blocks - 1) / n_blocks;
  hipLaunchKernelGGL(outer_gate_colsum_kernel, dim3(n_blocks), dim3(256), 0, s, g, w, (const __bf16*)gate, gate_scale, (__bf16*)dst,
                     n, d, rpb, workspace);
  hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(d, 32)), dim3(256), 0, s, workspace, n_blocks, d, out);
  return check_launch("bkg_outer_gate_colsum_bf16");
}
