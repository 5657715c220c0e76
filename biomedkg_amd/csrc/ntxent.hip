// NT-Xent / GRACE InfoNCE with intra-view negatives, flash-style on the fp32 matrix cores.
//
// Reference: PyGCL DualBranchContrast(InfoNCE(tau), "L2L", intraview_negs=True) called at
// /root/reference/biomedkg/gcl_module.py:171-173,189.  With
// Z = [normalize(h1); normalize(h2)] (M = 2N rows) that loss is exactly
//     L = 1/M * sum_i ( -z_i.z_p(i)/tau + log D_i ),   D_i = sum_{j != i} exp(z_i.z_j / tau),
// p(i) = (i + N) mod M.  PyGCL materialises ~8 [N,2N] fp32 temporaries per direction; here the
// Gram tiles live only in MFMA accumulators:
//   forward : per 64-row i-tile, stream j-tiles, S^T = Z_J Z_I^T on v_mfma_f32_32x32x2_f32 (exact
//             fp32, so the 1e-5 loss tolerance holds), exp, per-lane running row sums.
//   backward: recompute S^T, W = exp(S/tau) (1/D_i + 1/D_j), and feed W straight from the
//             accumulator registers as the A operand of a second MFMA  dZ_I += W^T Z_J
//             (the tile is computed transposed so that the contraction index sits in registers:
//             no LDS round trip, cdna_hip_programming.md "An accumulator tile as the next MFMA's
//             operand").
// No max-subtraction is needed (|cos| <= 1 so exp(s/tau) <= e^5 at tau = 0.2), matching PyGCL.
//
// LDS images are K-major: Zi[k][i] (stride 68) and Zj[k][j] (stride 65: conflict-free both for the
// first product's A fragment, lanes along j, and the second product's B fragment, lanes along k).
#include "common.h"

namespace bkg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NX_BI = 64, NX_BJ = 64, NX_SI = 68, NX_SJ = 65;

// stage rows [r0, r0+64) of zn (all h columns, zero padded to hp) K-major into lds[k * stride + r]
__device__ __forceinline__ void nx_stage(const float* __restrict__ zn, int64_t ld, int M, int h, int hp, int r0,
                                         float* __restrict__ lds, int stride) {
  const int chunks = hp / 4;  // float4 chunks per row (hp multiple of 4)
  for (int f = threadIdx.x; f < 64 * chunks; f += 256) {
    const int r = f / chunks, kc = (f % chunks) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < M) {
      const float* p = zn + (int64_t)(r0 + r) * ld + kc;
      if (kc + 3 < h && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) v = *reinterpret_cast<const float4*>(p);
      else {
        if (kc < h) v.x = p[0];
        if (kc + 1 < h) v.y = p[1];
        if (kc + 2 < h) v.z = p[2];
        if (kc + 3 < h) v.w = p[3];
      }
    }
    lds[(kc + 0) * stride + r] = v.x;
    lds[(kc + 1) * stride + r] = v.y;
    lds[(kc + 2) * stride + r] = v.z;
    lds[(kc + 3) * stride + r] = v.w;
  }
}

// HT = ceil(h / 32) column tiles of the dZ accumulator (backward only).
template <int HT, bool BWD>
__global__ __launch_bounds__(256) void ntxent_kernel(const float* __restrict__ zn, int64_t ld, int M, int h,
                                                     float inv_tau, int j_tiles_per_split,
                                                     const float* __restrict__ rden,  // [M] 1/D (BWD)
                                                     float* __restrict__ out_rowsum,  // [splits][M] (FWD)
                                                     float* __restrict__ out_dz,      // [splits][M][h] (BWD)
                                                     int n_i_tiles) {
  extern __shared__ float smem[];
  const int hp = (h + 3) & ~3;
  float* Zi = smem;                    // [hp][NX_SI]
  float* Zj = Zi + hp * NX_SI;         // [hp][NX_SJ]
  float* rdj = Zj + hp * NX_SJ;        // [64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int is = wave & 1, js = wave >> 1;
  const int i_tile = blockIdx.x % n_i_tiles, split = blockIdx.x / n_i_tiles;
  const int i0 = i_tile * NX_BI;
  const int n_j_tiles = (M + NX_BJ - 1) / NX_BJ;
  const int jt_beg = split * j_tiles_per_split, jt_end = min(jt_beg + j_tiles_per_split, n_j_tiles);

  nx_stage(zn, ld, M, h, hp, i0, Zi, NX_SI);
  const int i_glob = i0 + is * 32 + l31;
  const float rd_i = (BWD && i_glob < M) ? rden[i_glob] : 0.f;

  float rowsum = 0.f;
  f32x16 dacc[BWD ? HT : 1];
  if (BWD) {
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) dacc[t][r] = 0.f;
  }

  for (int jt = jt_beg; jt < jt_end; ++jt) {
    const int j0 = jt * NX_BJ;
    __syncthreads();  // previous tile's readers done (also orders the Zi staging on the first pass)
    nx_stage(zn, ld, M, h, hp, j0, Zj, NX_SJ);
    if (BWD && threadIdx.x < 64) rdj[threadIdx.x] = (j0 + threadIdx.x < M) ? rden[j0 + threadIdx.x] : 0.f;
    __syncthreads();

    // S^T[j, i] = sum_k Zj[k][j] * Zi[k][i]   (rows j -> registers, cols i -> lanes)
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    for (int k = 0; k < hp; k += 2) {
      const float a = Zj[(k + lh) * NX_SJ + js * 32 + l31];
      const float b = Zi[(k + lh) * NX_SI + is * 32 + l31];
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, s, 0, 0, 0);
    }
    f32x16 w;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jr = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int j_glob = j0 + js * 32 + jr;
      const bool valid = (j_glob < M) && (i_glob < M) && (j_glob != i_glob);
      const float e = valid ? expf(s[r] * inv_tau) : 0.f;
      if (BWD) w[r] = e * (rd_i + rdj[js * 32 + jr]);
      else rowsum += e;
    }
    if (BWD) {
      // dZ[i, c] += sum_j W[j, i] * Zj[c][j]: register r of W is the A operand for k = {jr(lh=0), jr+4(lh=1)}
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        const int c = t * 32 + l31;
        const bool c_ok = c < hp;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int jr = (r & 3) + 8 * (r >> 2) + 4 * lh;
          const float b = c_ok ? Zj[c * NX_SJ + js * 32 + jr] : 0.f;
          dacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], b, dacc[t], 0, 0, 0);
        }
      }
    }
  }

  __syncthreads();
  if (!BWD) {
    // combine lane halves, then the two j-slab waves sharing this i-slab
    rowsum += __shfl_xor(rowsum, 32, 64);
    float* red = Zj;  // reuse
    if (lh == 0) red[wave * 32 + l31] = rowsum;
    __syncthreads();
    if (threadIdx.x < 64) {
      const int isl = threadIdx.x >> 5, l = threadIdx.x & 31;
      const float v = red[(isl)*32 + l] + red[(isl + 2) * 32 + l];  // waves (is, js=0) and (is, js=1)
      const int ig = i0 + isl * 32 + l;
      if (ig < M) out_rowsum[(int64_t)split * M + ig] = v;
    }
  } else {
    // dacc[t][r]: row i = is*32 + (r&3)+8(r>>2)+4lh, col c = t*32 + l31.  Sum the js = 0/1 waves via LDS.
    float* red = Zj;  // needs 64 * (HT*32) floats <= hp * NX_SJ  (HT*32 <= hp + 31)
    const int cw = HT * 32;
    if (js == 1) {
#pragma unroll
      for (int t = 0; t < HT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ir = is * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          red[ir * cw + t * 32 + l31] = dacc[t][r];
        }
    }
    __syncthreads();
    if (js == 0) {
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        const int c = t * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ir = is * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int ig = i0 + ir;
          if (ig < M && c < h) out_dz[((int64_t)split * M + ig) * h + c] = dacc[t][r] + red[ir * cw + c];
        }
      }
    }
  }
}

static size_t nx_lds_bytes(int h, int ht) {
  int hp = (h + 3) & ~3;
  size_t base = (size_t)hp * (NX_SI + NX_SJ) + 64;
  size_t red = (size_t)hp * NX_SI + (size_t)64 * ht * 32;  // reduction buffer aliases Zj
  return sizeof(float) * (base > red ? base : red);
}

}  // namespace bkg

using namespace bkg;

// rowsum_partial: [splits, M]; caller sums over splits (fixed order) to get D.
extern "C" int bkg_ntxent_rowsum_f32(const float* zn, int64_t ld, int64_t M, int32_t h, float tau, int32_t splits,
                                     float* rowsum_partial, void* stream) {
  BKG_ARG(M > 0 && h > 0 && h <= 256 && splits > 0 && tau > 0.f, "bkg_ntxent_rowsum_f32: bad arguments (h <= 256)");
  const int n_i = ceil_div(M, NX_BI), n_j = ceil_div(M, NX_BJ);
  const int per = ceil_div(n_j, splits);
  size_t lds = nx_lds_bytes(h, 1);
  hipFuncSetAttribute((const void*)ntxent_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((ntxent_kernel<1, false>), dim3(n_i * splits), dim3(256), lds, (hipStream_t)stream, zn, ld, (int)M,
                     h, 1.0f / tau, per, nullptr, rowsum_partial, nullptr, n_i);
  return check_launch("bkg_ntxent_rowsum_f32");
}

// dz_partial: [splits, M, h] = sum_{j != i} exp(s_ij/tau) (1/D_i + 1/D_j) zn_j over the split's j range.
extern "C" int bkg_ntxent_bwd_f32(const float* zn, int64_t ld, int64_t M, int32_t h, float tau, const float* rden,
                                  int32_t splits, float* dz_partial, void* stream) {
  BKG_ARG(M > 0 && h > 0 && h <= 256 && splits > 0 && tau > 0.f, "bkg_ntxent_bwd_f32: bad arguments (h <= 256)");
  const int n_i = ceil_div(M, NX_BI), n_j = ceil_div(M, NX_BJ);
  const int per = ceil_div(n_j, splits);
  int ht = ceil_div(h, 32);
  ht = ht <= 1 ? 1 : ht <= 2 ? 2 : ht <= 4 ? 4 : 8;  // the instantiated accumulator widths
  size_t lds = nx_lds_bytes(h, ht);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(n_i * splits), block(256);
#define NX_LAUNCH(HT)                                                                                              \
  hipFuncSetAttribute((const void*)ntxent_kernel<HT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
  hipLaunchKernelGGL((ntxent_kernel<HT, true>), grid, block, lds, s, zn, ld, (int)M, h, 1.0f / tau, per, rden,     \
                     nullptr, dz_partial, n_i)
  if (ht <= 1) { NX_LAUNCH(1); }
  else if (ht <= 2) { NX_LAUNCH(2); }
  else if (ht <= 4) { NX_LAUNCH(4); }
  else { NX_LAUNCH(8); }
#undef NX_LAUNCH
  return check_launch("bkg_ntxent_bwd_f32");
}
