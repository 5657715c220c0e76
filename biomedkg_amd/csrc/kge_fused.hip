// Fused KGE training pass: triplet scores + BCE-with-logits terms + d loss / d score + d rel_emb in ONE
// sweep over the triplets' head / tail rows.
//
// Reference: KGEModule.training_step (/root/reference/biomedkg/kge_module.py:121-128) runs
// decode(pos), decode(neg), cat, ones/zeros, binary_cross_entropy_with_logits -- and autograd then walks
// the [T,d] gathers again.  Unfused, this build needs three passes over the 2T rows (score, node-side
// gradient, relation-side gradient); because BCE's gradient g = sigmoid(s) - y only needs the score and
// the label, the relation-side gradient can be accumulated in the scoring pass itself, leaving one more
// pass for the node side (kge_bwd_nodes.hip, fed with the g written here).  Triplets are visited in
// relation-sorted order so that a wave owns a slice of ONE relation and keeps its d rel_emb partial in
// registers (stored as one partial row per wave; the caller segment-reduces the waves of a relation).
//
// The node table z is fp32 or bf16 (rowpiece.h); rel_emb, scores and all gradients are fp32.
// One launch per label (positives y = 1, negatives y = 0).  Outputs: score[T] (original triplet order),
// g[T] = (sigmoid(s) - y) * inv_total (or, for the node-side backward, one packed int4 record {head, tail, rel,
// bits(g)} per triplet: a single 16-byte gather per incidence there), loss_part[n_waves] (sum it, multiply by inv_total) and
// drel_part[n_waves, d]: wave w's partial of d rel_emb[r(w)], where relation r owns the consecutive waves
// [sum_{q<r} ceil(cnt_q / 64), ...) -- rows of waves without work are left unwritten and are never read.
// The caller reduces drel_part per relation and multiplies by the upstream gradient of the loss.
#include "common.h"
#include "rowpiece.h"

#ifndef KF_CH
#define KF_CH 64     // triplets per wave (biomedkg_amd/ops.py _FUSED_CHUNK must agree: bkg_kge_fused_chunk)
#endif
#ifndef KF_L32
#define KF_L32 8     // lanes per row of a 32-piece table (d = 256 bf16) in the fused scorer (swept: docs/experiments.md 9-4)
#endif

namespace bkg {

enum { KF_DISTMULT = 0, KF_TRANSE = 1 };
constexpr float KF_EPS = 1e-12f;

__device__ __forceinline__ float kf_sgn(float v) { return (v > 0.f) - (v < 0.f); }
__device__ __forceinline__ float4 kf_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float kf_abs4(const float4& a) { return fabsf(a.x) + fabsf(a.y) + fabsf(a.z) + fabsf(a.w); }

template <int LPR>
__device__ __forceinline__ float kf_group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <typename T, int KIND, int LPR, int VPL>
__global__ __launch_bounds__(256) void kge_fused_bce_kernel(const T* __restrict__ z, int64_t ldz,
                                                            const int64_t* __restrict__ head,
                                                            const int64_t* __restrict__ tail, int64_t T_, float label,
                                                            float inv_t, const float* __restrict__ inv_t_dev,
                                                            const float* __restrict__ rel_emb,
                                                            int64_t ldr, int n_rel, const int32_t* __restrict__ order,
                                                            const int32_t* __restrict__ rel_ptr, int chunk,
                                                            float* __restrict__ score, float* __restrict__ g_out, int4* __restrict__ rec,
                                                            float* __restrict__ loss_part, float* __restrict__ drel,
                                                            int64_t lddr) {
  constexpr int G = 64 / LPR, Q = Piece<T>::Q, EPL = Piece<T>::EPL, VQ = VPL * Q;
  const int lane = threadIdx.x & 63, grp = lane / LPR, lig = lane % LPR;
  const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (inv_t_dev) inv_t = inv_t_dev[0];      // padded mini-batches: the triplet count lives in device memory
  int w = __builtin_amdgcn_readfirstlane(wave_id);
  int r = 0, beg = 0, end = 0;
  for (; r < n_rel; ++r) {  // wave-uniform scan of <= n_rel scalars
    const int a = rel_ptr[r], b = rel_ptr[r + 1];
    const int nc = (b - a + chunk - 1) / chunk;
    if (w < nc) { beg = a + w * chunk; end = min(beg + chunk, b); break; }
    w -= nc;
  }
  float loss_acc = 0.f;
  if (r < n_rel) {
    float4 acc[VQ], rr[VQ];
    zero4(acc);
#pragma unroll
    for (int q = 0; q < VQ; ++q) rr[q] = kf_ld4(rel_emb + (int64_t)r * ldr + piece_col<T, LPR>(lig, q));
    // Batches of U triplets per lane group through a three-stage software pipeline: while the rows of batch i
    // are in flight, the endpoints of batch i+1 and the triplet ids of batch i+2 are too (issued BEFORE the rows,
    // so the one wait per iteration -- for the rows -- retires all three).  The dependent chain id -> endpoints ->
    // rows therefore costs one memory latency per batch instead of three.  Ids past the slice are clamped.
    constexpr int U = (VQ == 1) ? 4 : (VQ >= 8 ? 1 : 2);  // keeps the batch at <= 8 float4 per operand (VGPRs -> occupancy)
    constexpr int STEP = U * G;
    int tt[U], tt_n[U];
    int64_t hh[U], ta[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      tt[u] = order[min(beg + u * G + grp, end - 1)];
      tt_n[u] = order[min(beg + STEP + u * G + grp, end - 1)];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) { hh[u] = head[tt[u]]; ta[u] = tail[tt[u]]; }
    for (int k0 = beg; k0 < end; k0 += STEP) {
      int tt_nn[U];
      int64_t hh_n[U], ta_n[U];
      bool lv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        lv[u] = k0 + u * G + grp < end;
        hh_n[u] = head[tt_n[u]];
        ta_n[u] = tail[tt_n[u]];
        tt_nn[u] = order[min(k0 + 2 * STEP + u * G + grp, end - 1)];
      }
      Piece<T> hb[U][VPL], tb[U][VPL];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          hb[u][v].load(z + hh[u] * ldz + EPL * (lig + v * LPR));
          tb[u][v].load(z + ta[u] * ldz + EPL * (lig + v * LPR));
        }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float4 h[VQ], tl[VQ];
#pragma unroll
        for (int v = 0; v < VPL; ++v) { hb[u][v].expand(h + v * Q); tb[u][v].expand(tl + v * Q); }
        float s = 0.f, ih = 0.f, it = 0.f;
        if (KIND == KF_DISTMULT) {
#pragma unroll
          for (int q = 0; q < VQ; ++q)
            s += h[q].x * rr[q].x * tl[q].x + h[q].y * rr[q].y * tl[q].y + h[q].z * rr[q].z * tl[q].z + h[q].w * rr[q].w * tl[q].w;
          s = kf_group_sum<LPR>(s);
        } else {
          float nh = 0.f, nt = 0.f;
#pragma unroll
          for (int q = 0; q < VQ; ++q) { nh += kf_abs4(h[q]); nt += kf_abs4(tl[q]); }
          ih = 1.0f / fmaxf(kf_group_sum<LPR>(nh), KF_EPS);
          it = 1.0f / fmaxf(kf_group_sum<LPR>(nt), KF_EPS);
#pragma unroll
          for (int q = 0; q < VQ; ++q)
            s += fabsf(h[q].x * ih + rr[q].x - tl[q].x * it) + fabsf(h[q].y * ih + rr[q].y - tl[q].y * it) +
                 fabsf(h[q].z * ih + rr[q].z - tl[q].z * it) + fabsf(h[q].w * ih + rr[q].w - tl[q].w * it);
          s = -kf_group_sum<LPR>(s);
        }
        // BCE with logits (torch formula): max(s,0) - s*y + log1p(exp(-|s|));  d/ds = sigmoid(s) - y
        const float e = expf(-fabsf(s));
        const float sig = s >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
        const float g = lv[u] ? (sig - label) * inv_t : 0.f;
        if (lv[u] && lig == 0) {
          score[tt[u]] = s;
          if (g_out) g_out[tt[u]] = g;
          // packed copy of the triplet for the node-side backward: one 16-byte gather there instead of four
          if (rec) rec[tt[u]] = make_int4((int)hh[u], (int)ta[u], r, __float_as_int(g));
          loss_acc += fmaxf(s, 0.f) - s * label + log1pf(e);
        }
        if (KIND == KF_DISTMULT) {
#pragma unroll
          for (int q = 0; q < VQ; ++q) {
            acc[q].x = fmaf(g * h[q].x, tl[q].x, acc[q].x); acc[q].y = fmaf(g * h[q].y, tl[q].y, acc[q].y);
            acc[q].z = fmaf(g * h[q].z, tl[q].z, acc[q].z); acc[q].w = fmaf(g * h[q].w, tl[q].w, acc[q].w);
          }
        } else {
#pragma unroll
          for (int q = 0; q < VQ; ++q) {
            acc[q].x -= g * kf_sgn(h[q].x * ih + rr[q].x - tl[q].x * it); acc[q].y -= g * kf_sgn(h[q].y * ih + rr[q].y - tl[q].y * it);
            acc[q].z -= g * kf_sgn(h[q].z * ih + rr[q].z - tl[q].z * it); acc[q].w -= g * kf_sgn(h[q].w * ih + rr[q].w - tl[q].w * it);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) { tt[u] = tt_n[u]; hh[u] = hh_n[u]; ta[u] = ta_n[u]; tt_n[u] = tt_nn[u]; }
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
    // Per-wave partial row, plain store.  (Float atomics into the R rows of drel were the first version: with
    // ~70 % of the triplets in two relations every wave hammers the same two 1-KiB rows, the 14x-slower
    // "everyone on one row" case of MI355X_MICROARCH.md "Global float atomics": 6.9 ms per launch.)  The
    // waves of one relation are consecutive, so the caller finishes with one segment reduce over wave rows.
    if (grp == 0) {
#pragma unroll
      for (int q = 0; q < VQ; ++q)
        *reinterpret_cast<float4*>(drel + (int64_t)wave_id * lddr + piece_col<T, LPR>(lig, q)) = acc[q];
    }
  }
  // per-wave loss partial (fixed-order sum by the caller): only lig == 0 lanes hold terms
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) loss_acc += __shfl_xor(loss_acc, o, 64);
  if (lane == 0) loss_part[wave_id] = loss_acc;
}

// wave_ptr[r] = first wave of relation r in the fused pass (relation r owns ceil(cnt_r / chunk) consecutive waves)
__global__ void relation_wave_ptr_kernel(const int32_t* __restrict__ rel_ptr, int n_rel, int chunk,
                                         int32_t* __restrict__ wave_ptr) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int acc = 0;
  for (int r = 0; r < n_rel; ++r) {
    wave_ptr[r] = acc;
    acc += (rel_ptr[r + 1] - rel_ptr[r] + chunk - 1) / chunk;
  }
  wave_ptr[n_rel] = acc;
}

// ---- the scalar tail of a MINI-BATCH step in one launch each (a captured step pays ~5 us per launch, and the torch
// expressions these replace were ~30 of them): fixed thread -> element mapping and a fixed-order tree, so repeatable.
__device__ __forceinline__ double block_sum_1024(double v, double* sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
#pragma unroll
  for (int o = 512; o > 0; o >>= 1) {
    if (t < o) sh[t] += sh[t + o];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

// loss = inv_total * sum(part) + reg * (sum_{rows < n_real}(z^2) / (mean_rows d) + mean(rel^2)).  The launch's blocks
// (FIN_BLOCKS for a mini-batch, up to FIN_BLOCKS_MAX for a whole graph's z) add their slices (fixed thread -> element
// mapping, fixed-order tree) into slots of `scratch`; the block that takes the last ticket adds the slots (one per
// thread, the same tree) and writes the loss -- one launch, repeatable.  scratch: 3 * FIN_BLOCKS_MAX doubles + one uint32
// ticket (zero before the first use; the kernel leaves it zero).
constexpr int FIN_BLOCKS = 64, FIN_BLOCKS_MAX = 1024;

__global__ __launch_bounds__(1024) void kge_loss_finish_kernel(const float* __restrict__ part, int n_part, float inv_total,
                                                               const float* __restrict__ inv_total_dev,
                                                               const float* __restrict__ z, int64_t ldz, int n_cap, int d,
                                                               const int64_t* __restrict__ n_real_dev, double mean_rows,
                                                               const float* __restrict__ rel, int64_t ldr, int n_rel,
                                                               float reg, double* __restrict__ scratch,
                                                               unsigned int* __restrict__ ticket, float* __restrict__ loss) {
  __shared__ double sh[1024];
  __shared__ bool last;
  const int t = threadIdx.x, b = blockIdx.x, nb = gridDim.x;
  const int64_t gt = (int64_t)b * 1024 + t, stride = (int64_t)nb * 1024;
  double a = 0.0, zz = 0.0, rr = 0.0;
  for (int64_t i = gt; i < n_part; i += stride) a += (double)part[i];
  int n_rows = n_cap;
  if (n_real_dev) n_rows = (int)min((int64_t)n_cap, n_real_dev[0]);
  if (reg != 0.f) {
    const int d4 = d >> 2;                     // rows are read as float4 (d % 4 == 0, 16-byte aligned rows: host check)
    const int64_t nz = (int64_t)n_rows * d4, nr = (int64_t)n_rel * d4;
    if (ldz == d) {                            // contiguous rows (a whole graph's z): one flat stream, no 64-bit division per piece
      const float4* zf = reinterpret_cast<const float4*>(z);
      for (int64_t e = gt; e < nz; e += stride) {
        const float4 v = zf[e];
        zz += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
      }
    } else {
      for (int64_t e = gt; e < nz; e += stride) {
        const float4 v = *reinterpret_cast<const float4*>(z + (e / d4) * ldz + (e % d4) * 4);
        zz += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
      }
    }
    for (int64_t e = gt; e < nr; e += stride) {
      const float4 v = *reinterpret_cast<const float4*>(rel + (e / d4) * ldr + (e % d4) * 4);
      rr += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
  }
  a = block_sum_1024(a, sh);
  zz = block_sum_1024(zz, sh);
  rr = block_sum_1024(rr, sh);
  if (t == 0) {
    scratch[b] = a; scratch[FIN_BLOCKS_MAX + b] = zz; scratch[2 * FIN_BLOCKS_MAX + b] = rr;
    __threadfence();
    last = atomicAdd(ticket, 1u) == (unsigned)(nb - 1);
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  const double sa = block_sum_1024(t < nb ? scratch[t] : 0.0, sh);
  const double sz = block_sum_1024(t < nb ? scratch[FIN_BLOCKS_MAX + t] : 0.0, sh);
  const double sr = block_sum_1024(t < nb ? scratch[2 * FIN_BLOCKS_MAX + t] : 0.0, sh);
  if (t != 0) return;
  const double it = inv_total_dev ? (double)inv_total_dev[0] : (double)inv_total;
  double out = sa * it;
  const double rows_of_mean = mean_rows > 0.0 ? mean_rows : (double)max(n_rows, 1);
  if (reg != 0.f) out += (double)reg * (sz / (rows_of_mean * d) + sr / ((double)n_rel * d));
  loss[0] = (float)out;
  *ticket = 0u;
}

// dst[row, :] = dst[row, :] * g + src[row, :] * (g * coef / rows)   for row < rows (= n_real or n), dst * g beyond;
// g = gloss[0] (the upstream gradient of the loss)
// (four columns per thread when the rows allow 16-byte pieces: the full-graph backward runs this over [N, d])
__global__ __launch_bounds__(256) void scale_add_rows4_kernel(float* __restrict__ dst, int64_t ldd,
                                                              const float* __restrict__ src, int64_t lds, int n, int d,
                                                              const float* __restrict__ gloss, float coef,
                                                              const int64_t* __restrict__ n_real_dev) {
  const int d4 = d >> 2;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)n * d4) return;
  const int row = (int)(e / d4), col = (int)(e % d4) * 4;
  const int rows = n_real_dev ? (int)min((int64_t)n, n_real_dev[0]) : n;
  const float g = gloss[0];
  float4 v = *reinterpret_cast<const float4*>(dst + (int64_t)row * ldd + col);
  v.x *= g; v.y *= g; v.z *= g; v.w *= g;
  if (row < rows) {
    const float c = g * (coef / (float)max(rows, 1));
    const float4 s = *reinterpret_cast<const float4*>(src + (int64_t)row * lds + col);
    v.x = fmaf(s.x, c, v.x); v.y = fmaf(s.y, c, v.y); v.z = fmaf(s.z, c, v.z); v.w = fmaf(s.w, c, v.w);
  }
  *reinterpret_cast<float4*>(dst + (int64_t)row * ldd + col) = v;
}

__global__ __launch_bounds__(256) void scale_add_rows_kernel(float* __restrict__ dst, int64_t ldd,
                                                             const float* __restrict__ src, int64_t lds, int n, int d,
                                                             const float* __restrict__ gloss, float coef,
                                                             const int64_t* __restrict__ n_real_dev) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)n * d) return;
  const int row = (int)(e / d), col = (int)(e % d);
  const int rows = n_real_dev ? (int)min((int64_t)n, n_real_dev[0]) : n;
  const float g = gloss[0];
  float v = dst[(int64_t)row * ldd + col] * g;
  if (row < rows) v = fmaf(src[(int64_t)row * lds + col], g * (coef / (float)max(rows, 1)), v);
  dst[(int64_t)row * ldd + col] = v;
}

// out[row, :] = base[row, :] * g + src[row, :] * (g * coef / n): the out-of-place form (base is a tensor the autograd graph
// saved: the relation embeddings' gradient partials, a few dozen rows)
__global__ __launch_bounds__(256) void scale_add_rows_out_kernel(const float* __restrict__ base, int64_t ldb,
                                                                 const float* __restrict__ src, int64_t lds, int n, int d,
                                                                 const float* __restrict__ gloss, float coef,
                                                                 float* __restrict__ out, int64_t ldo) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)n * d) return;
  const int row = (int)(e / d), col = (int)(e % d);
  const float g = gloss[0];
  out[(int64_t)row * ldo + col] = fmaf(src[(int64_t)row * lds + col], g * (coef / (float)max(n, 1)), base[(int64_t)row * ldb + col] * g);
}

// out = (a + b) * g + z * (g * coef / n): the two walks' node-side sums, the upstream gradient, the L2 term (float4 per lane)
__global__ __launch_bounds__(256) void kge_dz_finish_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b,
                                                            int64_t ldb, const float* __restrict__ z, int64_t ldz, int64_t n, int d4,
                                                            const float* __restrict__ gloss, float coef,
                                                            float* __restrict__ out, int64_t ldo) {
  const float g = gloss[0], c = g * (coef / (float)(n > 0 ? n : 1));
  const int64_t total = n * d4;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t row = e / d4;
    const int col = 4 * (int)(e - row * d4);
    float4 v = *reinterpret_cast<const float4*>(a + row * lda + col);
    if (b) {
      const float4 w = *reinterpret_cast<const float4*>(b + row * ldb + col);
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    v.x *= g; v.y *= g; v.z *= g; v.w *= g;
    if (z) {
      const float4 s = *reinterpret_cast<const float4*>(z + row * ldz + col);
      v.x = fmaf(s.x, c, v.x); v.y = fmaf(s.y, c, v.y); v.z = fmaf(s.z, c, v.z); v.w = fmaf(s.w, c, v.w);
    }
    *reinterpret_cast<float4*>(out + row * ldo + col) = v;
  }
}

}  // namespace bkg

using namespace bkg;

extern "C" int bkg_kge_dz_finish_f32(const float* a, int64_t lda, const float* b, int64_t ldb, const float* z, int64_t ldz,
                                     int64_t n, int32_t d, const float* gloss, float coef, float* out, int64_t ldo,
                                     void* stream) {
  BKG_ARG(a && gloss && out && n >= 0 && d > 0 && d % 4 == 0, "bkg_kge_dz_finish_f32: bad arguments");
  BKG_ARG((uintptr_t)a % 16 == 0 && lda % 4 == 0 && (!b || ((uintptr_t)b % 16 == 0 && ldb % 4 == 0)) &&
              (!z || ((uintptr_t)z % 16 == 0 && ldz % 4 == 0)) && (uintptr_t)out % 16 == 0 && ldo % 4 == 0,
          "bkg_kge_dz_finish_f32: 16-byte aligned rows");
  if (n == 0) return 0;
  const int64_t total = n * (d / 4);
  const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(kge_dz_finish_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb, z, ldz, n, d / 4,
                     gloss, coef, out, ldo);
  return check_launch("bkg_kge_dz_finish_f32");
}

extern "C" int64_t bkg_kge_loss_finish_scratch_bytes() { return 3 * FIN_BLOCKS_MAX * sizeof(double) + 16; }

extern "C" int bkg_kge_loss_finish_f32(const float* part, int32_t n_part, float inv_total, const float* inv_total_dev,
                                       const float* z, int64_t ldz, int32_t n_cap, int32_t d, const int64_t* n_real_dev,
                                       double mean_rows, const float* rel, int64_t ldr, int32_t n_rel, float reg,
                                       void* scratch, float* loss, void* stream) {
  BKG_ARG(n_part >= 0 && n_cap >= 0 && d > 0 && n_rel > 0 && mean_rows >= 0.0, "bkg_kge_loss_finish_f32: bad sizes");
  BKG_ARG(d % 4 == 0 && ldz % 4 == 0 && ldr % 4 == 0 && (uintptr_t)z % 16 == 0 && (uintptr_t)rel % 16 == 0 &&
              (uintptr_t)scratch % 8 == 0, "bkg_kge_loss_finish_f32: 16-byte aligned rows, d % 4 == 0");
  double* slots = reinterpret_cast<double*>(scratch);
  unsigned int* ticket = reinterpret_cast<unsigned int*>(slots + 3 * FIN_BLOCKS_MAX);
  // ~8 sixteen-byte pieces per thread: 64 blocks for a mini-batch, the whole chip for a full graph's z
  const int64_t work = std::max<int64_t>(n_part, reg != 0.f ? (int64_t)n_cap * (d / 4) : 0);
  const int blocks = (int)std::min<int64_t>(FIN_BLOCKS_MAX, std::max<int64_t>(FIN_BLOCKS, ceil_div(work, (int64_t)1024 * 8)));
  hipLaunchKernelGGL(kge_loss_finish_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, part, n_part, inv_total,
                     inv_total_dev, z, ldz, n_cap, d, n_real_dev, mean_rows, rel, ldr, n_rel, reg, slots, ticket, loss);
  return check_launch("bkg_kge_loss_finish_f32");
}

extern "C" int bkg_scale_add_rows_f32(float* dst, int64_t ldd, const float* src, int64_t lds, int32_t n, int32_t d,
                                      const float* gloss, float coef, const int64_t* n_real_dev, void* stream) {
  BKG_ARG(n >= 0 && d > 0, "bkg_scale_add_rows_f32: bad sizes");
  if (n == 0) return 0;
  if (d % 4 == 0 && ldd % 4 == 0 && lds % 4 == 0 && (uintptr_t)dst % 16 == 0 && (uintptr_t)src % 16 == 0)
    hipLaunchKernelGGL(scale_add_rows4_kernel, dim3(ceil_div((int64_t)n * (d / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                       dst, ldd, src, lds, n, d, gloss, coef, n_real_dev);
  else
    hipLaunchKernelGGL(scale_add_rows_kernel, dim3(ceil_div((int64_t)n * d, 256)), dim3(256), 0, (hipStream_t)stream, dst, ldd,
                       src, lds, n, d, gloss, coef, n_real_dev);
  return check_launch("bkg_scale_add_rows_f32");
}

extern "C" int bkg_relation_wave_ptr(const int32_t* rel_ptr, int32_t n_rel, int32_t chunk, int32_t* wave_ptr, void* stream) {
  BKG_ARG(n_rel > 0 && chunk > 0, "bkg_relation_wave_ptr: bad sizes");
  hipLaunchKernelGGL(relation_wave_ptr_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, rel_ptr, n_rel, chunk, wave_ptr);
  return check_launch("bkg_relation_wave_ptr");
}

extern "C" int32_t bkg_kge_fused_chunk() { return KF_CH; }
extern "C" int64_t bkg_kge_fused_bce_waves(int64_t T, int32_t n_rel) {
  return ((T + KF_CH - 1) / KF_CH + n_rel + 3) / 4 * 4;
}

// `inv_total_dev` (nullable): read 1 / total from device memory instead of the scalar argument.  Triplets whose
// relation id sorts past `n_rel` (rel_ptr has more groups than n_rel) are never visited: that is how the padded
// mini-batches of biomedkg_amd/sampled.py keep their pad triplets out of the loss.
extern "C" int bkg_kge_fused_bce_dev_f32(int32_t kind, const void* z, int32_t z_is_bf16, int64_t ldz, int32_t d,
                                         const int64_t* head, const int64_t* tail, int64_t T, float label,
                                         float inv_total, const float* inv_total_dev, const float* rel_emb, int64_t ldr,
                                         int32_t n_rel, const int32_t* order, const int32_t* rel_ptr, float* score,
                                         float* g, void* rec, float* loss_part, float* drel, int64_t lddr, void* stream) {
  BKG_ARG(kind == KF_DISTMULT || kind == KF_TRANSE, "bkg_kge_fused_bce_f32: bad kind");
  const int epl = z_is_bf16 ? 8 : 4;
  BKG_ARG(((uintptr_t)z % 16 == 0) && (ldz % epl == 0) && (d % epl == 0) && ((uintptr_t)rel_emb % 16 == 0) &&
              (ldr % 4 == 0) && ((uintptr_t)drel % 16 == 0) && (lddr % 4 == 0), "bkg_kge_fused_bce_f32: alignment");
  BKG_ARG(T > 0 && n_rel > 0 && (label == 0.f || label == 1.f), "bkg_kge_fused_bce_f32: bad sizes / label");
  hipStream_t s = (hipStream_t)stream;
  // triplets per wave: small enough that T / CH waves oversubscribe the chip several times (each wave walks its
  // slice serially through a 3-deep dependent-load chain), large enough that the 1-KiB atomic flush stays rare
  constexpr int CH = KF_CH;
  dim3 grid((unsigned)(bkg_kge_fused_bce_waves(T, n_rel) / 4)), block(256);
#define KF_LAUNCH(TT, K)                                                                                            \
  hipLaunchKernelGGL((kge_fused_bce_kernel<TT, K, LPR, VPL>), grid, block, 0, s, (const TT*)z, ldz, head, tail, T, \
                     label, inv_total, inv_total_dev, rel_emb, ldr, n_rel, order, rel_ptr, CH, score, g, reinterpret_cast<int4*>(rec), loss_part,  \
                     drel, lddr)
  BKG_PIECE_DISPATCH_(d / epl, KF_L32, 16, {
    if (z_is_bf16) { if (kind == KF_DISTMULT) KF_LAUNCH(__bf16, KF_DISTMULT); else KF_LAUNCH(__bf16, KF_TRANSE); }
    else { if (kind == KF_DISTMULT) KF_LAUNCH(float, KF_DISTMULT); else KF_LAUNCH(float, KF_TRANSE); }
  })
#undef KF_LAUNCH
  return check_launch("bkg_kge_fused_bce_f32");
}

extern "C" int bkg_kge_fused_bce_f32(int32_t kind, const void* z, int32_t z_is_bf16, int64_t ldz, int32_t d,
                                     const int64_t* head, const int64_t* tail, int64_t T, float label,
                                     float inv_total, const float* rel_emb, int64_t ldr, int32_t n_rel,
                                     const int32_t* order, const int32_t* rel_ptr, float* score, float* g,
                                     void* rec, float* loss_part, float* drel, int64_t lddr, void* stream) {
  return bkg_kge_fused_bce_dev_f32(kind, z, z_is_bf16, ldz, d, head, tail, T, label, inv_total, nullptr, rel_emb, ldr, n_rel,
                                   order, rel_ptr, score, g, rec, loss_part, drel, lddr, stream);
}

extern "C" int bkg_scale_add_rows_out_f32(const float* base, int64_t ldb, const float* src, int64_t lds, int32_t n, int32_t d,
                                          const float* gloss, float coef, float* out, int64_t ldo, void* stream) {
  BKG_ARG(n >= 0 && d > 0, "bkg_scale_add_rows_out_f32: bad sizes");
  if (n == 0) return 0;
  hipLaunchKernelGGL(scale_add_rows_out_kernel, dim3(ceil_div((int64_t)n * d, 256)), dim3(256), 0, (hipStream_t)stream, base, ldb,
                     src, lds, n, d, gloss, coef, out, ldo);
  return check_launch("bkg_scale_add_rows_out_f32");
}
