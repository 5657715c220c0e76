// Tall-skinny "activation x weight" products of the bf16 mode:  C[M, N] = epi(alpha * A[M, K] @ Wt[N, K]^T)
// with M ~ 1e5..1e6 rows and N, K in {256, 768}: GCNConv.lin / the GGD heads / RGCNConv's root and per-relation
// weights, forward and input-gradient (reference biomedkg/model/encoder.py:50-59,153-162, gcl.py:86-90).
//
// These products are bound by streaming A once; the LDS-staged tile of gemm_tile_bf16.h runs them with the loads
// fully exposed (measured: t = t_compute + t_loadA + t_loadB, no overlap -- two barriers per 64-deep K step and only
// two resident blocks per CU).  This kernel removes every barrier from the K loop:
//   * the weight slab Wt[n0 : n0+128, k0 : k0+256] (bf16, prepared once per step by bkg_weight_to_bf16) is copied
//     into LDS once and stays there while the block walks over its row tiles (K <= 256: for the whole launch);
//   * each wave streams ITS OWN 64 rows of A straight from HBM into MFMA operand registers (lane = row, 8 consecutive
//     k: 32 B of fp32 or 16 B of bf16 per lane and step), four 16-deep steps ahead of the MFMAs, so the only waits
//     in the loop are counted vmcnt waits of that wave;
//   * operands are swapped (D = W_frag x A_frag), which leaves each lane with 4 CONSECUTIVE output columns of one
//     row per accumulator quad: C is written as 16 B/lane (fp32) or 8 B/lane (bf16) row pieces straight from the
//     accumulators -- no LDS transpose, no epilogue barrier.
// Rows may be grouped (RGCNConv: rows of relation g use Wt + g * stride_w); tiles never straddle groups.
#include "common.h"
#include "gemm_epilogue.h"

namespace bkg {

typedef float xw_acc_t __attribute__((ext_vector_type(16)));
typedef __bf16 xw_bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 xw_bf16x4_t __attribute__((ext_vector_type(4)));

constexpr int XW_BN = 128;                 // output columns per block: 4 MFMA tiles
constexpr int XW_TN = XW_BN / 32;
constexpr int XW_WAVES = 4;
constexpr int XW_TM = 2;                   // 32-row MFMA tiles per wave
constexpr int XW_BM = XW_WAVES * XW_TM * 32;   // 256 rows per block
constexpr int XW_KS = 256;                 // K slab kept in LDS
constexpr int XW_LDB = XW_KS + 8;          // bf16 per LDS row: 528 B = 33 x 16 B -> conflict-free ds_read_b128
// A prefetch depth in 16-deep steps: 64 B per lane and row tile in flight for either element type
template <typename AT> constexpr int xw_pd() { return sizeof(AT) == 4 ? 2 : 4; }

template <typename AT> struct XwFrag;
template <> struct XwFrag<float> {
  float4 lo, hi;
  __device__ __forceinline__ void load(const float* p) {
    lo = *reinterpret_cast<const float4*>(p);
    hi = *reinterpret_cast<const float4*>(p + 4);
  }
  __device__ __forceinline__ xw_bf16x8_t get() const {
    xw_bf16x8_t v;
    v[0] = (__bf16)lo.x; v[1] = (__bf16)lo.y; v[2] = (__bf16)lo.z; v[3] = (__bf16)lo.w;
    v[4] = (__bf16)hi.x; v[5] = (__bf16)hi.y; v[6] = (__bf16)hi.z; v[7] = (__bf16)hi.w;
    return v;
  }
};
template <> struct XwFrag<__bf16> {
  xw_bf16x8_t v;
  __device__ __forceinline__ void load(const __bf16* p) { v = *reinterpret_cast<const xw_bf16x8_t*>(p); }
  __device__ __forceinline__ xw_bf16x8_t get() const { return v; }
};

struct XwArgs {
  const void* A; int64_t lda;
  const int32_t* a_rows;   // nullable: output row m reads A row a_rows[m] (RGCNConv backward: dY[s] = dOut[dst_s], never stored)
  const __bf16* Wt; int64_t ldw, stride_w;
  void* C; int64_t ldc;
  const int32_t* group_ptr; int G;
  int M, N, K;
  int persist;   // row-tile stride of the persistent loop (plain mode)
  // ReDAF epilogue (gemm_xw_kernel<..., REDAF>): C is the fp32 [M / 2, N] fused output, rows 2n / 2n + 1 of the product are
  // node n's two modality tokens
  const float* rd_scale;       // [2][N]  modal_weights * zeta
  const uint8_t* rd_keep;      // nullable [M][N] dropout keep mask (replayed draws); else the in-kernel keep hash (hash_keep4)
  float rd_p; uint64_t rd_seed; uint32_t rd_stream; int64_t rd_row0;
  const uint32_t* rd_epoch;
};

// Two adjacent accumulator quads of one lane (columns 8q + 4lh + 0..3 and 8(q+1) + 4lh + 0..3 of a 16-column group).
// fp32 C: two 16-byte stores.  bf16 C: the lane pair (lh = 0 / 1, same row) swaps one packed quad so that each lane owns
// 8 CONSECUTIVE columns and writes one 16-byte piece -- 8-byte pieces cost 17-19 us per launch at 130k x 256.
__device__ __forceinline__ void xw_store_pair(float* base, int lh, const float4& v0, const float4& v1) {
  *reinterpret_cast<float4*>(base + 4 * lh) = v0;
  *reinterpret_cast<float4*>(base + 8 + 4 * lh) = v1;
}
__device__ __forceinline__ uint2 xw_pack4(const float4& v) {
  xw_bf16x4_t h;
  h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
  return __builtin_bit_cast(uint2, h);
}
__device__ __forceinline__ uint4 xw_pair8(int lh, const float4& v0, const float4& v1) {
  const uint2 p0 = xw_pack4(v0), p1 = xw_pack4(v1);
  const uint2 give = lh ? p0 : p1;                       // lh 0 keeps columns 0..3, hands over 8..11; lh 1 the reverse
  uint2 got;
  got.x = __shfl_xor(give.x, 32, 64);
  got.y = __shfl_xor(give.y, 32, 64);
  return lh ? make_uint4(got.x, got.y, p1.x, p1.y) : make_uint4(p0.x, p0.y, got.x, got.y);
}
__device__ __forceinline__ void xw_store_pair(__bf16* base, int lh, const float4& v0, const float4& v1) {
  *reinterpret_cast<uint4*>(base + 8 * lh) = xw_pair8(lh, v0, v1);   // lh 0: columns 0..7, lh 1: columns 8..15
}

// ---- the ReLU / dropout backward gate of the PREVIOUS layer and that layer's bias gradient in the epilogue of the
// product that makes its output gradient (gemm_xw8_kernel<__bf16, true>): C = bf16(gate > 0 ? bf16(A W) * scale : 0),
// partial[block][c] = sum over the block's rows of C[:, c].  The values are those of the two-pass form (product
// rounded to bf16, gate_cast_colsum over it); the column sums add the same values in another order.
struct XwGate {
  const __bf16* gate;    // [M, ldg] bf16: the saved activation the gradient flows back through
  int64_t ldg;
  float scale;           // 1 / (1 - p) of the dropout behind the ReLU (1: none)
  float* partial;        // [gridDim.x][256]
};
typedef unsigned xw_u32x4_t __attribute__((ext_vector_type(4)));
// hand-issued: a load hipcc knows about makes it drain the LDS-DMA ring (vmcnt(0)) before the first use
__device__ __forceinline__ xw_u32x4_t xw_gload16(const void* p) {
  xw_u32x4_t v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// one packed pair of bf16 values through the gate; adds the two results (as stored) to lo / hi
__device__ __forceinline__ unsigned xw_gate2(unsigned x, unsigned gt, float scale, float& lo, float& hi) {
  const float x0 = __uint_as_float(x << 16), x1 = __uint_as_float(x & 0xffff0000u);
  const float g0 = __uint_as_float(gt << 16), g1 = __uint_as_float(gt & 0xffff0000u);
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 o;
  o[0] = (__bf16)(g0 > 0.f ? x0 * scale : 0.f);
  o[1] = (__bf16)(g1 > 0.f ? x1 * scale : 0.f);
  lo += (float)o[0];
  hi += (float)o[1];
  return __builtin_bit_cast(unsigned, o);
}

// one 16-deep step: 4 weight fragments from LDS x 2 row fragments -> 8 MFMAs
__device__ __forceinline__ void xw_step(xw_acc_t (&acc)[XW_TN][XW_TM], const __bf16 (*Bs)[XW_LDB], int l31, int lh, int ks,
                                        const xw_bf16x8_t (&a)[XW_TM]) {
  xw_bf16x8_t w[XW_TN];
#pragma unroll
  for (int i = 0; i < XW_TN; ++i) w[i] = *reinterpret_cast<const xw_bf16x8_t*>(&Bs[i * 32 + l31][ks * 16 + lh * 8]);
#pragma unroll
  for (int i = 0; i < XW_TN; ++i)
#pragma unroll
    for (int j = 0; j < XW_TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[i], a[j], acc[i][j], 0, 0, 0);
}

template <typename AT, typename CT, bool GROUPED, bool REDAF = false>
__global__ __launch_bounds__(256, 2) void gemm_xw_kernel(XwArgs g, GemmEpilogue ep) {
  __shared__ __attribute__((aligned(16))) __bf16 Bs[XW_BN][XW_LDB];
  __shared__ int s_g, s_m0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
  const int n_tiles = (g.N + XW_BN - 1) / XW_BN;
  const XcdSlot slot = xcd_slot(blockIdx.x, n_tiles);
  const int n0 = slot.inner * XW_BN;
  const int n_slabs = (g.K + XW_KS - 1) / XW_KS;
  const AT* __restrict__ A = reinterpret_cast<const AT*>(g.A);
  CT* __restrict__ C = reinterpret_cast<CT*>(g.C);

  int mt = slot.outer, m_tiles, m_lo = 0, m_hi = g.M, step = g.persist;
  const __bf16* Wg = g.Wt;
  if constexpr (GROUPED) {
    if (threadIdx.x == 0) {
      int t = slot.outer, gi = 0, found = -1, m0 = 0;
      for (; gi < g.G; ++gi) {
        const int rows = g.group_ptr[gi + 1] - g.group_ptr[gi];
        const int nt = (rows + XW_BM - 1) / XW_BM;
        if (t < nt) { found = gi; m0 = g.group_ptr[gi] + t * XW_BM; break; }
        t -= nt;
      }
      s_g = found; s_m0 = m0;
    }
    __syncthreads();
    if (s_g < 0) return;
    Wg = g.Wt + (int64_t)s_g * g.stride_w;
    m_lo = s_m0; m_hi = g.group_ptr[s_g + 1];
    mt = 0; m_tiles = 1; step = 1;
  } else {
    m_tiles = (g.M + XW_BM - 1) / XW_BM;
  }

  int have_slab = -1;
  for (; mt < m_tiles; mt += step) {
    const int m_base = (GROUPED ? m_lo : mt * XW_BM) + wave * (XW_TM * 32);
    // this lane's two source rows (clamped: out-of-range rows are computed on valid memory and never stored)
    const AT* arow[XW_TM];
#pragma unroll
    for (int j = 0; j < XW_TM; ++j) {
      const int r = min(m_base + j * 32 + l31, m_hi - 1);
      arow[j] = A + (int64_t)(g.a_rows ? g.a_rows[r] : r) * g.lda + lh * 8;
    }

    xw_acc_t acc[XW_TN][XW_TM];
#pragma unroll
    for (int i = 0; i < XW_TN; ++i)
#pragma unroll
      for (int j = 0; j < XW_TM; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int s = 0; s < n_slabs; ++s) {
      const int k0 = s * XW_KS;
      if (have_slab != s) {   // block-uniform
        __syncthreads();      // every wave is done with the previous slab
        {
          // 16-byte pieces, 32 per row: one wave-instruction copies two whole 512-byte rows of the slab.  All 16
          // loads of a thread are issued before the first LDS store (a load / store pair per iteration makes the
          // compiler wait for each load in turn: 16 serial L2 round trips per block).
          const int pc = threadIdx.x % (XW_KS / 8), r_in = threadIdx.x / (XW_KS / 8);
          const __bf16* src = Wg + k0 + pc * 8;
          constexpr int NP = (XW_BN * XW_KS / 8) / 256, RS = 256 / (XW_KS / 8);
          constexpr int NB = sizeof(AT) == 4 ? 4 : 2;   // batches of 4 / 8 loads (the accumulators are live: registers)
#pragma unroll
          for (int h = 0; h < NB; ++h) {
            xw_bf16x8_t v[NP / NB];
#pragma unroll
            for (int i = 0; i < NP / NB; ++i)
              v[i] = *reinterpret_cast<const xw_bf16x8_t*>(src + (int64_t)min(n0 + r_in + (h * (NP / NB) + i) * RS, g.N - 1) * g.ldw);
#pragma unroll
            for (int i = 0; i < NP / NB; ++i)
              *reinterpret_cast<xw_bf16x8_t*>(&Bs[r_in + (h * (NP / NB) + i) * RS][pc * 8]) = v[i];
          }
        }
        __syncthreads();
        have_slab = s;
      }
      constexpr int PD = xw_pd<AT>(), NSTEP = XW_KS / 16;
      XwFrag<AT> ring[PD][XW_TM];
      {   // a rolled loop of PD-step groups: counted vmcnt waits only, no barrier (K % 256 == 0: whole slabs)
#pragma unroll
        for (int p = 0; p < PD; ++p)
#pragma unroll
          for (int j = 0; j < XW_TM; ++j) ring[p][j].load(arow[j] + k0 + p * 16);
#pragma unroll 1
        for (int kb = 0; kb < NSTEP - PD; kb += PD) {
#pragma unroll
          for (int p = 0; p < PD; ++p) {
            xw_bf16x8_t a[XW_TM];
#pragma unroll
            for (int j = 0; j < XW_TM; ++j) {
              a[j] = ring[p][j].get();
              ring[p][j].load(arow[j] + k0 + (kb + PD + p) * 16);
            }
            xw_step(acc, Bs, l31, lh, kb + p, a);
          }
        }
#pragma unroll
        for (int p = 0; p < PD; ++p) {
          xw_bf16x8_t a[XW_TM];
#pragma unroll
          for (int j = 0; j < XW_TM; ++j) a[j] = ring[p][j].get();
          xw_step(acc, Bs, l31, lh, NSTEP - PD + p, a);
        }
      }
    }
    if (n_slabs > 1) have_slab = n_slabs - 1;

    // epilogue: acc[i][j][4q..4q+3] = C[m_base + 32 j + l31][n0 + 32 i + 8 q + 4 lh + 0..3]; one row pointer per j,
    // every piece at a compile-time offset from it (N % 32 == 0: a 32-column tile is in range or not as a whole)
    if constexpr (REDAF) {
      // ReDAF.forward behind the product (reference biomedkg/utils/fusion.py:49-90): t = relu(x W^T + b) is this lane's
      // accumulator row; out[n] = mean_m relu(dropout(t[2n + m] * scale[m])) -- the two modality rows of a node are the
      // lanes l31 and l31 ^ 1 (lane = row in this kernel), so the [2n, D] transform never leaves the registers.  The
      // keep hash (common.h hash_keep4) is indexed exactly as bkg_redaf_combine_f32 indexes it: the same masks, the same bits out.
      const int mrow = l31 & 1;
      const float ks = (g.rd_keep || g.rd_p > 0.f) ? 1.0f / (1.0f - g.rd_p) : 1.0f;
      const uint32_t thresh = (uint32_t)fminf(g.rd_p * 4294967296.0f, 4294967295.0f);
      const uint32_t sid = with_epoch(g.rd_stream, g.rd_epoch);
#pragma unroll
      for (int j = 0; j < XW_TM; ++j) {
        const int row = m_base + j * 32 + l31;
        const bool live = row < m_hi;
        float* __restrict__ orow = reinterpret_cast<float*>(g.C) + (int64_t)(row >> 1) * g.ldc + n0;
#pragma unroll
        for (int i = 0; i < XW_TN; ++i) {
          if (n0 + i * 32 < g.N) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int off = i * 32 + q * 8 + lh * 4, col = n0 + off;
              float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
              const float4 b = ep.bias ? *reinterpret_cast<const float4*>(ep.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
              const float4 sc = *reinterpret_cast<const float4*>(g.rd_scale + (int64_t)mrow * g.N + col);
              v.x = fmaxf(v.x + b.x, 0.f) * sc.x; v.y = fmaxf(v.y + b.y, 0.f) * sc.y;
              v.z = fmaxf(v.z + b.z, 0.f) * sc.z; v.w = fmaxf(v.w + b.w, 0.f) * sc.w;
              if (g.rd_keep) {
                uchar4 k = make_uchar4(0, 0, 0, 0);
                if (live) k = *reinterpret_cast<const uchar4*>(g.rd_keep + (int64_t)row * g.N + col);
                v.x = k.x ? v.x * ks : 0.f; v.y = k.y ? v.y * ks : 0.f; v.z = k.z ? v.z * ks : 0.f; v.w = k.w ? v.w * ks : 0.f;
              } else if (g.rd_p > 0.f) {
                const uint4 r = hash_keep4(g.rd_seed, (uint64_t)(g.rd_row0 + row) * g.N + col, sid);
                v.x = r.x >= thresh ? v.x * ks : 0.f; v.y = r.y >= thresh ? v.y * ks : 0.f;
                v.z = r.z >= thresh ? v.z * ks : 0.f; v.w = r.w >= thresh ? v.w * ks : 0.f;
              }
              v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
              // modality 0 first, as the two-kernel form adds them (y = 0 + f(t_0) + f(t_1)): bit-equal
              float4 o;
              o.x = __shfl_xor(v.x, 1, 64); o.y = __shfl_xor(v.y, 1, 64); o.z = __shfl_xor(v.z, 1, 64); o.w = __shfl_xor(v.w, 1, 64);
              if (live && mrow == 0)
                *reinterpret_cast<float4*>(orow + off) = make_float4((v.x + o.x) * 0.5f, (v.y + o.y) * 0.5f, (v.z + o.z) * 0.5f,
                                                                     (v.w + o.w) * 0.5f);
            }
          }
        }
      }
      continue;
    }
    const bool plain = !ep.bias && ep.act == 0 && !ep.gate && ep.beta == 0.f && ep.alpha == 1.f;
#pragma unroll
    for (int j = 0; j < XW_TM; ++j) {
      const int row = m_base + j * 32 + l31;
      if (row < m_hi) {
        CT* __restrict__ drow = C + (int64_t)row * g.ldc + n0;
        if (plain) {
#pragma unroll
          for (int i = 0; i < XW_TN; ++i) {
            if (n0 + i * 32 < g.N) {
#pragma unroll
              for (int qp = 0; qp < 2; ++qp)
                xw_store_pair(drow + i * 32 + qp * 16, lh,
                              make_float4(acc[i][j][8 * qp], acc[i][j][8 * qp + 1], acc[i][j][8 * qp + 2], acc[i][j][8 * qp + 3]),
                              make_float4(acc[i][j][8 * qp + 4], acc[i][j][8 * qp + 5], acc[i][j][8 * qp + 6],
                                          acc[i][j][8 * qp + 7]));
            }
          }
        } else {
          const float* __restrict__ brow = ep.bias ? ep.bias + n0 + lh * 4 : nullptr;
          const float* __restrict__ grow = ep.gate ? ep.gate + (int64_t)row * ep.ldg + n0 + lh * 4 : nullptr;
#pragma unroll
          for (int i = 0; i < XW_TN; ++i) {
            if (n0 + i * 32 < g.N) {
#pragma unroll
              for (int qp = 0; qp < 2; ++qp) {
                float4 vv[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                  const int q = 2 * qp + h, off = i * 32 + q * 8;
                  float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                  const float4 b = brow ? *reinterpret_cast<const float4*>(brow + off) : make_float4(0.f, 0.f, 0.f, 0.f);
                  v.x = ep.alpha * v.x + b.x; v.y = ep.alpha * v.y + b.y; v.z = ep.alpha * v.z + b.z; v.w = ep.alpha * v.w + b.w;
                  if (ep.act == 1) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                  } else if (ep.act == 2) {
                    v.x = v.x > 0.f ? v.x : expm1f(v.x); v.y = v.y > 0.f ? v.y : expm1f(v.y);
                    v.z = v.z > 0.f ? v.z : expm1f(v.z); v.w = v.w > 0.f ? v.w : expm1f(v.w);
                  }
                  if (grow) {
                    const float4 gt = *reinterpret_cast<const float4*>(grow + off);
                    v.x = gt.x > 0.f ? v.x * ep.gate_scale : 0.f; v.y = gt.y > 0.f ? v.y * ep.gate_scale : 0.f;
                    v.z = gt.z > 0.f ? v.z * ep.gate_scale : 0.f; v.w = gt.w > 0.f ? v.w * ep.gate_scale : 0.f;
                  }
                  if constexpr (sizeof(CT) == 4) {
                    if (ep.beta != 0.f) {
                      const float4 o = *reinterpret_cast<const float4*>(drow + off + lh * 4);
                      v.x = fmaf(ep.beta, o.x, v.x); v.y = fmaf(ep.beta, o.y, v.y);
                      v.z = fmaf(ep.beta, o.z, v.z); v.w = fmaf(ep.beta, o.w, v.w);
                    }
                  }
                  vv[h] = v;
                }
                xw_store_pair(drow + i * 32 + qp * 16, lh, vv[0], vv[1]);
              }
            }
          }
        }
      }
    }
  }
}

// ---- K = 256, N = 256, bf16 rows: the whole weight in REGISTERS, A through LDS in full lines -------------------------
// The kernel above reads A in MFMA-fragment shape (32 rows x 32 B per wave-instruction: every 128-byte line is asked
// for four times) and twice (once per 128-column slab): 3.6 TB/s of operand + result bytes.  Here a block is 8 waves
// and covers all 256 output columns: wave w keeps W[32 w : 32 w + 32, 0 : 256] as 16 MFMA operands (64 VGPRs) for its
// whole life, so LDS holds nothing but A: 64-row tiles (32 KB) copied by LDS-DMA, two whole 512-byte rows per
// wave-instruction, into a ring of four slots -- up to three tiles in flight per CU across one raw barrier per tile,
// retired by counted vmcnt waits -- and read by every wave as b128 fragments.  A 512-byte row stride would put the 32
// rows of a fragment read on the same banks: slot c' of row r holds chunk c' ^ (r & 31), applied to the DMA's SOURCE
// address (its destination is lane-linear).  Per tile and CU: 256 KB of LDS reads and 2 x 1024 MFMA cycles per SIMD --
// both 2048 clocks for 32 KB of A, three times the chip's HBM share.
constexpr int X8_BM = 64, X8_NS = 4, X8_K = 256, X8_N = 256;
constexpr int X8_STAGE = X8_BM * X8_K * 2;

template <typename CT, bool GATED = false>
__global__ __launch_bounds__(512) void gemm_xw8_kernel(XwArgs g, GemmEpilogue ep, XwGate gt) {
  static_assert(!GATED || sizeof(CT) == 2, "the gated product writes a bf16 table");
  __shared__ __attribute__((aligned(1024))) unsigned char ring[X8_NS * X8_STAGE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lh = lane >> 5;
  const int n0 = wave * 32;
  const __bf16* __restrict__ A = reinterpret_cast<const __bf16*>(g.A);
  CT* __restrict__ C = reinterpret_cast<CT*>(g.C);

  const int m_tiles = (g.M + X8_BM - 1) / X8_BM, stride = gridDim.x;
  // DMA pieces of this lane: rows wave * 8 + 2 j + (lane >> 5) of a tile, slot lane & 31
  int64_t src_off[4];
  int drow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    drow[j] = wave * 8 + 2 * j + (lane >> 5);
    src_off[j] = (int64_t)(((lane & 31) ^ (drow[j] & 31)) * 8);
  }
  int t_issue = blockIdx.x;
  auto issue = [&](int slot) {
    unsigned char* base = ring + slot * X8_STAGE + wave * (8 * 512);
    const int m0 = t_issue * X8_BM;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = max(0, min(m0 + drow[j], g.M - 1));   // past the end: valid memory, never stored
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (int64_t)r * g.lda + src_off[j]),
                                       (__attribute__((address_space(3))) void*)(base + j * 1024), 16, 0, 0);
    }
    t_issue += stride;
  };
#pragma unroll
  for (int s = 0; s < X8_NS - 1; ++s) issue(s);
  // the weights and the bias travel together with the first tiles (hipcc waits for everything in flight at their first
  // use, i.e. once, before the first tile); from then on the loop issues nothing but DMAs and stores
  xw_bf16x8_t w[X8_K / 16];
#pragma unroll
  for (int ks = 0; ks < X8_K / 16; ++ks)
    w[ks] = *reinterpret_cast<const xw_bf16x8_t*>(g.Wt + (int64_t)(n0 + l31) * g.ldw + ks * 16 + lh * 8);
  // bias of this lane's columns n0 + 8 q + 4 lh + 0..3 (an ordinary load inside the loop would drain the DMAs)
  float4 bias[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    bias[q] = ep.bias ? *reinterpret_cast<const float4*>(ep.bias + n0 + 8 * q + 4 * lh) : make_float4(0.f, 0.f, 0.f, 0.f);


  const int p16 = (lh ^ l31) * 16;
  int slot = 0, it = 0;
  float cs[2][8];       // gated: column sums of the lane's 2 x 8 columns over its rows
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int c = 0; c < 8; ++c) cs[p][c] = 0.f;
  for (int t = blockIdx.x; t < m_tiles; t += stride) {
    // This wave's DMAs of tile t have landed.  vmcnt retires in issue order and counts the result stores too: behind
    // tile t's DMAs this wave has issued the DMAs of two more tiles and the stores of up to three finished ones (a
    // partial tile -- fewer stores -- can only be a block's last).  Counting the stores keeps both younger tiles in
    // flight; a wait for "at most 8" would retire the older of them as well.
    constexpr int SPT = sizeof(CT) == 2 ? 4 : 8;   // 16-byte stores per wave and tile
    // (gated: every iteration also issues GL gate loads, BEFORE its DMAs -- two iterations' worth sit behind tile t's)
    constexpr int GL = GATED ? 4 : 0;
    if (it >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + 3 * SPT + 2 * GL) : "memory");
    else if (it == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + 2 * SPT + 2 * GL) : "memory");
    else if (it == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 + SPT + GL) : "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    ++it;
    __builtin_amdgcn_s_barrier();   // ... everyone else's too, and every wave is done with the slot refilled next
    // this tile's gate pieces: the 16 bytes beside each 16-byte piece of C this lane stores (rows past the end: row M - 1)
    xw_u32x4_t gq[2][2];
    if constexpr (GATED) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int64_t row = min(t * X8_BM + j * 32 + l31, g.M - 1);
#pragma unroll
        for (int p = 0; p < 2; ++p) gq[j][p] = xw_gload16(gt.gate + row * gt.ldg + n0 + 16 * p + 8 * lh);
      }
    }
    issue(slot == 0 ? X8_NS - 1 : slot - 1);
    const unsigned char* tile = ring + slot * X8_STAGE + l31 * 512;
    xw_acc_t acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < X8_K / 16; ++ks) {
      const int off = (ks * 32) ^ p16;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const xw_bf16x8_t a = *reinterpret_cast<const xw_bf16x8_t*>(tile + j * (32 * 512) + off);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ks], a, acc[j], 0, 0, 0);
      }
    }
    // behind the gate loads this wave has issued one tile's DMAs, nothing else
    if constexpr (GATED)
      asm volatile("s_waitcnt vmcnt(4)" : "+v"(gq[0][0]), "+v"(gq[0][1]), "+v"(gq[1][0]), "+v"(gq[1][1])::"memory");
    // acc[j][4q..4q+3] = C[t * 64 + 32 j + l31][n0 + 8 q + 4 lh + 0..3]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = t * X8_BM + j * 32 + l31;
      if (row < g.M) {
        CT* __restrict__ drow_p = C + (int64_t)row * g.ldc + n0;
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if constexpr (GATED) {      // a plain product under the gate
            v[q] = make_float4(acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
            continue;
          }
          v[q] = make_float4(ep.alpha * acc[j][4 * q] + bias[q].x, ep.alpha * acc[j][4 * q + 1] + bias[q].y,
                             ep.alpha * acc[j][4 * q + 2] + bias[q].z, ep.alpha * acc[j][4 * q + 3] + bias[q].w);
          if (ep.act == 1) {
            v[q].x = fmaxf(v[q].x, 0.f); v[q].y = fmaxf(v[q].y, 0.f); v[q].z = fmaxf(v[q].z, 0.f); v[q].w = fmaxf(v[q].w, 0.f);
          } else if (ep.act == 2) {
            v[q].x = v[q].x > 0.f ? v[q].x : expm1f(v[q].x); v[q].y = v[q].y > 0.f ? v[q].y : expm1f(v[q].y);
            v[q].z = v[q].z > 0.f ? v[q].z : expm1f(v[q].z); v[q].w = v[q].w > 0.f ? v[q].w : expm1f(v[q].w);
          }
        }
        if constexpr (GATED) {
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const uint4 x = xw_pair8(lh, v[2 * p], v[2 * p + 1]);
            uint4 o;
            o.x = xw_gate2(x.x, gq[j][p][0], gt.scale, cs[p][0], cs[p][1]);
            o.y = xw_gate2(x.y, gq[j][p][1], gt.scale, cs[p][2], cs[p][3]);
            o.z = xw_gate2(x.z, gq[j][p][2], gt.scale, cs[p][4], cs[p][5]);
            o.w = xw_gate2(x.w, gq[j][p][3], gt.scale, cs[p][6], cs[p][7]);
            *reinterpret_cast<uint4*>(drow_p + 16 * p + 8 * lh) = o;
          }
        } else {
          xw_store_pair(drow_p, lh, v[0], v[1]);
          xw_store_pair(drow_p + 16, lh, v[2], v[3]);
        }
      }
    }
    slot = slot + 1 == X8_NS ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the look-ahead DMAs past the last tile
  if constexpr (GATED) {
    // this block's column sums: lanes of one half hold the same 16 columns over different rows
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float v = cs[p][c];
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) v += __shfl_xor(v, m, 64);
        cs[p][c] = v;
      }
    if (l31 == 0) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        float* dst = gt.partial + (int64_t)blockIdx.x * X8_N + n0 + 16 * p + 8 * lh;
        *reinterpret_cast<float4*>(dst) = make_float4(cs[p][0], cs[p][1], cs[p][2], cs[p][3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(cs[p][4], cs[p][5], cs[p][6], cs[p][7]);
      }
    }
  }
}

// The same block for ROW GROUPS (RGCNConv: rows of relation g use Wt + g * stride_w) and gathered rows (a_rows).  Tiles
// never straddle a group; a block walks a CONTIGUOUS range of the concatenated tile list, so it changes group -- and
// reloads its 64 weight registers, draining the DMAs once -- two or three times in its life, not per tile.  Partial
// tiles end every group here, so the number of stores behind a tile's DMAs is not a constant: the waits count DMAs only.
// Source rows of a gathered product come through scalar loads (two rows per DMA; an ordinary load would make hipcc
// drain the ring every tile).
template <typename CT>
__global__ __launch_bounds__(512) void gemm_xw8_grouped_kernel(XwArgs g, GemmEpilogue ep, const int32_t* __restrict__ gp,
                                                               const int32_t* __restrict__ a_rows) {
  // (gp / a_rows as __restrict__ PARAMETERS: read through the argument struct hipcc cannot tell them from C and turns
  // every group_ptr / index read into a vector load + vmcnt(0))
  __shared__ __attribute__((aligned(1024))) unsigned char ring[X8_NS * X8_STAGE];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l31 = lane & 31, lh = lane >> 5;
  const int n0 = wave * 32;
  const __bf16* __restrict__ A = reinterpret_cast<const __bf16*>(g.A);
  CT* __restrict__ C = reinterpret_cast<CT*>(g.C);
  // The group table lives in the lanes (G <= 63): lane q holds group q's row range, tile count and first tile index
  // (one vector load and a wave prefix sum, before any DMA is in flight); finding a tile's group is a ballot.
  const int my_lo = gp[min(lane, g.G)], my_hi = gp[min(lane + 1, g.G)];
  const int my_nt = lane < g.G ? (my_hi - my_lo + X8_BM - 1) / X8_BM : 0;
  int incl = my_nt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(incl, d, 64);
    if (lane >= d) incl += v;
  }
  const int my_first = incl - my_nt;
  const int total = __builtin_amdgcn_readlane(incl, 63);
  // The blocks' ranges are cut by COST, not by tile count: entering a group costs a weight reload and a drained DMA
  // ring -- about two tiles' worth -- and the relations' row counts are skewed (a range of seventeen one-tile groups
  // would otherwise be the launch's straggler).  Cost of group q in quarter tiles: 8 + 4 nt_q; tile i of it starts at
  // pc_q + 8 + 4 i, and a block owns the tiles whose start lies in its share of the total.
  constexpr int CU = 4, CR = 8;
  const int my_cost = my_nt > 0 ? CR + CU * my_nt : 0;
  int cincl = my_cost;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(cincl, d, 64);
    if (lane >= d) cincl += v;
  }
  const int my_pc = cincl - my_cost;
  const int total_cost = __builtin_amdgcn_readlane(cincl, 63);
  auto tile_at = [&](int c) {             // index of the first tile whose start cost is >= c (total: none)
    const unsigned long long m = __ballot(my_nt > 0 && my_pc <= c);
    if (m == 0ull) return 0;
    const int q = 63 - __builtin_clzll(m);
    const int pc = __builtin_amdgcn_readlane(my_pc, q), nt = __builtin_amdgcn_readlane(my_nt, q);
    const int i = max(0, (c - pc - CR + CU - 1) / CU);
    return __builtin_amdgcn_readlane(my_first, q) + min(i, nt);
  };
  const int t_lo = tile_at((int)((int64_t)total_cost * blockIdx.x / gridDim.x));
  const int t_hi = blockIdx.x + 1 == gridDim.x ? total : tile_at((int)((int64_t)total_cost * (blockIdx.x + 1) / gridDim.x));
  if (t_lo >= t_hi) return;
  struct Tile { int grp, m0, m_hi; };
  auto locate = [&](int t) {              // t < total: the last non-empty group that starts at or before tile t
    const unsigned long long m = __ballot(my_nt > 0 && my_first <= t);
    const int q = 63 - __builtin_clzll(m);
    const int lo = __builtin_amdgcn_readlane(my_lo, q), first = __builtin_amdgcn_readlane(my_first, q);
    return Tile{q, lo + (t - first) * X8_BM, __builtin_amdgcn_readlane(my_hi, q)};
  };

  int64_t src_off[4];
  int drow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    drow[j] = wave * 8 + 2 * j + (lane >> 5);
    src_off[j] = (int64_t)(((lane & 31) ^ (drow[j] & 31)) * 8);
  }
  int t_issue = t_lo;
  auto issue = [&](int slot) {
    if (t_issue < t_hi) {
      const Tile ti = locate(t_issue);
      unsigned char* base = ring + slot * X8_STAGE + wave * (8 * 512);
      const int m0 = ti.m0, m_last = ti.m_hi - 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int r;
        if (a_rows) {
          const int i0 = min(m0 + wave * 8 + 2 * j, m_last), i1 = min(m0 + wave * 8 + 2 * j + 1, m_last);
          const int r0 = a_rows[i0], r1 = a_rows[i1];
          r = (lane >> 5) ? r1 : r0;
        } else {
          r = min(m0 + drow[j], m_last);   // past the end of the group: valid memory, never stored
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (int64_t)r * g.lda + src_off[j]),
                                         (__attribute__((address_space(3))) void*)(base + j * 1024), 16, 0, 0);
      }
    } else {
      // past this block's range: the waits below count four DMAs per tile -- issue them on memory that is surely there
      unsigned char* base = ring + slot * X8_STAGE + wave * (8 * 512);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g.Wt + (lane & 31) * 8),
                                         (__attribute__((address_space(3))) void*)(base + j * 1024), 16, 0, 0);
    }
    ++t_issue;
  };
#pragma unroll
  for (int s = 0; s < X8_NS - 1; ++s) issue(s);

  float4 bias[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    bias[q] = ep.bias ? *reinterpret_cast<const float4*>(ep.bias + n0 + 8 * q + 4 * lh) : make_float4(0.f, 0.f, 0.f, 0.f);
  xw_bf16x8_t w[X8_K / 16];
  int g_have = -1;

  const int p16 = (lh ^ l31) * 16;
  int slot = 0;
  for (int t = t_lo; t < t_hi; ++t) {
    const Tile tc = locate(t);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((X8_NS - 2) * 4) : "memory");   // this tile's DMAs (and every older store)
    __builtin_amdgcn_s_barrier();
    issue(slot == 0 ? X8_NS - 1 : slot - 1);
    if (tc.grp != g_have) {   // block-uniform; hipcc waits for these loads -- and with them for the DMAs in flight
      const __bf16* Wg = g.Wt + (int64_t)tc.grp * g.stride_w;
#pragma unroll
      for (int ks = 0; ks < X8_K / 16; ++ks)
        w[ks] = *reinterpret_cast<const xw_bf16x8_t*>(Wg + (int64_t)(n0 + l31) * g.ldw + ks * 16 + lh * 8);
      g_have = tc.grp;
    }
    const int m0 = tc.m0, m_hi = tc.m_hi;
    const unsigned char* tile = ring + slot * X8_STAGE + l31 * 512;
    xw_acc_t acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < X8_K / 16; ++ks) {
      const int off = (ks * 32) ^ p16;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const xw_bf16x8_t a = *reinterpret_cast<const xw_bf16x8_t*>(tile + j * (32 * 512) + off);
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ks], a, acc[j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = m0 + j * 32 + l31;
      if (row < m_hi) {
        CT* __restrict__ drow_p = C + (int64_t)row * g.ldc + n0;
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[q] = make_float4(ep.alpha * acc[j][4 * q] + bias[q].x, ep.alpha * acc[j][4 * q + 1] + bias[q].y,
                             ep.alpha * acc[j][4 * q + 2] + bias[q].z, ep.alpha * acc[j][4 * q + 3] + bias[q].w);
          if (ep.act == 1) {
            v[q].x = fmaxf(v[q].x, 0.f); v[q].y = fmaxf(v[q].y, 0.f); v[q].z = fmaxf(v[q].z, 0.f); v[q].w = fmaxf(v[q].w, 0.f);
          } else if (ep.act == 2) {
            v[q].x = v[q].x > 0.f ? v[q].x : expm1f(v[q].x); v[q].y = v[q].y > 0.f ? v[q].y : expm1f(v[q].y);
            v[q].z = v[q].z > 0.f ? v[q].z : expm1f(v[q].z); v[q].w = v[q].w > 0.f ? v[q].w : expm1f(v[q].w);
          }
        }
        xw_store_pair(drow_p, lh, v[0], v[1]);
        xw_store_pair(drow_p + 16, lh, v[2], v[3]);
      }
    }
    slot = slot + 1 == X8_NS ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- K = 256 KC (KC = 2, 3), N a multiple of 256, bf16 rows: a 256-column WEIGHT SLAB in registers for the block's life ---
// The K = 768 products (ReDAF, AttentionFusion's two products, the first GCN layer; reference biomedkg/utils/fusion.py:17-31,
// 49-90, model/encoder.py:153-162) through gemm_xw_kernel re-read A once per 128-column slab -- six times at N = 768 -- in
// MFMA-fragment shape (every 128-byte line asked for four times): 0.84 ms = 365 TF for ReDAF at 130k nodes.
// Here the block is gemm_xw8_kernel's -- 8 waves, wave w owns columns [32 w, 32 w + 32) of the block's 256, A tiles by
// LDS-DMA into a ring, one raw barrier per tile, counted vmcnt waits -- extended along K: the wave keeps its 32 weight rows
// for ALL K as 16 KC MFMA operands (192 VGPRs at K = 768).  That leaves 64 registers, so a tile is 32 rows (four 16 x 16
// accumulators, 16 registers; the 64-row form spilled -- and a spill is a scratch access on the vmcnt counter the ring is
// waited by) x all K: KC planes of 32 rows x 512 B, gemm_xw8_kernel's swizzle inside a plane; three ring slots (48 KB each
// at K = 768), two tiles in flight.  Per tile and CU: 48 KB through the DMA, 384 KB of LDS fragment reads (1536 clocks at
// 256 B/clk) under 2 x 1536 MFMA clocks per SIMD: matrix-pipe bound, not LDS bound.
//   A block's column group is FIXED (the weights are loaded once): N / 256 column groups x row-tile streams.  The blocks
// that walk the same row tiles for the N / 256 column groups carry the same blockIdx.x % 8 -- one XCD under round-robin
// placement (speed only) -- and run in step, so A comes from HBM once and from that XCD's L2 for the other groups.
//   Nothing in the loop is an ordinary global load (hipcc would drain the DMA ring at its first use): bias and the ReDAF
// scale rows of the block's columns are staged into LDS behind the ring (ONE __shared__ object) before the first DMA.
//   vmcnt bookkeeping: tile i's DMAs are issued in iteration i - 2, behind its barrier; behind them this wave issues the
// stores of tile i - 2, the DMAs of tile i + 1 and the stores of tile i - 1 (a partial tile -- fewer stores -- can only be
// a block's last): the wait before tile i leaves exactly those in flight.
constexpr int XK_BM = 32, XK_NS = 3;
constexpr int XK_CONST_BYTES = 3 * 256 * 4;   // bias | ReDAF scale of modality 0 | of modality 1, this block's 256 columns
constexpr int XK_JUNK_BYTES = 8 * 256;        // where the L2 prefetch's dwords land (one 256-byte target per wave; never read)
#ifndef XK_L2PF
#define XK_L2PF 0        // tiles an L2 prefetch runs ahead of the DMA (0: none -- measured: 2 / 3 / 6 tiles ahead cost 3 / 6 / 7 %)
#endif
#ifndef XK_ABL
#define XK_ABL 0         // ablations for timing only (wrong results): 1 no DMA, 2 no epilogue, 4 no matrix phase, 8 no fragment reads in it
#endif
#ifndef XK_STAGGER
#define XK_STAGGER 1     // waves 4-7 defer their epilogue by one tile (0: all eight waves in step)
#endif
#ifndef XK_PF
#define XK_PF 3          // A fragments read ahead of their MFMA (4 registers each; 192 of the 256 are the weights)
#endif

template <int N> __device__ __forceinline__ void xk_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// LDS reads of this kernel are hand-issued: (a) left to hipcc, the 16 KC fragment reads are either hoisted until the
// kernel's state spills (a spill is a scratch access on the vmcnt counter the ring is waited by) or, with the order pinned
// by sched_group_barrier, each read is waited for right in front of its MFMA; (b) hipcc puts s_waitcnt vmcnt(0) in front of
// a read of the constants behind the ring (it cannot tell them from the slots the DMAs in flight are writing).
typedef float xk_f4_t __attribute__((ext_vector_type(4)));
template <int OFF> __device__ __forceinline__ xw_bf16x8_t xk_frag(uint32_t addr) {
  xw_bf16x8_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ xw_bf16x8_t xk_frag_rt(uint32_t addr, int off) {      // (off: 0 or 8192, compile-time after unrolling)
  return off ? xk_frag<8192>(addr) : xk_frag<0>(addr);
}
template <int N> __device__ __forceinline__ void xk_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);      // (an MFMA touches no memory: the clobber alone does not keep it behind the wait)
}
__device__ __forceinline__ float xk_swap1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
template <int OFF> __device__ __forceinline__ float4 xk_const4(uint32_t addr) {
  xk_f4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return make_float4(v[0], v[1], v[2], v[3]);
}
typedef float xk_acc4_t __attribute__((ext_vector_type(4)));
// Half-step H of a tile's 16 KC: k-step k32 = H / 2 (32 deep), row group rg = H % 2 (16 rows).  The fragment of half-step
// H + XK_PF leaves, the wait leaves the younger reads in flight, then the fragment meets the wave's two 16-column weight
// fragments of this k-step: v_mfma_f32_16x16x32_bf16 into FOUR independent accumulators (rg, ch) -- the same accumulator
// comes round every fourth MFMA.  (Round 5 first built this on v_mfma_f32_32x32x16_bf16 with ONE 32 x 32 accumulator; the
// two forms run at the same speed and give the same bits -- the matrix phase alone, without DMA, epilogue or fragment
// reads (XK_ABL = 11), takes 0.19 ms for 307 GFLOP = 1.6 PF: the matrix pipe is saturated at the clock the chip holds under
// back-to-back bf16 MFMAs -- this form keeps ten more registers free of the 256.)
template <int H, int KC> __device__ __forceinline__ void xk_step(xk_acc4_t (&acc)[2][2], const xw_bf16x8_t (&w)[2][8 * KC],
                                                                 xw_bf16x8_t (&a)[XK_PF + 1], uint32_t tile, int q0) {
  constexpr int T = 16 * KC, NX = H + XK_PF;
  if constexpr (NX < T && !(XK_ABL & 8)) {      // (XK_ABL & 8: timing only -- no fragment reads in the loop)
    constexpr int k32 = NX / 2, rg = NX % 2;
    // row 16 rg + fr, 16-byte chunk 4 (k32 % 8) + fq of plane k32 / 8, at position chunk ^ row (the source swizzle)
    a[NX % (XK_PF + 1)] = xk_frag<(k32 / 8) * (XK_BM * 512) + rg * 8192>(tile + (uint32_t)((64 * (k32 % 8)) ^ q0 ^ (256 * rg)));
  }
  xk_lgkm<(T - 1 - H) < XK_PF ? (T - 1 - H) : XK_PF>();
  acc[H % 2][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0][H / 2], a[H % (XK_PF + 1)], acc[H % 2][0], 0, 0, 0);
  acc[H % 2][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1][H / 2], a[H % (XK_PF + 1)], acc[H % 2][1], 0, 0, 0);
  if constexpr (H + 1 < T) xk_step<H + 1, KC>(acc, w, a, tile, q0);
}
template <typename CT, int KC, bool REDAF>
__global__ __launch_bounds__(512) void gemm_xw8k_kernel(XwArgs g, GemmEpilogue ep, int n_groups, int streams) {
  constexpr int SLOT = XK_BM * KC * 512;
  __shared__ __attribute__((aligned(1024))) unsigned char ring[XK_NS * SLOT + XK_CONST_BYTES + XK_JUNK_BYTES];
  float* cst = reinterpret_cast<float*>(ring + XK_NS * SLOT);
  const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)ring;
  // 192 of the 256 registers hold weights: everything wave-uniform is kept in SGPRs (readfirstlane), and what depends on the
  // lane is derived from the thread id again in every iteration (tid is laundered there, so hipcc cannot keep a dozen
  // loop-invariant lane values in registers for the kernel's life -- it then spilled weights, into the loop)
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int cg = jb % n_groups, rs = jb / n_groups;
  const int n0 = cg * 256 + wave * 32;
  const __bf16* __restrict__ A = reinterpret_cast<const __bf16*>(g.A);
  const int m_tiles = (g.M + XK_BM - 1) / XK_BM, stride = streams * 8, first = rs * 8 + xcd;
  if (first >= m_tiles) return;   // block-uniform, before any barrier

  // constants of the block's columns -> LDS (ordinary loads: BEFORE the first DMA)
  if (threadIdx.x < 256) {
    const int e = threadIdx.x;
    cst[e] = ep.bias ? ep.bias[cg * 256 + e] : 0.f;
    if constexpr (REDAF) {
      cst[256 + e] = g.rd_scale[cg * 256 + e];
      cst[512 + e] = g.rd_scale[(int64_t)g.N + cg * 256 + e];
    }
  }
  uint32_t sid = 0;
  if constexpr (REDAF) sid = __builtin_amdgcn_readfirstlane(with_epoch(g.rd_stream, g.rd_epoch));
  // the weights: the wave's two 16-column fragments (rows n0 + 16 ch + fr of Wt) of every 32-deep k-step
  xw_bf16x8_t w[2][8 * KC];
  {
    const int fr = threadIdx.x & 15, fq = (threadIdx.x >> 4) & 3;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int k32 = 0; k32 < 8 * KC; ++k32)
        w[ch][k32] = *reinterpret_cast<const xw_bf16x8_t*>(g.Wt + (int64_t)(n0 + 16 * ch + fr) * g.ldw + k32 * 32 + fq * 8);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                 // cst is visible; nothing is in flight

  // DMA pieces of a lane: rows wave * 4 + 2 j + (lane >> 5) of a tile, every K plane, 16-byte chunk lane & 31 (swizzled source)
  auto issue = [&](int slot, int tile, int l31, int lh) {
    const uint32_t base = (uint32_t)(slot * SLOT + wave * (4 * 512));
    const int m0 = tile * XK_BM;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rr = wave * 4 + 2 * j + lh;
      const int r = max(0, min(m0 + rr, g.M - 1));   // past the end: valid memory, never stored
      const __bf16* src = A + (int64_t)r * g.lda + ((l31 ^ rr) * 8);
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + kc * 256),
                                         (__attribute__((address_space(3))) void*)(ring + base + kc * (XK_BM * 512) + j * 1024), 16, 0, 0);
    }
  };
  // L2 prefetch.  The three blocks that read a row tile (one per column group) ask for it together and all wait for HBM: a CU
  // with two tiles (96 KB) in flight streams from HBM at ~23 GB/s (MI355X_MICROARCH.md "Indexed rows"), 2.1 us per 48 KB tile
  // against 1.3 us of matrix work.  LDS has no room for more tiles in flight, the L2 has: every wave touches each 128-byte
  // line of ITS rows of the tile XK_L2PF tiles further on with one dword-per-lane LDS-DMA into a junk area -- one more
  // instruction on the vmcnt counter per tile, no registers -- so that the tile's real DMA finds its lines in the L2.
  constexpr int LPR64 = KC * 8;                        // 64-byte sectors per row (a 4-byte touch brings in ONE 64-byte sector)
  constexpr int PFL = XK_L2PF > 0 ? (4 * LPR64 + 63) / 64 : 0;      // prefetch instructions per wave and tile
  auto prefetch = [&](int tile, int lane) {
#pragma unroll
    for (int i = 0; i < PFL; ++i) {
      const int sct = i * 64 + lane;                   // sector of the wave's 4 rows
      const int rr = wave * 4 + sct / LPR64;
      const int r = max(0, min(tile * XK_BM + rr, g.M - 1));
      if (sct < 4 * LPR64)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + (int64_t)r * g.lda + (sct % LPR64) * 32),
                                         (__attribute__((address_space(3))) void*)(ring + XK_NS * SLOT + XK_CONST_BYTES + wave * 256), 4, 0, 0);
    }
  };
  if constexpr (PFL) {
#pragma unroll
    for (int d = 0; d < XK_L2PF; ++d) prefetch(first + (2 + d) * stride, threadIdx.x & 63);
  }
  issue(0, first, threadIdx.x & 31, (threadIdx.x >> 5) & 1);
  issue(1, first + stride, threadIdx.x & 31, (threadIdx.x >> 5) & 1);

  constexpr int D = 2 * KC;                                         // DMA instructions per wave and tile
  constexpr int SPT = REDAF ? 4 : (sizeof(CT) == 2 ? 2 : 4);        // store instructions per wave and tile
  int slot = 0, it = 0;
  float keep_scale = 1.0f;
  uint32_t thresh = 0u;
  if constexpr (REDAF) {
    if (g.rd_p > 0.f) keep_scale = 1.0f / (1.0f - g.rd_p);
    keep_scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, keep_scale)));
    thresh = __builtin_amdgcn_readfirstlane((uint32_t)fminf(g.rd_p * 4294967296.0f, 4294967295.0f));
  }
  // the tile's epilogue: acc[rg][ch][j] = C[t * 32 + 16 rg + fr][n0 + 16 ch + 4 fq + j]   (fr = lane & 15, fq = lane >> 4)
  auto epilogue = [&](const xk_acc4_t (&acc)[2][2], int t) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int fr = tid & 15, fq = (tid >> 4) & 3;
    const uint32_t cb = ring_lds + (uint32_t)(XK_NS * SLOT + (wave * 32 + 4 * fq) * 4);
    if constexpr (REDAF) {
      // ReDAF.forward behind the product (reference biomedkg/utils/fusion.py:49-90), as gemm_xw_kernel<.., REDAF> runs it:
      // the two modality rows of a node are the lanes fr and fr ^ 1; the keep hash (hash_keep4) is indexed exactly as
      // bkg_redaf_combine_f32 indexes it (the same masks)
      const int mrow = fr & 1;
#pragma unroll
      for (int rg = 0; rg < 2; ++rg) {
        const int row = t * XK_BM + 16 * rg + fr;
        const bool live = row < g.M;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          const int col = n0 + 16 * ch + 4 * fq;
          float4 v = make_float4(acc[rg][ch][0], acc[rg][ch][1], acc[rg][ch][2], acc[rg][ch][3]);
          const float4 b = ch ? xk_const4<64>(cb) : xk_const4<0>(cb);
          v.x = fmaxf(v.x + b.x, 0.f); v.y = fmaxf(v.y + b.y, 0.f); v.z = fmaxf(v.z + b.z, 0.f); v.w = fmaxf(v.w + b.w, 0.f);
          const float4 sc = ch ? xk_const4<1024 + 64>(cb + 1024 * mrow) : xk_const4<1024>(cb + 1024 * mrow);
          v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w;
          if (g.rd_p > 0.f) {
            const uint4 r = hash_keep4(g.rd_seed, (uint64_t)(g.rd_row0 + row) * g.N + col, sid);
            v.x = r.x >= thresh ? v.x * keep_scale : 0.f; v.y = r.y >= thresh ? v.y * keep_scale : 0.f;
            v.z = r.z >= thresh ? v.z * keep_scale : 0.f; v.w = r.w >= thresh ? v.w * keep_scale : 0.f;
          }
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          float4 o;   // the partner lane's values (lane ^ 1: quad_perm [1, 0, 3, 2] -- a DPP move, no LDS round trip)
          o.x = xk_swap1(v.x); o.y = xk_swap1(v.y); o.z = xk_swap1(v.z); o.w = xk_swap1(v.w);
          if (live && mrow == 0)   // modality 0 first, as the two-kernel form adds them
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.C) + (int64_t)(row >> 1) * g.ldc + col) =
                make_float4((v.x + o.x) * 0.5f, (v.y + o.y) * 0.5f, (v.z + o.z) * 0.5f, (v.w + o.w) * 0.5f);
        }
      }
    } else {
#pragma unroll
      for (int rg = 0; rg < 2; ++rg) {
        const int row = t * XK_BM + 16 * rg + fr;
        float4 v[2];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          const float4 b = ch ? xk_const4<64>(cb) : xk_const4<0>(cb);
          v[ch] = make_float4(ep.alpha * acc[rg][ch][0] + b.x, ep.alpha * acc[rg][ch][1] + b.y,
                              ep.alpha * acc[rg][ch][2] + b.z, ep.alpha * acc[rg][ch][3] + b.w);
          if (ep.act == 1) {
            v[ch].x = fmaxf(v[ch].x, 0.f); v[ch].y = fmaxf(v[ch].y, 0.f); v[ch].z = fmaxf(v[ch].z, 0.f); v[ch].w = fmaxf(v[ch].w, 0.f);
          } else if (ep.act == 2) {
            v[ch].x = v[ch].x > 0.f ? v[ch].x : expm1f(v[ch].x); v[ch].y = v[ch].y > 0.f ? v[ch].y : expm1f(v[ch].y);
            v[ch].z = v[ch].z > 0.f ? v[ch].z : expm1f(v[ch].z); v[ch].w = v[ch].w > 0.f ? v[ch].w : expm1f(v[ch].w);
          }
        }
        if constexpr (sizeof(CT) == 4) {
          if (row < g.M) {
            float* __restrict__ drow_p = reinterpret_cast<float*>(g.C) + (int64_t)row * g.ldc + n0 + 4 * fq;
            *reinterpret_cast<float4*>(drow_p) = v[0];
            *reinterpret_cast<float4*>(drow_p + 16) = v[1];
          }
        } else {
          // bf16: the lane pair (fq, fq ^ 1: lanes 16 apart, the same row) trades one packed quad so that each lane owns 8
          // CONSECUTIVE columns and writes one 16-byte piece: even fq the columns 4 fq .. + 7 of the first 16, odd fq the
          // columns 4 (fq - 1) .. + 7 of the second 16
          const uint2 p0 = xw_pack4(v[0]), p1 = xw_pack4(v[1]);
          const bool odd = fq & 1;
          const uint2 give = odd ? p0 : p1;
          uint2 got;
          got.x = __shfl_xor(give.x, 16, 64);
          got.y = __shfl_xor(give.y, 16, 64);
          const uint4 o = odd ? make_uint4(got.x, got.y, p1.x, p1.y) : make_uint4(p0.x, p0.y, got.x, got.y);
          if (row < g.M)
            *reinterpret_cast<uint4*>(reinterpret_cast<__bf16*>(g.C) + (int64_t)row * g.ldc + n0 + (odd ? 16 + 4 * (fq - 1) : 4 * fq)) = o;
        }
      }
    }
  };
  // The two waves of a SIMD (w and w + 4) would run in lockstep -- matrix work together, epilogue (VALU, keep hash, stores: no
  // MFMA) together.  Waves 4-7 therefore DEFER each tile's epilogue behind the next tile's barrier (MI355X_MICROARCH.md
  // "Two waves per SIMD", item 9): after every barrier waves 0-3 multiply while waves 4-7 finish their previous tile, then
  // the roles swap.  Their stores come one iteration later, so their vmcnt ladder starts one step later.
  const bool late = XK_STAGGER && wave >= 4;
  xk_acc4_t acc[2][2];
  int t_prev = -1;
  for (int t = first; t < m_tiles; t += stride) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int l31 = tid & 31, lh = (tid >> 5) & 1;      // (the DMA pieces' lane roles)
    // behind this tile's DMAs: the next tile's (D), the prefetches of the iterations since (at most two) and the stores of
    // the tiles finished since (at most two; one iteration later for the deferring waves)
    const int stores_behind = min(late ? max(it - 1, 0) : it, 2), pf_behind = min(it, 2);
    if (pf_behind == 0) xk_wait<D>();
    else if (pf_behind == 1) { if (stores_behind == 1) xk_wait<D + PFL + SPT>(); else xk_wait<D + PFL>(); }
    else { if (stores_behind == 2) xk_wait<D + 2 * PFL + 2 * SPT>(); else xk_wait<D + 2 * PFL + SPT>(); }
    ++it;
    __builtin_amdgcn_s_barrier();      // ... every wave's pieces have landed, and every wave is done with the slot refilled next
    if constexpr (!(XK_ABL & 1)) issue(slot == 0 ? XK_NS - 1 : slot - 1, t + 2 * stride, l31, lh);
    if constexpr (PFL) prefetch(t + (2 + XK_L2PF) * stride, tid & 63);
    if (late && t_prev >= 0 && !(XK_ABL & 2)) epilogue(acc, t_prev);
#pragma unroll
    for (int rg = 0; rg < 2; ++rg)
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) acc[rg][ch] = xk_acc4_t{0.f, 0.f, 0.f, 0.f};
    if constexpr (!(XK_ABL & 4)) {
      const int fr = tid & 15, fq = (tid >> 4) & 3;
      const int q0 = 16 * (fq ^ fr);
      const uint32_t tile = ring_lds + (uint32_t)(slot * SLOT + fr * 512);
      xw_bf16x8_t a[XK_PF + 1];
#pragma unroll
      for (int h0 = 0; h0 < XK_PF; ++h0)       // (XK_PF <= 16: plane 0) half-step h0: k-step h0 / 2, row group h0 % 2
        a[h0] = xk_frag_rt(tile + (uint32_t)((64 * ((h0 / 2) % 8)) ^ q0 ^ (256 * (h0 % 2))), (h0 % 2) * 8192);
      xk_step<0, KC>(acc, w, a, tile, q0);
    }
    slot = slot + 1 == XK_NS ? 0 : slot + 1;
    if (!late && !(XK_ABL & 2)) epilogue(acc, t);
    if constexpr ((XK_ABL & 2) != 0) asm volatile("" ::"v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[1][0]), "v"(acc[1][1]));
    t_prev = t;
  }
  if (late && t_prev >= 0) epilogue(acc, t_prev);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the look-ahead DMAs past the last tile
}

// grid of gemm_xw8k_kernel: 8 (XCD classes) x column groups x row-tile streams, at most one block per CU
static inline void xk_grid(int64_t M, int64_t N, int* n_groups, int* streams, unsigned* blocks) {
  const int ng = (int)(N / 256);
  int st = std::max(1, (256 / 8) / ng);
  st = (int)std::min<int64_t>(st, ceil_div(ceil_div(M, XK_BM), 8));
  *n_groups = ng; *streams = st; *blocks = (unsigned)(8 * ng * st);
}

// weights -> the [N][K] bf16 image the kernel above reads (a 64 x 64 patch per block through LDS)
__global__ __launch_bounds__(256) void weight_to_bf16_kernel(const float* __restrict__ W, int64_t ldw, int64_t stride_w,
                                                             int rows, int cols, __bf16* __restrict__ out, int64_t ldo,
                                                             int64_t stride_o, __bf16* __restrict__ out_t, int64_t ldt,
                                                             int64_t stride_t) {
  __shared__ float t[64][65];
  const float* Wb = W + (int64_t)blockIdx.z * stride_w;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int r = e / 64, c = e % 64;
    t[r][c] = (r0 + r < rows && c0 + c < cols) ? Wb[(int64_t)(r0 + r) * ldw + c0 + c] : 0.f;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int a = e / 64, b = e % 64;
    if (out && r0 + a < rows && c0 + b < cols)
      out[(int64_t)blockIdx.z * stride_o + (int64_t)(r0 + a) * ldo + c0 + b] = (__bf16)t[a][b];
    if (out_t && c0 + a < cols && r0 + b < rows)
      out_t[(int64_t)blockIdx.z * stride_t + (int64_t)(c0 + a) * ldt + r0 + b] = (__bf16)t[b][a];
  }
}

}  // namespace bkg

using namespace bkg;

// BKG_XW_SLAB=0: the K = 256 KC products keep the streaming kernel (A / B measurements, scripts/bench_gemm.py)
static bool xk_disabled() {
  const char* v = getenv("BKG_XW_SLAB");      // (read per call: the tests switch between the two kernels in one process)
  return v && v[0] == '0';
}

extern "C" int bkg_weight_to_bf16(const float* W, int64_t ldw, int64_t stride_w, int64_t rows, int64_t cols,
                                  int32_t batch, void* out, int64_t ldo, int64_t stride_o, void* out_t, int64_t ldt,
                                  int64_t stride_t, void* stream) {
  BKG_ARG(rows > 0 && cols > 0 && batch > 0 && rows < (1 << 24) && cols < (1 << 24), "bkg_weight_to_bf16: bad shape");
  BKG_ARG(out || out_t, "bkg_weight_to_bf16: no output");
  dim3 grid(ceil_div(cols, 64), ceil_div(rows, 64), batch);
  hipLaunchKernelGGL(weight_to_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, W, ldw, stride_w, (int)rows,
                     (int)cols, reinterpret_cast<__bf16*>(out), ldo, stride_o, reinterpret_cast<__bf16*>(out_t), ldt,
                     stride_t);
  return check_launch("bkg_weight_to_bf16");
}

// dX = gate(dXW W) of a 256 -> 256 layer as a bf16 table, and the bias gradient of the layer BELOW (the column sums of
// that table), in one launch pair: bkg_gemm_xw's 8-wave kernel with the gate in its epilogue + colsum.hip's final add.
// `workspace`: bkg_gemm_xw8_gated_workspace_floats(M) floats.
extern "C" int64_t bkg_gemm_xw8_gated_workspace_floats(int64_t M) {
  return std::min<int64_t>(256, ceil_div(std::max<int64_t>(M, 1), X8_BM)) * X8_N;
}
extern "C" int bkg_gemm_xw8_gated(int64_t M, const void* A, int64_t lda, const void* Wt, int64_t ldw, void* C, int64_t ldc,
                                  const void* gate, int64_t ldg, float gate_scale, float* workspace, float* colsum,
                                  void* stream) {
  BKG_ARG(M > 0 && M < (1ll << 31), "bkg_gemm_xw8_gated: bad row count");
  BKG_ARG(A && Wt && C && gate && workspace && colsum, "bkg_gemm_xw8_gated: null argument");
  BKG_ARG((uintptr_t)A % 16 == 0 && (lda * 2) % 16 == 0 && (uintptr_t)Wt % 16 == 0 && (ldw * 2) % 16 == 0 &&
              (uintptr_t)C % 16 == 0 && (ldc * 2) % 16 == 0 && (uintptr_t)gate % 16 == 0 && (ldg * 2) % 16 == 0 &&
              (uintptr_t)workspace % 16 == 0,
          "bkg_gemm_xw8_gated: 16-byte aligned rows");
  GemmEpilogue ep{1.0f, 0.0f, nullptr, 0, nullptr, 0, 1.0f, 0};
  XwArgs g{A, lda, nullptr, reinterpret_cast<const __bf16*>(Wt), ldw, 0, C, ldc, nullptr, 0, (int)M, X8_N, X8_K, 0};
  const unsigned blocks = (unsigned)std::min<int64_t>(256, ceil_div(M, X8_BM));
  XwGate gt{reinterpret_cast<const __bf16*>(gate), ldg, gate_scale, workspace};
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((gemm_xw8_kernel<__bf16, true>), dim3(blocks), dim3(512), 0, s, g, ep, gt);
  colsum_final_launch(workspace, (int)blocks, X8_N, colsum, s);
  return check_launch("bkg_gemm_xw8_gated");
}

extern "C" int bkg_gemm_xw(int64_t M, int64_t N, int64_t K, float alpha, const void* A, int32_t a_is_bf16, int64_t lda,
                           const int32_t* a_rows, const void* Wt, int64_t ldw, int64_t stride_w, const int32_t* group_ptr, int32_t G,
                           float beta, void* C, int32_t c_is_bf16, int64_t ldc, const float* bias, int32_t act,
                           const float* gate, int64_t ldg, float gate_scale, void* stream) {
  BKG_ARG(M >= 0 && N > 0 && K > 0 && M < (1ll << 31) && N < (1 << 24) && K < (1 << 24), "bkg_gemm_xw: bad shape");
  BKG_ARG(K % XW_KS == 0 && N % 32 == 0, "bkg_gemm_xw: K must be a multiple of 256 and N of 32");
  BKG_ARG(act >= 0 && act <= 2, "bkg_gemm_xw: bad act");
  const int a_el = a_is_bf16 ? 2 : 4, c_el = c_is_bf16 ? 2 : 4;
  BKG_ARG((uintptr_t)A % 16 == 0 && (lda * a_el) % 16 == 0 && (uintptr_t)Wt % 16 == 0 && (ldw * 2) % 16 == 0 &&
              (stride_w * 2) % 16 == 0,
          "bkg_gemm_xw: A / Wt rows must be 16-byte aligned");
  BKG_ARG((uintptr_t)C % 16 == 0 && (ldc * c_el) % (4 * c_el) == 0 && (!bias || (uintptr_t)bias % 16 == 0) &&
              (!gate || ((uintptr_t)gate % 16 == 0 && ldg % 4 == 0)),
          "bkg_gemm_xw: C / bias / gate must allow 4-column vector access");
  BKG_ARG(!(c_is_bf16 && beta != 0.f), "bkg_gemm_xw: beta needs an fp32 C");
  BKG_ARG(group_ptr == nullptr || G > 0, "bkg_gemm_xw: G must be positive with group_ptr");
  if (M == 0) return 0;
  GemmEpilogue ep{alpha, beta, bias, act, gate, ldg, gate_scale, 0};
  const int n_tiles = ceil_div(N, XW_BN);
  XwArgs g{A, lda, a_rows, reinterpret_cast<const __bf16*>(Wt), ldw, stride_w, C, ldc, group_ptr, G, (int)M, (int)N, (int)K, 0};
  hipStream_t s = (hipStream_t)stream;
  if (a_is_bf16 && !group_ptr && !a_rows && !gate && beta == 0.f && K == X8_K && N == X8_N) {
    const unsigned blocks = (unsigned)std::min<int64_t>(256, ceil_div(M, X8_BM));
    if (c_is_bf16) hipLaunchKernelGGL(gemm_xw8_kernel<__bf16>, dim3(blocks), dim3(512), 0, s, g, ep, XwGate{});
    else hipLaunchKernelGGL(gemm_xw8_kernel<float>, dim3(blocks), dim3(512), 0, s, g, ep, XwGate{});
    return check_launch("bkg_gemm_xw(8 waves)");
  }
  if (a_is_bf16 && !group_ptr && !a_rows && !gate && beta == 0.f && (K == 512 || K == 768) && N % 256 == 0 && N <= 8 * 256 &&
      !xk_disabled()) {
    int ng, st; unsigned blocks;
    xk_grid(M, N, &ng, &st, &blocks);
#define BKG_XK(CT, KC) hipLaunchKernelGGL((gemm_xw8k_kernel<CT, KC, false>), dim3(blocks), dim3(512), 0, s, g, ep, ng, st)
    const int kc = (int)(K / 256);
    if (c_is_bf16) { if (kc == 2) BKG_XK(__bf16, 2); else BKG_XK(__bf16, 3); }
    else { if (kc == 2) BKG_XK(float, 2); else BKG_XK(float, 3); }
#undef BKG_XK
    return check_launch("bkg_gemm_xw(weight slab in registers)");
  }
  if (a_is_bf16 && group_ptr && G <= 63 && !gate && beta == 0.f && K == X8_K && N == X8_N) {
    const unsigned blocks = (unsigned)std::min<int64_t>(256, ceil_div(M, X8_BM) + G);
    if (c_is_bf16) hipLaunchKernelGGL(gemm_xw8_grouped_kernel<__bf16>, dim3(blocks), dim3(512), 0, s, g, ep, group_ptr, a_rows);
    else hipLaunchKernelGGL(gemm_xw8_grouped_kernel<float>, dim3(blocks), dim3(512), 0, s, g, ep, group_ptr, a_rows);
    return check_launch("bkg_gemm_xw(8 waves, grouped)");
  }
  unsigned grid;
  if (group_ptr) {
    grid = xcd_grid(ceil_div(M, XW_BM) + G, n_tiles);
  } else {
    // persistent: two blocks per CU walk the row tiles with a fixed stride; the weight slab is loaded once (K <= 256)
    const int64_t m_tiles = ceil_div(M, XW_BM);
    int64_t lanes = std::max<int64_t>(BKG_XCDS, (2 * 256 / n_tiles) / BKG_XCDS * BKG_XCDS);
    lanes = std::min<int64_t>(lanes, (m_tiles + BKG_XCDS - 1) / BKG_XCDS * BKG_XCDS);
    g.persist = (int)lanes;
    grid = (unsigned)(lanes * n_tiles);
  }
#define BKG_XW(AT, CT)                                                                                          \
  do {                                                                                                          \
    if (group_ptr) hipLaunchKernelGGL((gemm_xw_kernel<AT, CT, true>), dim3(grid), dim3(256), 0, s, g, ep);      \
    else hipLaunchKernelGGL((gemm_xw_kernel<AT, CT, false>), dim3(grid), dim3(256), 0, s, g, ep);               \
  } while (0)
  if (a_is_bf16) { if (c_is_bf16) BKG_XW(__bf16, __bf16); else BKG_XW(__bf16, float); }
  else { if (c_is_bf16) BKG_XW(float, __bf16); else BKG_XW(float, float); }
#undef BKG_XW
  return check_launch("bkg_gemm_xw");
}

// ReDAF.forward as ONE launch (reference biomedkg/utils/fusion.py:49-90; M = 2 modalities): out[n] = mean_m relu(dropout(
// relu(x[n, m] W^T + b) * scale[m])), x rows 2n / 2n + 1 = node n's tokens.  keep (nullable): replayed dropout draws [rows][N].
extern "C" int bkg_redaf_fused_xw(int64_t rows, int64_t N, int64_t K, const void* A, int32_t a_is_bf16, int64_t lda,
                                  const void* Wt, int64_t ldw, const float* bias, const float* scale, const uint8_t* keep,
                                  float drop_p, uint64_t seed, uint32_t stream_id, float* out, int64_t ldo, void* stream) {
  BKG_ARG(rows >= 0 && rows % 2 == 0 && rows < (1ll << 31) && N > 0 && N % 32 == 0 && K > 0 && K % XW_KS == 0 &&
              N * K <= (1ll << 22), "bkg_redaf_fused_xw: rows even, N % 32 == 0, K % 256 == 0, N * K <= 2^22");
  const int a_el = a_is_bf16 ? 2 : 4;
  BKG_ARG((uintptr_t)A % 16 == 0 && (lda * a_el) % 16 == 0 && (uintptr_t)Wt % 16 == 0 && (ldw * 2) % 16 == 0 &&
              (uintptr_t)out % 16 == 0 && ldo % 4 == 0 && (uintptr_t)scale % 16 == 0 && (!bias || (uintptr_t)bias % 16 == 0) &&
              (!keep || (uintptr_t)keep % 4 == 0),
          "bkg_redaf_fused_xw: operands must allow 4-column vector access");
  BKG_ARG(drop_p >= 0.f && drop_p < 1.f, "bkg_redaf_fused_xw: drop_p in [0,1)");
  if (rows == 0) return 0;
  GemmEpilogue ep{1.0f, 0.0f, bias, 1, nullptr, 0, 1.0f, 0};
  const int n_tiles = ceil_div(N, XW_BN);
  XwArgs g{A, lda, nullptr, reinterpret_cast<const __bf16*>(Wt), ldw, 0, out, ldo, nullptr, 0, (int)rows, (int)N, (int)K, 0,
           scale, keep, drop_p, seed, stream_id, 0, rng_epoch_ptr()};
  hipStream_t s = (hipStream_t)stream;
  if (a_is_bf16 && !keep && (K == 512 || K == 768) && N % 256 == 0 && N <= 8 * 256 && !xk_disabled()) {
    int ng, st; unsigned blocks;
    xk_grid(rows, N, &ng, &st, &blocks);
    const int kc = (int)(K / 256);
    if (kc == 2) hipLaunchKernelGGL((gemm_xw8k_kernel<float, 2, true>), dim3(blocks), dim3(512), 0, s, g, ep, ng, st);
    else hipLaunchKernelGGL((gemm_xw8k_kernel<float, 3, true>), dim3(blocks), dim3(512), 0, s, g, ep, ng, st);
    return check_launch("bkg_redaf_fused_xw(weight slab in registers)");
  }
  const int64_t m_tiles = ceil_div(rows, XW_BM);
  int64_t lanes = std::max<int64_t>(BKG_XCDS, (2 * 256 / n_tiles) / BKG_XCDS * BKG_XCDS);
  lanes = std::min<int64_t>(lanes, (m_tiles + BKG_XCDS - 1) / BKG_XCDS * BKG_XCDS);
  g.persist = (int)lanes;
  const unsigned grid = (unsigned)(lanes * n_tiles);
  if (a_is_bf16) hipLaunchKernelGGL((gemm_xw_kernel<__bf16, float, false, true>), dim3(grid), dim3(256), 0, s, g, ep);
  else hipLaunchKernelGGL((gemm_xw_kernel<float, float, false, true>), dim3(grid), dim3(256), 0, s, g, ep);
  return check_launch("bkg_redaf_fused_xw");
}
