// Exact-fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate,
// bit-for-bit a k-ordered fmaf chain), with the epilogues the GCL / KGE hot path needs fused in:
//   C = epi(alpha * op(A) @ op(B))      epi: + bias[col], ReLU / ELU, * gate(aux), + beta * C_old
// Used for GCNConv.lin (encoder.py:155,160 -> PyG Linear), the GRACE / GGD / DGI projection heads
// (gcl.py:19-21,49-51,86-90), the fusion projections (fusion.py:17-20,70) and, in its grouped
// forms, RGCNConv's per-relation weight products.
//
// Layout: op(A) is [M,K], op(B) is [K,N], C is [M,N] row-major.  TA: A is stored [K,M]; TB: B is
// stored [N,K] (a torch Linear weight).  Block = 256 threads = 4 waves in a 2x2 grid, each wave
// owning a (BM/2)x(BN/2) patch as 32x32 MFMA tiles.  Operand tiles are staged through LDS
// K-major ([BK][BM+4]) so that both fragments are conflict-free ds_read_b32 (lane -> consecutive
// address) regardless of the global orientation; global loads are 16 B/lane where alignment
// allows.  The next K-tile's global loads are issued before the MFMAs of the current one.
//
// Grouped forms (RGCNConv): rows of A / C are partitioned into G groups by a device-side
// group_ptr; group g multiplies by its own B_g.  `grouped rows` (C_g = A_g B_g) and `grouped K`
// (C_g += A_g^T B_g, split-K over row chunks, f32 atomics into a zeroed C) need no host sync.
#include "common.h"
#include "gemm_epilogue.h"
#include "gemm_tile_bf16.h"

namespace bkg {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;
constexpr int PAD = 4;

// Load 4 consecutive elements along the contiguous dimension with bounds / alignment guards.
__device__ __forceinline__ float4 load4(const float* __restrict__ base, int64_t ld, int r, int c, int R, int C,
                                        bool vec) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r < R) {
    const float* p = base + (int64_t)r * ld + c;
    if (vec && c + 3 < C) {
      v = *reinterpret_cast<const float4*>(p);
    } else {
      if (c < C) v.x = p[0];
      if (c + 1 < C) v.y = p[1];
      if (c + 2 < C) v.z = p[2];
      if (c + 3 < C) v.w = p[3];
    }
  }
  return v;
}

// Stage one operand tile.  `contig_is_k`: the stored matrix is [rows = tile dim (M or N), cols = K].
// Otherwise it is [rows = K, cols = tile dim].  LDS image is always [BK][BT+PAD].
template <int BT, bool CONTIG_IS_K>
struct Stager {
  static constexpr int NV = (BT * BK / 4) / 256;  // float4 per thread
  float4 reg[NV];

  __device__ __forceinline__ void load(const float* __restrict__ base, int64_t ld, int t0, int k0, int T, int K,
                                       bool vec) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int f = threadIdx.x + i * 256;
      if (CONTIG_IS_K) {
        int r = f / (BK / 4), kc = f % (BK / 4);
        reg[i] = load4(base, ld, t0 + r, k0 + kc * 4, T, K, vec);
      } else {
        int kr = f / (BT / 4), tc = f % (BT / 4);
        reg[i] = load4(base, ld, k0 + kr, t0 + tc * 4, K, T, vec);
      }
    }
  }
  __device__ __forceinline__ void store(float (*lds)[BT + PAD]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int f = threadIdx.x + i * 256;
      if (CONTIG_IS_K) {
        int r = f / (BK / 4), kc = f % (BK / 4);
        lds[kc * 4 + 0][r] = reg[i].x;
        lds[kc * 4 + 1][r] = reg[i].y;
        lds[kc * 4 + 2][r] = reg[i].z;
        lds[kc * 4 + 3][r] = reg[i].w;
      } else {
        int kr = f / (BT / 4), tc = f % (BT / 4);
        *reinterpret_cast<float4*>(&lds[kr][tc * 4]) = reg[i];
      }
    }
  }
};

template <int BM, int BN, bool TA, bool TB>
__device__ __forceinline__ void gemm_tile(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                          int64_t ldb, float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                          int m0, int n0, int kbeg, int kend, const GemmEpilogue& ep, bool vecA,
                                          bool vecB) {
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  __shared__ float As[BK][BM + PAD];
  __shared__ float Bs[BK][BN + PAD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Stager<BM, !TA> sa;  // A stored [M,K] (K contiguous) unless TA
  Stager<BN, TB> sb;   // B stored [K,N] (N contiguous) unless TB
  sa.load(A, lda, m0, kbeg, M, kend, vecA);
  sb.load(B, ldb, n0, kbeg, N, kend, vecB);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();
    sa.store(As);
    sb.store(Bs);
    __syncthreads();
    if (k0 + BK < kend) {
      sa.load(A, lda, m0, k0 + BK, M, kend, vecA);
      sb.load(B, ldb, n0, k0 + BK, N, kend, vecB);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[kk + lh][wm0 + i * 32 + l31];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[kk + lh][wn0 + j * 32 + l31];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  __shared__ __attribute__((aligned(16))) float patch[4][16 * (WN + 4)];
  store_acc_tile<TM, TN>(acc, C, ldc, M, N, m0 + wm0, n0 + wn0, ep, patch[wave]);
}

// PREC 0: exact fp32 (v_mfma_f32_32x32x2_f32); PREC 1: bf16 operands, fp32 accumulate (gemm_tile_bf16.h)
template <int PREC, int BM, int BN, bool TA, bool TB, typename AT = float, typename BT = float>
__device__ __forceinline__ void gemm_tile_any(const AT* __restrict__ A, int64_t lda, const BT* __restrict__ B,
                                              int64_t ldb, float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                              int m0, int n0, int kbeg, int kend, const GemmEpilogue& ep, bool vecA,
                                              bool vecB) {
  if constexpr (PREC == 0) gemm_tile<BM, BN, TA, TB>(A, lda, B, ldb, C, ldc, M, N, K, m0, n0, kbeg, kend, ep, vecA, vecB);
  else gemm_tile_bf16<BM, BN, TA, TB, GemmEpilogue, AT, BT>(A, lda, B, ldb, C, ldc, M, N, m0, n0, kbeg, kend, ep, vecA, vecB);
}

__device__ __forceinline__ bool vec_ok(const float* p, int64_t ld) {
  return ((reinterpret_cast<uintptr_t>(p) & 15) == 0) && (ld % 4 == 0);
}
__device__ __forceinline__ bool vec_ok(const __bf16* p, int64_t ld) {
  return ((reinterpret_cast<uintptr_t>(p) & 7) == 0) && (ld % 4 == 0);
}

template <int PREC, int BM, int BN, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int64_t lda,
                                                       const float* __restrict__ B, int64_t ldb,
                                                       float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                       GemmEpilogue ep) {
  // Workgroup b is dispatched to XCD b % 8 and each XCD has its own L2: the N tiles that share an A row-panel take
  // consecutive slots of ONE XCD, so the panel leaves HBM once (xcd_slot in common.h).
  const int n_tiles = (N + BN - 1) / BN, m_tiles = (M + BM - 1) / BM;
  const XcdSlot t = xcd_slot(blockIdx.x, n_tiles);
  if (t.outer >= m_tiles) return;
  const int m0 = t.outer * BM, n0 = t.inner * BN;
  gemm_tile_any<PREC, BM, BN, TA, TB>(A, lda, B, ldb, C, ldc, M, N, K, m0, n0, 0, K, ep, vec_ok(A, lda), vec_ok(B, ldb));
}

// Split-K: blockIdx.y owns K range [y*kchunk, (y+1)*kchunk) and adds its tile into a zeroed C with
// f32 atomics.  For the weight-gradient products dW = dY^T X (K = number of nodes, M x N = a few tiles).
template <int PREC, int BM, int BN, bool TA, bool TB, typename AT = float, typename BT = float>
__global__ __launch_bounds__(256) void gemm_f32_splitk_kernel(const AT* __restrict__ A, int64_t lda,
                                                              const BT* __restrict__ B, int64_t ldb,
                                                              float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                              int kchunk, GemmEpilogue ep, float* __restrict__ det) {
  // the output tiles of one K chunk re-read the same operand strips: keep them on one XCD (see gemm_f32_kernel)
  const int n_tiles = (N + BN - 1) / BN, tiles = n_tiles * ((M + BM - 1) / BM);
  const XcdSlot t = xcd_slot(blockIdx.x, tiles);
  const int m0 = (t.inner / n_tiles) * BM, n0 = (t.inner % n_tiles) * BN;
  const int kbeg = t.outer * kchunk, kend = min(kbeg + kchunk, K);
  if (kbeg >= kend) return;
  if (det) { C = det + (int64_t)t.outer * M * N; ldc = N; }      // deterministic mode: this chunk's own slab, plain stores
  gemm_tile_any<PREC, BM, BN, TA, TB, AT, BT>(A, lda, B, ldb, C, ldc, M, N, K, m0, n0, kbeg, kend, ep, vec_ok(A, lda), vec_ok(B, ldb));
}

// C_g[rows of group g, :] = A_g @ op(B_g);  B_g = B + g * strideB.  Tiles never straddle groups.
template <int PREC, int BM, int BN, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_grouped_rows_kernel(const float* __restrict__ A, int64_t lda,
                                                                    const float* __restrict__ B, int64_t ldb,
                                                                    int64_t strideB, float* __restrict__ C,
                                                                    int64_t ldc, const int32_t* __restrict__ group_ptr,
                                                                    int G, int N, int K, GemmEpilogue ep) {
  __shared__ int s_g, s_m0;
  const int n_tiles = (N + BN - 1) / BN;
  const XcdSlot xs = xcd_slot(blockIdx.x, n_tiles);
  if (threadIdx.x == 0) {
    int t = xs.outer, g = 0, found = -1, m0 = 0;
    for (; g < G; ++g) {
      int rows = group_ptr[g + 1] - group_ptr[g];
      int nt = (rows + BM - 1) / BM;
      if (t < nt) { found = g; m0 = group_ptr[g] + t * BM; break; }
      t -= nt;
    }
    s_g = found; s_m0 = m0;
  }
  __syncthreads();
  const int g = s_g;
  if (g < 0) return;
  const int m0 = s_m0, mend = group_ptr[g + 1];
  const float* Bg = B + (int64_t)g * strideB;
  gemm_tile_any<PREC, BM, BN, false, TB>(A, lda, Bg, ldb, C, ldc, mend, N, K, m0, xs.inner * BN, 0, K, ep,
                                         vec_ok(A, lda), vec_ok(Bg, ldb));
}

// C_g += A_g^T @ B_g over the rows of group g (K dimension = rows), split into chunks of KCHUNK
// rows; C_g = C + g * strideC is [M = colsA, N = colsB] and must be zeroed by the caller.
template <int PREC, int BM, int BN, typename AT = float, typename BT = float>
__global__ __launch_bounds__(256) void gemm_f32_grouped_k_kernel(const AT* __restrict__ A, int64_t lda,
                                                                 const BT* __restrict__ B, int64_t ldb,
                                                                 float* __restrict__ C, int64_t ldc, int64_t strideC,
                                                                 const int32_t* __restrict__ group_ptr, int G, int M,
                                                                 int N, int kchunk, GemmEpilogue ep, float* __restrict__ det) {
  __shared__ int s_g, s_k0;
  const int m_tiles = (M + BM - 1) / BM, n_tiles = (N + BN - 1) / BN;
  const int tiles = m_tiles * n_tiles;
  const XcdSlot xs = xcd_slot(blockIdx.x, tiles);
  if (threadIdx.x == 0) {
    int t = xs.outer, g = 0, found = -1, k0 = 0;
    for (; g < G; ++g) {
      int rows = group_ptr[g + 1] - group_ptr[g];
      int nc = (rows + kchunk - 1) / kchunk;
      if (t < nc) { found = g; k0 = group_ptr[g] + t * kchunk; break; }
      t -= nc;
    }
    s_g = found; s_k0 = k0;
  }
  __syncthreads();
  const int g = s_g;
  if (g < 0) return;
  const int k0 = s_k0;
  const int kend = min(k0 + kchunk, group_ptr[g + 1]);
  const int tile = xs.inner;
  float* Cg = C + (int64_t)g * strideC;
  if (det) { Cg = det + (int64_t)xs.outer * M * N; ldc = N; }     // deterministic mode: chunk xs.outer's own slab
  // op(A) = A^T: stored [K(rows), M] ; op(B) = B stored [K(rows), N]
  gemm_tile_any<PREC, BM, BN, true, false, AT, BT>(A, lda, B, ldb, Cg, ldc, M, N, kend, (tile / n_tiles) * BM,
                                           (tile % n_tiles) * BN, k0, kend, ep, vec_ok(A, lda), vec_ok(B, ldb));
}

template <int PREC, int BM, int BN>
static void launch_gemm(bool ta, bool tb, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                        int64_t ldc, int M, int N, int K, const GemmEpilogue& ep, hipStream_t s) {
  dim3 grid(xcd_grid(ceil_div(M, BM), ceil_div(N, BN))), block(256);
  if (!ta && !tb) hipLaunchKernelGGL((gemm_f32_kernel<PREC, BM, BN, false, false>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, ep);
  else if (!ta && tb) hipLaunchKernelGGL((gemm_f32_kernel<PREC, BM, BN, false, true>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, ep);
  else if (ta && !tb) hipLaunchKernelGGL((gemm_f32_kernel<PREC, BM, BN, true, false>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, ep);
  else hipLaunchKernelGGL((gemm_f32_kernel<PREC, BM, BN, true, true>), grid, block, 0, s, A, lda, B, ldb, C, ldc, M, N, K, ep);
}

}  // namespace bkg

namespace bkg {   // gemm_xw32.hip: exact-fp32 tall products with K = N = 256, the weight in registers
bool gemm_xw32_ok(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, bool plain_or_bias_act);
void launch_gemm_xw32(int64_t M, const float* A, int64_t lda, const float* W, int64_t ldw, int trans_w, float* C, int64_t ldc,
                      const GemmEpilogue& ep, hipStream_t s);
}  // namespace bkg

namespace bkg {   // gemm_tn32.hip: exact-fp32 weight gradients over row-major [K, M] / [K, N] operands, tile-aligned output
bool gemm_tn32_ok(int64_t M, int64_t N, const void* A, int64_t lda, const void* B, int64_t ldb);
void launch_gemm_tn32_splitk(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                             int64_t ldc, int kchunk, int64_t nsplit, const GemmEpilogue& ep, float* det, hipStream_t s);
void launch_gemm_tn32_grouped(int64_t n_chunks, int64_t M, int64_t N, const float* A, int64_t lda, const float* B, int64_t ldb,
                              float* C, int64_t ldc, int64_t strideC, const int32_t* group_ptr, int G, int kchunk,
                              const GemmEpilogue& ep, float* det, hipStream_t s);
}  // namespace bkg

using namespace bkg;

extern "C" int bkg_gemm_f32(int32_t trans_a, int32_t trans_b, int64_t M, int64_t N, int64_t K, float alpha,
                            const float* A, int64_t lda, const float* B, int64_t ldb, float beta, float* C,
                            int64_t ldc, const float* bias, int32_t act, const float* gate, int64_t ldg,
                            float gate_scale, int32_t precision, void* stream) {
  BKG_ARG(M >= 0 && N >= 0 && K >= 0 && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "bkg_gemm_f32: bad shape");
  BKG_ARG(act >= 0 && act <= 2, "bkg_gemm_f32: bad act");
  BKG_ARG(precision == 0 || precision == 1, "bkg_gemm_f32: precision is 0 (fp32) or 1 (bf16 operands)");
  if (M == 0 || N == 0) return 0;
  GemmEpilogue ep{alpha, beta, bias, act, gate, ldg, gate_scale, 0};
  hipStream_t s = (hipStream_t)stream;
  // Deep-K plain products with few output tiles (weight gradients): split K over the grid.
  const bool plain = !bias && act == 0 && !gate && beta == 0.f;
  const int64_t tiles64 = ((M + 63) / 64) * ((N + 63) / 64);
  // (a mini-batch weight gradient -- K ~ 10^3 rows, 16 tiles -- would otherwise run on 16 workgroups)
  const bool shallow = plain && K >= 512 && K < 4096 && tiles64 <= 64;
  if (plain && ((K >= 4096 && tiles64 <= 256) || shallow)) {
    // bf16 operands: 128x128 tiles whenever the output allows -- the product is bound by operand traffic and
    // every strip is then re-read by half as many tiles (measured 0.21 -> 0.14 ms at 256x256x130k; the split-K
    // grid supplies the parallelism).  fp32 operands are MFMA-bound there and prefer more, smaller tiles.
    // (only when its 512-row chunks x 128-wide tiles fill the chip: a 5,000-row GRACE batch has 60 such blocks and runs
    // faster as 240 small LDS-staged tiles -- its captured step went 1.86 -> 2.02 ms when it took this kernel)
    if (precision == 0 && trans_a && !trans_b && ((K + 511) / 512) * ((M / 128) * (N / 128)) >= 256 &&
        gemm_tn32_ok(M, N, A, lda, B, ldb)) {
      // exact-fp32 weight gradient over row-major operands: the LDS-DMA ring form (gemm_tn32.hip), two blocks per CU
      const int64_t t128 = (M / 128) * (N / 128);
      int64_t ns = std::min<int64_t>((K + 511) / 512, std::max<int64_t>(1, 512 / t128));
      const int kc = (int)(((K + ns - 1) / ns + 63) / 64 * 64);
      ns = (K + kc - 1) / kc;
      ep.atomic = 1;
      float* det32 = nullptr;
      BKG_DET_SLABS(det32, ep, ns, M, N, "bkg_gemm_f32");
      if (!det32) fill_zero_rows_f32(C, ldc, M, N, s);
      launch_gemm_tn32_splitk(M, N, K, A, lda, B, ldb, C, ldc, kc, ns, ep, det32, s);
      if (det32) splitk_reduce(det32, C, ldc, 0, (int)M, (int)N, nullptr, 1, kc, (int)ns, s);
      return check_launch("bkg_gemm_f32(tn32)");
    }
    const bool big = (precision == 1 && M >= 128 && N >= 128) || tiles64 >= 64;
    const int64_t tiles = big ? ((M + 127) / 128) * ((N + 127) / 128) : tiles64;
    int64_t nsplit = std::min<int64_t>((K + 511) / 512, std::max<int64_t>(1, (precision ? 768 : 2048) / tiles));   // bf16: ~3 blocks per CU = the tile's occupancy
    if (shallow) nsplit = std::min<int64_t>((K + 127) / 128, std::max<int64_t>(1, 512 / tiles));
    int kchunk = (int)(((K + nsplit - 1) / nsplit + 63) / 64 * 64);   // whole staged K tiles
    nsplit = (K + kchunk - 1) / kchunk;
    ep.atomic = 1;
    float* det = nullptr;
    BKG_DET_SLABS(det, ep, nsplit, M, N, "bkg_gemm_f32");
    if (!det) {      // atomics accumulate into a zeroed C; the deterministic reduce writes every element itself
      fill_zero_rows_f32(C, ldc, M, N, s);
    }
    dim3 grid(xcd_grid(nsplit, tiles)), block(256);
#define BKG_SPLITK(BMN, TA_, TB_)                                                                                      \
  if (precision) hipLaunchKernelGGL((gemm_f32_splitk_kernel<1, BMN, BMN, TA_, TB_>), grid, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, kchunk, ep, det); \
  else hipLaunchKernelGGL((gemm_f32_splitk_kernel<0, BMN, BMN, TA_, TB_>), grid, block, 0, s, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, kchunk, ep, det)
#define BKG_SPLITK_T(BMN)                                 \
  if (!trans_a && !trans_b) { BKG_SPLITK(BMN, false, false); } \
  else if (!trans_a && trans_b) { BKG_SPLITK(BMN, false, true); } \
  else if (trans_a && !trans_b) { BKG_SPLITK(BMN, true, false); } \
  else { BKG_SPLITK(BMN, true, true); }
    if (big) { BKG_SPLITK_T(128) } else { BKG_SPLITK_T(64) }
#undef BKG_SPLITK_T
#undef BKG_SPLITK
    if (det) splitk_reduce(det, C, ldc, 0, (int)M, (int)N, nullptr, 1, kchunk, (int)nsplit, s);
    return check_launch("bkg_gemm_f32(split-K)");
  }
  if (precision == 0 && !trans_a && gemm_xw32_ok(M, N, K, A, lda, !gate && beta == 0.f)) {
    launch_gemm_xw32(M, A, lda, B, ldb, trans_b, C, ldc, ep, s);
    return check_launch("bkg_gemm_f32(xw32)");
  }
  // big tiles once they fill the chip, small tiles otherwise
  const bool big_tiles = (M / 128) * ((N + 127) / 128) >= 256;
  if (precision) {
    if (big_tiles) launch_gemm<1, 128, 128>(trans_a, trans_b, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, ep, s);
    else launch_gemm<1, 64, 64>(trans_a, trans_b, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, ep, s);
  } else {
    if (big_tiles) launch_gemm<0, 128, 128>(trans_a, trans_b, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, ep, s);
    else launch_gemm<0, 64, 64>(trans_a, trans_b, A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, ep, s);
  }
  return check_launch("bkg_gemm_f32");
}

extern "C" int bkg_gemm_grouped_rows_f32(int32_t trans_b, int64_t M_total, int64_t N, int64_t K, const float* A,
                                         int64_t lda, const float* B, int64_t ldb, int64_t strideB, float* C,
                                         int64_t ldc, const int32_t* group_ptr, int32_t G, const float* gate,
                                         int64_t ldg, float gate_scale, int32_t precision, void* stream) {
  BKG_ARG(M_total >= 0 && N > 0 && K > 0 && G > 0, "bkg_gemm_grouped_rows_f32: bad shape");
  BKG_ARG(precision == 0 || precision == 1, "bkg_gemm_grouped_rows_f32: precision is 0 or 1");
  if (M_total == 0) return 0;
  GemmEpilogue ep{1.f, 0.f, nullptr, 0, gate, ldg, gate_scale, 0};
  hipStream_t s = (hipStream_t)stream;
  // 128x128 tiles once they fill the chip; 64x64 for mini-batches (most groups then hold fewer than 128 rows)
  const bool big = (ceil_div(M_total, 128) + G) * (int64_t)ceil_div(N, 128) >= 256;
  const int BT = big ? 128 : 64;
  dim3 grid(xcd_grid(ceil_div(M_total, BT) + G, ceil_div(N, BT))), block(256);
#define BKG_GROWS(P, TB_)                                                                                             \
  if (big) hipLaunchKernelGGL((gemm_f32_grouped_rows_kernel<P, 128, 128, TB_>), grid, block, 0, s, A, lda, B, ldb, strideB, C, ldc, group_ptr, G, (int)N, (int)K, ep); \
  else hipLaunchKernelGGL((gemm_f32_grouped_rows_kernel<P, 64, 64, TB_>), grid, block, 0, s, A, lda, B, ldb, strideB, C, ldc, group_ptr, G, (int)N, (int)K, ep)
  if (precision) { if (trans_b) { BKG_GROWS(1, true); } else { BKG_GROWS(1, false); } }
  else { if (trans_b) { BKG_GROWS(0, true); } else { BKG_GROWS(0, false); } }
#undef BKG_GROWS
  return check_launch("bkg_gemm_grouped_rows_f32");
}

// rows per workgroup of the grouped weight-gradient products: 512 (fp32 operands) / 1024 (bf16) at scale, shorter
// chunks -- multiples of 64 -- when the whole product would otherwise occupy a fraction of the chip (mini-batches)
extern "C" int32_t bkg_gemm_grouped_k_chunk(int64_t rows_total, int64_t M, int64_t N, int32_t precision) {
  const int64_t full = precision ? 1024 : 512;
  const int64_t tiles = (int64_t)ceil_div(M, 128) * ceil_div(N, 128);
  const int64_t want = std::max<int64_t>(1, 512 / std::max<int64_t>(1, tiles));        // chunks that fill the chip twice
  int64_t chunk = (rows_total / want + 63) / 64 * 64;
  return (int32_t)std::min<int64_t>(full, std::max<int64_t>(64, chunk));
}

extern "C" int bkg_gemm_grouped_k_f32(int64_t rows_total, int64_t M, int64_t N, const float* A, int64_t lda,
                                      const float* B, int64_t ldb, float* C, int64_t ldc, int64_t strideC,
                                      const int32_t* group_ptr, int32_t G, int32_t precision, void* stream) {
  BKG_ARG(rows_total >= 0 && M > 0 && N > 0 && G > 0, "bkg_gemm_grouped_k_f32: bad shape");
  BKG_ARG(precision == 0 || precision == 1, "bkg_gemm_grouped_k_f32: precision is 0 or 1");
  if (rows_total == 0) return 0;
  GemmEpilogue ep{1.f, 0.f, nullptr, 0, nullptr, 0, 1.f, 1};
  hipStream_t s = (hipStream_t)stream;
  constexpr int BM = 128, BN = 128;
  const int KCHUNK = bkg_gemm_grouped_k_chunk(rows_total, M, N, precision);   // rows per block
  float* det = nullptr;
  BKG_DET_SLABS(det, ep, ceil_div(rows_total, KCHUNK) + G, M, N, "bkg_gemm_grouped_k_f32");
  if (precision == 0 && (ceil_div(rows_total, KCHUNK) + G) * ((M / 128) * (N / 128)) >= 256 && gemm_tn32_ok(M, N, A, lda, B, ldb)) {
    launch_gemm_tn32_grouped(ceil_div(rows_total, KCHUNK) + G, M, N, A, lda, B, ldb, C, ldc, strideC, group_ptr, G, KCHUNK, ep, det, s);
    if (det) splitk_reduce(det, C, ldc, strideC, (int)M, (int)N, group_ptr, G, KCHUNK, 0, s);
    return check_launch("bkg_gemm_grouped_k_f32(tn32)");
  }
  dim3 grid(xcd_grid(ceil_div(rows_total, KCHUNK) + G, ceil_div(M, BM) * ceil_div(N, BN))), block(256);
  if (precision) hipLaunchKernelGGL((gemm_f32_grouped_k_kernel<1, BM, BN>), grid, block, 0, s, A, lda, B, ldb, C, ldc, strideC, group_ptr, G, (int)M, (int)N, KCHUNK, ep, det);
  else hipLaunchKernelGGL((gemm_f32_grouped_k_kernel<0, BM, BN>), grid, block, 0, s, A, lda, B, ldb, C, ldc, strideC, group_ptr, G, (int)M, (int)N, KCHUNK, ep, det);
  if (det) splitk_reduce(det, C, ldc, strideC, (int)M, (int)N, group_ptr, G, KCHUNK, 0, s);
  return check_launch("bkg_gemm_grouped_k_f32");
}

namespace bkg {   // gemm_tn16.hip: both operands bf16, tile-aligned output
bool gemm_tn16_ok(int64_t M, int64_t N, const void* A, int64_t lda, const void* B, int64_t ldb);
int launch_gemm_tn16_splitk(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                            float* C, int64_t ldc, hipStream_t s);
int launch_gemm_tn16_grouped(int64_t rows_total, int64_t M, int64_t N, const void* A, int64_t lda, const void* B,
                             int64_t ldb, const int32_t* b_rows, float* C, int64_t ldc, int64_t strideC, const int32_t* group_ptr, int G,
                             hipStream_t s);
}  // namespace bkg

// ---- bf16-mode weight gradients with fp32 OR bf16 operands:  C[M,N] = A[K,M]^T B[K,N]  (split-K, f32 atomics)
#define BKG_MIXED_DISPATCH(CALL)                                      \
  if (a_is_bf16) { if (b_is_bf16) { CALL(__bf16, __bf16); } else { CALL(__bf16, float); } } \
  else { if (b_is_bf16) { CALL(float, __bf16); } else { CALL(float, float); } }

extern "C" int bkg_gemm_tn_bf16(int64_t M, int64_t N, int64_t K, const void* A, int32_t a_is_bf16, int64_t lda,
                                const void* B, int32_t b_is_bf16, int64_t ldb, float* C, int64_t ldc, void* stream) {
  BKG_ARG(M > 0 && N > 0 && K >= 0 && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31), "bkg_gemm_tn_bf16: bad shape");
  hipStream_t s = (hipStream_t)stream;
  fill_zero_rows_f32(C, ldc, M, N, s);
  if (K == 0) return 0;
  if (a_is_bf16 && b_is_bf16 && gemm_tn16_ok(M, N, A, lda, B, ldb)) {
    if (int rc = launch_gemm_tn16_splitk(M, N, K, A, lda, B, ldb, C, ldc, s)) return rc;
    return check_launch("bkg_gemm_tn_bf16(tn16)");
  }
  GemmEpilogue ep{1.f, 0.f, nullptr, 0, nullptr, 0, 1.f, 1};
  const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
  int64_t nsplit = std::min<int64_t>((K + 511) / 512, std::max<int64_t>(1, 768 / tiles));
  const int kchunk = (int)(((K + nsplit - 1) / nsplit + 63) / 64 * 64);
  nsplit = (K + kchunk - 1) / kchunk;
  float* det = nullptr;
  BKG_DET_SLABS(det, ep, nsplit, M, N, "bkg_gemm_tn_bf16");
  dim3 grid(xcd_grid(nsplit, tiles)), block(256);
#define BKG_TN(AT_, BT_)                                                                                             \
  hipLaunchKernelGGL((gemm_f32_splitk_kernel<1, 128, 128, true, false, AT_, BT_>), grid, block, 0, s, (const AT_*)A, lda, \
                     (const BT_*)B, ldb, C, ldc, (int)M, (int)N, (int)K, kchunk, ep, det)
  BKG_MIXED_DISPATCH(BKG_TN)
#undef BKG_TN
  if (det) splitk_reduce(det, C, ldc, 0, (int)M, (int)N, nullptr, 1, kchunk, (int)nsplit, s);
  return check_launch("bkg_gemm_tn_bf16");
}

extern "C" int bkg_gemm_grouped_k_bf16(int64_t rows_total, int64_t M, int64_t N, const void* A, int32_t a_is_bf16,
                                       int64_t lda, const void* B, int32_t b_is_bf16, int64_t ldb, const int32_t* b_rows,
                                       float* C, int64_t ldc, int64_t strideC, const int32_t* group_ptr, int32_t G,
                                       void* stream) {
  BKG_ARG(rows_total >= 0 && M > 0 && N > 0 && G > 0, "bkg_gemm_grouped_k_bf16: bad shape");
  if (rows_total == 0) return 0;
  GemmEpilogue ep{1.f, 0.f, nullptr, 0, nullptr, 0, 1.f, 1};
  hipStream_t s = (hipStream_t)stream;
  if (a_is_bf16 && b_is_bf16 && gemm_tn16_ok(M, N, A, lda, B, ldb)) {
    if (int rc = launch_gemm_tn16_grouped(rows_total, M, N, A, lda, B, ldb, b_rows, C, ldc, strideC, group_ptr, G, s)) return rc;
    return check_launch("bkg_gemm_grouped_k_bf16(tn16)");
  }
  BKG_ARG(b_rows == nullptr, "bkg_gemm_grouped_k_bf16: indexed B rows need two bf16 operands and 128-aligned M, N");
  constexpr int BM = 128, BN = 128, KCHUNK = 1024;
  float* det = nullptr;
  BKG_DET_SLABS(det, ep, ceil_div(rows_total, KCHUNK) + G, M, N, "bkg_gemm_grouped_k_bf16");
  dim3 grid(xcd_grid(ceil_div(rows_total, KCHUNK) + G, ceil_div(M, BM) * ceil_div(N, BN))), block(256);
#define BKG_GK(AT_, BT_)                                                                                            \
  hipLaunchKernelGGL((gemm_f32_grouped_k_kernel<1, BM, BN, AT_, BT_>), grid, block, 0, s, (const AT_*)A, lda,      \
                     (const BT_*)B, ldb, C, ldc, strideC, group_ptr, G, (int)M, (int)N, KCHUNK, ep, det)
  BKG_MIXED_DISPATCH(BKG_GK)
#undef BKG_GK
  if (det) splitk_reduce(det, C, ldc, strideC, (int)M, (int)N, group_ptr, G, KCHUNK, 0, s);
  return check_launch("bkg_gemm_grouped_k_bf16");
}
