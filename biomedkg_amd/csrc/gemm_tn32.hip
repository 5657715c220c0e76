// Exact-fp32 weight gradients:  C[M,N] += A[K,M]^T B[K,N]  with K = nodes / segments (dW = dXW^T X, dW_r = H_r^T dY_r
// in the parity mode), 128 x 128 tiles, K split over the grid.
//
// v_mfma_f32_32x32x2_f32 takes ONE scalar per lane and operand -- lane (l31, kh) holds A[k + kh][m + l31] -- and both
// operands are stored with the contracted index as the row, so the LDS image of a stage is the tile exactly as it lies
// in memory (rows of 512 bytes) and a fragment is a conflict-free ds_read_b32: no transpose, no swizzle.  What the
// LDS-staged form of gemm_f32.hip lacks is overlap: one register-staged stage per block, two barriers per stage,
// 64 x 64 tiles (every operand strip re-read by four tiles) -- 287 us for 256 x 256 x 130k against 110 us of MFMA time.
// Here the stages arrive by LDS-DMA into a ring (gemm_tn16.hip's structure: NS - 1 stages in flight across one raw
// barrier per stage, counted vmcnt waits, rows past the K range from a zero page).
#include "common.h"
#include "gemm_epilogue.h"

namespace bkg {

typedef float t32_acc_t __attribute__((ext_vector_type(16)));

constexpr int T32_BT = 128;                      // tile width (m and n)
__device__ uint4 t32_zero_page[32];              // 512 zero bytes

template <int BK, int NS> constexpr int t32_ring_bytes() { return NS * 2 * BK * 512; }

template <int BK, int NS>
__device__ __forceinline__ void gemm_tn32_tile(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                               int64_t ldb, float* __restrict__ C, int64_t ldc, int M, int N, int m0, int n0,
                                               int kbeg, int kend, const GemmEpilogue& ep, unsigned char* ring) {
  constexpr int STAGE = 2 * BK * 512;   // A image, then B image
  constexpr int NJ = BK / 8;            // DMAs per wave, stage and operand (two 512-byte rows each)
  static_assert(t32_ring_bytes<BK, NS>() >= 4 * 16 * (64 + 4) * 4, "the epilogue patch lies over the ring");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, kh = lane >> 5;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;

  t32_acc_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // this lane's NJ pieces per operand and stage: rows wave * (BK / 4) + 2 j + lane / 32 of the stage, 16-byte chunk lane % 32
  int srow[NJ];
  const float* pa[NJ];
  const float* pb[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    srow[j] = wave * (BK / 4) + 2 * j + (lane >> 5);
    pa[j] = A + (int64_t)(kbeg + srow[j]) * lda + m0 + (lane & 31) * 4;
    pb[j] = B + (int64_t)(kbeg + srow[j]) * ldb + n0 + (lane & 31) * 4;
  }
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(t32_zero_page) + 16 * (lane & 31);
  const int64_t a_step = (int64_t)BK * lda, b_step = (int64_t)BK * ldb;
  int knext = kbeg;
  auto issue = [&](int slot) {
    unsigned char* base = ring + slot * STAGE + wave * (BK / 4 * 512);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const bool ok = knext + srow[j] < kend;
      const void* sa = ok ? (const void*)pa[j] : (const void*)zero;
      const void* sb = ok ? (const void*)pb[j] : (const void*)zero;
      pa[j] += a_step;
      pb[j] += b_step;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sa,
                                       (__attribute__((address_space(3))) void*)(base + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sb,
                                       (__attribute__((address_space(3))) void*)(base + BK * 512 + j * 1024), 16, 0, 0);
    }
    knext += BK;
  };

  const int n_st = (kend - kbeg + BK - 1) / BK;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s);   // stages past the end read the zero page
  int slot = 0;
  for (int st = 0; st < n_st; ++st) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * 2 * NJ) : "memory");   // this wave's DMAs of stage st have landed
    __builtin_amdgcn_s_barrier();   // ... everyone's; and every wave is done reading the slot refilled next
    issue(slot == 0 ? NS - 1 : slot - 1);
    const float* As = reinterpret_cast<const float*>(ring + slot * STAGE);
    const float* Bs = As + BK * 128;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[(kk + kh) * 128 + wm0 + i * 32 + l31];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[(kk + kh) * 128 + wn0 + j * 32 + l31];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    slot = slot + 1 == NS ? 0 : slot + 1;
  }
  // the epilogue's transpose patch lies over the ring: drain the zero-page DMAs still in flight, wait for every reader
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  store_acc_tile<2, 2>(acc, C, ldc, M, N, m0 + wm0, n0 + wn0, ep, reinterpret_cast<float*>(ring) + wave * (16 * (64 + 4)));
}

template <int BK, int NS>
__global__ __launch_bounds__(256) void gemm_tn32_splitk_kernel(const float* __restrict__ A, int64_t lda,
                                                               const float* __restrict__ B, int64_t ldb,
                                                               float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                               int kchunk, GemmEpilogue ep, float* __restrict__ det) {
  __shared__ __attribute__((aligned(1024))) unsigned char ring[t32_ring_bytes<BK, NS>()];
  const int n_tiles = N / T32_BT, tiles = n_tiles * (M / T32_BT);
  const XcdSlot t = xcd_slot(blockIdx.x, tiles);
  const int kbeg = t.outer * kchunk, kend = min(kbeg + kchunk, K);
  if (kbeg >= kend) return;
  if (det) { C = det + (int64_t)t.outer * M * N; ldc = N; }
  gemm_tn32_tile<BK, NS>(A, lda, B, ldb, C, ldc, M, N, (t.inner / n_tiles) * T32_BT, (t.inner % n_tiles) * T32_BT, kbeg, kend,
                         ep, ring);
}

template <int BK, int NS>
__global__ __launch_bounds__(256) void gemm_tn32_grouped_kernel(const float* __restrict__ A, int64_t lda,
                                                                const float* __restrict__ B, int64_t ldb,
                                                                float* __restrict__ C, int64_t ldc, int64_t strideC,
                                                                const int32_t* __restrict__ group_ptr, int G, int M,
                                                                int N, int kchunk, GemmEpilogue ep, float* __restrict__ det) {
  // all LDS in ONE object (a second __shared__ beside a DMA-filled array can make hipcc drain the DMAs before every read)
  __shared__ __attribute__((aligned(1024))) unsigned char ring[t32_ring_bytes<BK, NS>()];
  int* s_pick = reinterpret_cast<int*>(ring);
  const int n_tiles = N / T32_BT, tiles = n_tiles * (M / T32_BT);
  const XcdSlot xs = xcd_slot(blockIdx.x, tiles);
  if (threadIdx.x == 0) {
    int t = xs.outer, g = 0, found = -1, k0 = 0;
    for (; g < G; ++g) {
      const int rows = group_ptr[g + 1] - group_ptr[g];
      const int nc = (rows + kchunk - 1) / kchunk;
      if (t < nc) { found = g; k0 = group_ptr[g] + t * kchunk; break; }
      t -= nc;
    }
    s_pick[0] = found; s_pick[1] = k0;
  }
  __syncthreads();
  const int g = s_pick[0], k0 = s_pick[1];
  __syncthreads();   // the ring is about to be filled
  if (g < 0) return;
  const int kend = min(k0 + kchunk, group_ptr[g + 1]);
  float* Cg = C + (int64_t)g * strideC;
  if (det) { Cg = det + (int64_t)xs.outer * M * N; ldc = N; }
  gemm_tn32_tile<BK, NS>(A, lda, B, ldb, Cg, ldc, M, N, (xs.inner / n_tiles) * T32_BT, (xs.inner % n_tiles) * T32_BT, k0, kend,
                         ep, ring);
}

bool gemm_tn32_ok(int64_t M, int64_t N, const void* A, int64_t lda, const void* B, int64_t ldb) {
  return M % T32_BT == 0 && N % T32_BT == 0 && lda % 4 == 0 && ldb % 4 == 0 && (uintptr_t)A % 16 == 0 &&
         (uintptr_t)B % 16 == 0;
}

// 16-row stages x 4 slots = 64 KB of LDS: two blocks per CU, each with three 16 KB stages in flight
constexpr int T32_BK = 16, T32_NS = 4;

// C must be zeroed (atomic form) by the caller; `ep.atomic` / `det` as in the split-K forms of gemm_f32.hip
void launch_gemm_tn32_splitk(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                             int64_t ldc, int kchunk, int64_t nsplit, const GemmEpilogue& ep, float* det, hipStream_t s) {
  const int64_t tiles = (M / T32_BT) * (N / T32_BT);
  hipLaunchKernelGGL((gemm_tn32_splitk_kernel<T32_BK, T32_NS>), dim3(xcd_grid(nsplit, tiles)), dim3(256), 0, s, A, lda, B, ldb, C,
                     ldc, (int)M, (int)N, (int)K, kchunk, ep, det);
}

void launch_gemm_tn32_grouped(int64_t n_chunks, int64_t M, int64_t N, const float* A, int64_t lda, const float* B, int64_t ldb,
                              float* C, int64_t ldc, int64_t strideC, const int32_t* group_ptr, int G, int kchunk,
                              const GemmEpilogue& ep, float* det, hipStream_t s) {
  const int64_t tiles = (M / T32_BT) * (N / T32_BT);
  hipLaunchKernelGGL((gemm_tn32_grouped_kernel<T32_BK, T32_NS>), dim3(xcd_grid(n_chunks, tiles)), dim3(256), 0, s, A, lda, B, ldb,
                     C, ldc, strideC, group_ptr, G, (int)M, (int)N, kchunk, ep, det);
}

}  // namespace bkg
