// Weight gradients of the bf16 mode when BOTH operands are bf16 tables:  C[M,N] += A[K,M]^T B[K,N]
// (dW = dXW^T X, dW_r = H_r^T dY_r: K = number of nodes / segments, M x N = a few 128 x 128 tiles).
//
// Both operands are stored with the CONTRACTED index as the row, which is the wrong way round for an MFMA operand
// (8 consecutive k per lane).  The fp32 form (gemm_tile_bf16.h) transposes 4 x 4 patches in registers while staging.
// With bf16 sources nothing has to be converted, so the tile is copied into LDS as it lies in memory -- 16 B per
// lane, rows of 256 B -- and the k-major fragments are produced by the LDS transpose read ds_read_b64_tr_b16
// (cdna_hip_programming.md T10): per 16-lane group a 4 (k) x 16 (m) block comes back column-major, i.e. each lane
// gets 4 consecutive k of ITS column; two reads make the 8-deep operand.  The 256-byte rows use the guide's chunk
// XOR ((row & 3) << 2 | (row >> 2) & 3), without which the four rows of a block share their banks (4-way).
// Result tiles are added with f32 atomics (split over K), as in the fp32 form.
#include "common.h"
#include "gemm_epilogue.h"

namespace bkg {

typedef float t16_acc_t __attribute__((ext_vector_type(16)));
typedef __bf16 t16_bf16x8_t __attribute__((ext_vector_type(8)));
typedef short t16_s4_t __attribute__((ext_vector_type(4)));

constexpr int T16_BT = 128;    // tile width (m and n)

__device__ __forceinline__ int t16_off(int row, int ch) {     // byte offset of 16-byte chunk ch of row `row`
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

// 4 consecutive k (rows r0 .. r0+3 chosen by the lane's q) of this lane's column, see T10
__device__ __forceinline__ t16_s4_t t16_tr_read(const unsigned char* img, int r0, int c0, int lane) {
  const int q = (lane & 15) >> 2, p = lane & 3;
  const unsigned char* a = img + t16_off(r0 + q, c0 + (p >> 1)) + 8 * (p & 1);
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) t16_s4_t*)a);
}

// ---- one block: rows [kbeg, kend) of A[:, m0 : m0+128] and B[:, n0 : n0+128]  ->  C[m0.., n0..] -------------------------
// Staging is LDS-DMA into a ring of stages.  (A register-staged loop -- load, barrier, ds_write, barrier, MFMA -- holds
// ONE 16 KB stage in flight per block and restarts the loads after two barriers: with three blocks per CU it was bound
// by the load latency, 3.2 us per 32-row stage at K = 130k = 1.9 TB/s of operand bytes.)  `global_load_lds_dwordx4`
// writes the stage image directly (no staging registers, so depth costs LDS
// only): NS - 1 stages are in flight per block across ONE raw barrier per stage, retired by counted vmcnt waits
// (cdna_hip_programming.md section 5, "Pipelining across barriers").  A DMA's LDS destination is lane-linear (1 KiB =
// four 256-byte rows per wave-instruction), so the chunk XOR of t16_off is applied to the SOURCE address: slot c' of
// row r receives chunk c' ^ f(r), which is where the transposed reads look for it (the XOR is an involution).
// Rows past the end of the K range come from a zero page (an LDS-DMA cannot mask a lane to zero).
__device__ uint4 t16_zero_page[16];              // 256 zero bytes

__device__ __forceinline__ void t16_dma(const void* src, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int BK, int NS> constexpr int t16_ring_bytes() { return NS * 2 * BK * 256; }

template <int BK, int NS, bool ROWS>
__device__ __forceinline__ void gemm_tn16_tile_ring(const __bf16* __restrict__ A, int64_t lda, const __bf16* __restrict__ B,
                                                    int64_t ldb, const int32_t* __restrict__ b_rows, float* __restrict__ C,
                                                    int64_t ldc, int M, int N, int m0, int n0, int kbeg, int kend,
                                                    const GemmEpilogue& ep, unsigned char* ring) {
  constexpr int STAGE = 2 * BK * 256;   // A image, then B image
  constexpr int NJ = BK / 16;           // DMAs per wave, stage and operand (four rows each)
  static_assert(t16_ring_bytes<BK, NS>() >= 4 * 16 * (64 + 4) * 4, "the epilogue patch lies over the ring");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  const int grp = lane >> 4, lh = grp >> 1, half = grp & 1;

  t16_acc_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // this lane's NJ pieces per operand and stage: rows wave * (BK / 4) + j * 4 + lane / 16 of the stage, slot lane % 16
  int srow[NJ], sch[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    srow[j] = wave * (BK / 4) + j * 4 + (lane >> 4);
    sch[j] = (lane & 15) ^ (((srow[j] & 3) << 2) | ((srow[j] >> 2) & 3));
  }
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(t16_zero_page) + 16 * (lane & 15);
  // running source pointers of the next stage to issue (the gathered form looks its B rows up instead)
  const __bf16* pa[NJ];
  const __bf16* pb[NJ];
  int knext = kbeg;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    pa[j] = A + (int64_t)(kbeg + srow[j]) * lda + m0 + sch[j] * 8;
    pb[j] = B + (ROWS ? 0 : (int64_t)(kbeg + srow[j]) * ldb) + n0 + sch[j] * 8;
  }
  const int64_t a_step = (int64_t)BK * lda, b_step = (int64_t)BK * ldb;
  // gathered form: the B row of this lane's piece in the NEXT stage to issue.  Four consecutive indices at a
  // wave-uniform address: scalar loads (an ordinary load here would make hipcc drain every DMA in flight with
  // vmcnt(0)), fetched one stage early so that their latency is behind the MFMAs.
  int ridx[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) ridx[j] = 0;
  auto fetch_rows = [&]() {
    if constexpr (ROWS) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int kq = __builtin_amdgcn_readfirstlane(knext + wave * (BK / 4) + j * 4), q = lane >> 4;
        int r = b_rows[min(kq, kend - 1)];
        const int r1 = b_rows[min(kq + 1, kend - 1)], r2 = b_rows[min(kq + 2, kend - 1)], r3 = b_rows[min(kq + 3, kend - 1)];
        r = q == 1 ? r1 : r;
        r = q == 2 ? r2 : r;
        ridx[j] = q == 3 ? r3 : r;
      }
    }
  };
  auto issue = [&](int slot) {
    unsigned char* base = ring + slot * STAGE + wave * (BK / 4 * 256);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const bool ok = knext + srow[j] < kend;
      const void* sa = ok ? (const void*)pa[j] : (const void*)zero;
      const void* sb;
      if constexpr (ROWS) {
        sb = ok ? (const void*)(pb[j] + (int64_t)ridx[j] * ldb) : (const void*)zero;
      } else {
        sb = ok ? (const void*)pb[j] : (const void*)zero;
        pb[j] += b_step;
      }
      pa[j] += a_step;
      t16_dma(sa, base + j * 1024);
      t16_dma(sb, base + BK * 256 + j * 1024);
    }
    knext += BK;
    fetch_rows();
  };
  fetch_rows();

  const int n_st = (kend - kbeg + BK - 1) / BK;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue(s);   // stages past the end read the zero page
  int slot = 0;
  for (int st = 0; st < n_st; ++st) {
    // this wave's DMAs of stage st have landed (the NS - 2 younger stages may still be in flight) ...
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * 2 * NJ) : "memory");
    // ... and so have everyone else's; every wave is also done reading stage st - 1, whose slot is refilled next
    __builtin_amdgcn_s_barrier();
    issue(slot == 0 ? NS - 1 : slot - 1);
    const unsigned char* As = ring + slot * STAGE;
    const unsigned char* Bs = As + BK * 256;
#pragma unroll
    for (int ks = 0; ks < BK; ks += 16) {
      t16_bf16x8_t a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ca = (wm0 + i * 32 + 16 * half) >> 3, cb = (wn0 + i * 32 + 16 * half) >> 3;
        const t16_s4_t a0 = t16_tr_read(As, ks + 8 * lh, ca, lane), a1 = t16_tr_read(As, ks + 8 * lh + 4, ca, lane);
        const t16_s4_t b0 = t16_tr_read(Bs, ks + 8 * lh, cb, lane), b1 = t16_tr_read(Bs, ks + 8 * lh + 4, cb, lane);
        typedef short s8_t __attribute__((ext_vector_type(8)));
        const s8_t av = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
        const s8_t bv = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
        a[i] = __builtin_bit_cast(t16_bf16x8_t, av);
        b[i] = __builtin_bit_cast(t16_bf16x8_t, bv);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    slot = slot + 1 == NS ? 0 : slot + 1;
  }
  // the epilogue's transpose patch lies over the ring: drain the zero-page DMAs still in flight, wait for every reader
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  store_acc_tile<2, 2>(acc, C, ldc, M, N, m0 + wm0, n0 + wn0, ep, reinterpret_cast<float*>(ring) + wave * (16 * (64 + 4)));
}

template <int BK, int NS>
__global__ __launch_bounds__(256) void gemm_tn16_splitk_kernel(const __bf16* __restrict__ A, int64_t lda,
                                                               const __bf16* __restrict__ B, int64_t ldb,
                                                               float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                               int kchunk, GemmEpilogue ep, float* __restrict__ det) {
  __shared__ __attribute__((aligned(1024))) unsigned char ring[t16_ring_bytes<BK, NS>()];
  const int n_tiles = N / T16_BT, tiles = n_tiles * (M / T16_BT);
  const XcdSlot t = xcd_slot(blockIdx.x, tiles);
  const int kbeg = t.outer * kchunk, kend = min(kbeg + kchunk, K);
  if (kbeg >= kend) return;
  if (det) { C = det + (int64_t)t.outer * M * N; ldc = N; }
  gemm_tn16_tile_ring<BK, NS, false>(A, lda, B, ldb, nullptr, C, ldc, M, N, (t.inner / n_tiles) * T16_BT,
                                     (t.inner % n_tiles) * T16_BT, kbeg, kend, ep, ring);
}

template <int BK, int NS>
__global__ __launch_bounds__(256) void gemm_tn16_grouped_kernel(const __bf16* __restrict__ A, int64_t lda,
                                                                const __bf16* __restrict__ B, int64_t ldb,
                                                                const int32_t* __restrict__ b_rows,
                                                                float* __restrict__ C, int64_t ldc, int64_t strideC,
                                                                const int32_t* __restrict__ group_ptr, int G, int M,
                                                                int N, int kchunk, GemmEpilogue ep, float* __restrict__ det) {
  // all LDS in ONE object: a second __shared__ beside a DMA-filled array can make hipcc drain the DMAs before every read
  __shared__ __attribute__((aligned(1024))) unsigned char ring[t16_ring_bytes<BK, NS>()];
  int* s_pick = reinterpret_cast<int*>(ring);
  const int n_tiles = N / T16_BT, tiles = n_tiles * (M / T16_BT);
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
  const int m0 = (xs.inner / n_tiles) * T16_BT, n0 = (xs.inner % n_tiles) * T16_BT;
  if (b_rows) gemm_tn16_tile_ring<BK, NS, true>(A, lda, B, ldb, b_rows, Cg, ldc, M, N, m0, n0, k0, kend, ep, ring);
  else gemm_tn16_tile_ring<BK, NS, false>(A, lda, B, ldb, nullptr, Cg, ldc, M, N, m0, n0, k0, kend, ep, ring);
}

// ring shapes: (rows per stage, slots)
#define BKG_TN16_CFG(cfg, call)            \
  switch (cfg) {                           \
    case 2: { call(64, 4); break; }        \
    default: { call(32, 3); break; }       \
  }

bool gemm_tn16_ok(int64_t M, int64_t N, const void* A, int64_t lda, const void* B, int64_t ldb) {
  return M % T16_BT == 0 && N % T16_BT == 0 && lda % 8 == 0 && ldb % 8 == 0 && (uintptr_t)A % 16 == 0 &&
         (uintptr_t)B % 16 == 0;
}

int launch_gemm_tn16_splitk(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, const void* B, int64_t ldb,
                            float* C, int64_t ldc, hipStream_t s) {
  GemmEpilogue ep{1.f, 0.f, nullptr, 0, nullptr, 0, 1.f, 1};
  const int64_t tiles = (M / T16_BT) * (N / T16_BT);
  // The result tiles are ADDED with float atomics, which the memory side executes at 1.3 TB/s chip-wide
  // (MI355X_MICROARCH.md "Global float atomics"): 64 KB per block.  Few tiles (their operand stream is short): one deep
  // block per CU (64-row stages x 4 slots, 96 KB in flight) keeps the adds at 17 MB; many tiles (every operand column
  // block is re-read from L2 by several tiles: the stream dominates): three shallower blocks per CU.
  // (measured at K = 130k, us: 256 x 256 45 against 60; 512 x 256 65 / 75; 256 x 768 147 / 88)
  const int cfg = tiles <= 8 ? 2 : 1;
  const int64_t blocks = tiles <= 8 ? 256 : 768;
  int64_t nsplit = std::min<int64_t>((K + 511) / 512, std::max<int64_t>(1, blocks / tiles));
  const int kchunk = (int)(((K + nsplit - 1) / nsplit + 63) / 64 * 64);
  nsplit = (K + kchunk - 1) / kchunk;
  float* det = nullptr;
  BKG_DET_SLABS(det, ep, nsplit, M, N, "bkg_gemm_tn_bf16");
#define BKG_TN16_SPLITK(BK, NS)                                                                                        \
  hipLaunchKernelGGL((gemm_tn16_splitk_kernel<BK, NS>), dim3(xcd_grid(nsplit, tiles)), dim3(256), 0, s, (const __bf16*)A, \
                     lda, (const __bf16*)B, ldb, C, ldc, (int)M, (int)N, (int)K, kchunk, ep, det)
  BKG_TN16_CFG(cfg, BKG_TN16_SPLITK)
#undef BKG_TN16_SPLITK
  if (det) splitk_reduce(det, C, ldc, 0, (int)M, (int)N, nullptr, 1, kchunk, (int)nsplit, s);
  return 0;
}

int launch_gemm_tn16_grouped(int64_t rows_total, int64_t M, int64_t N, const void* A, int64_t lda, const void* B,
                             int64_t ldb, const int32_t* b_rows, float* C, int64_t ldc, int64_t strideC, const int32_t* group_ptr, int G,
                             hipStream_t s) {
  GemmEpilogue ep{1.f, 0.f, nullptr, 0, nullptr, 0, 1.f, 1};
  const int64_t tiles = (M / T16_BT) * (N / T16_BT);
  const int cfg = 1;
  // rows per block: 64 KB of atomic adds each (see above), so long chunks at scale; short ones for a mini-batch
  // (300k rows over 30 skewed groups, us: 1024 rows 103, 2048 92, 3072 84 / 100 gathered, 4096 96 / 120)
  const int KCHUNK = (int)std::min<int64_t>(2048, std::max<int64_t>(256, (rows_total / 96 + 63) / 64 * 64));
  float* det = nullptr;
  BKG_DET_SLABS(det, ep, ceil_div(rows_total, KCHUNK) + G, M, N, "bkg_gemm_grouped_k_bf16");
#define BKG_TN16_GROUPED(BK, NS)                                                                                        \
  hipLaunchKernelGGL((gemm_tn16_grouped_kernel<BK, NS>), dim3(xcd_grid(ceil_div(rows_total, KCHUNK) + G, tiles)),        \
                     dim3(256), 0, s, (const __bf16*)A, lda, (const __bf16*)B, ldb, b_rows, C, ldc, strideC, group_ptr, G, \
                     (int)M, (int)N, KCHUNK, ep, det)
  BKG_TN16_CFG(cfg, BKG_TN16_GROUPED)
#undef BKG_TN16_GROUPED
  if (det) splitk_reduce(det, C, ldc, strideC, (int)M, (int)N, group_ptr, G, KCHUNK, 0, s);
  return 0;
}

}  // namespace bkg
