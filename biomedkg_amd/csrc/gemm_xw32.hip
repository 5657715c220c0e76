// Exact-fp32 tall "activation x weight" products with K = N = 256:  C[M, 256] = epi(alpha * A[M, 256] @ op(W))
// (every hidden Linear / GCNConv.lin / RGCNConv root product of the parity mode, forward and input gradient).
//
// gemm_xw.hip's 8-wave block in fp32: wave w keeps W[:, 32 w : 32 w + 32] as 128 single-float MFMA operands in
// registers for its whole life (v_mfma_f32_32x32x2_f32: lane (l31, kh) supplies W[k + kh][n + l31]), LDS holds only
// A -- 64-row tiles (64 KB) copied by LDS-DMA, one whole 1-KiB row per wave-instruction, into two slots: the next tile
// lands while this one is multiplied (2 x 128 MFMAs of 64 clocks per wave: 13 us per tile, far longer than a load).
// Slot c' of row r holds 16-byte chunk c' ^ (r & 31) (on the DMA's source address), which spreads a fragment read --
// 32 rows, one k -- over 16 bank groups instead of one.  The sum over k runs in k order inside one accumulator: the
// same fmaf chain as the LDS-staged kernel of gemm_f32.hip (bit-identical results), at ~1.0 instead of ~0.5 of the
// fp32 matrix rate.
#include "common.h"
#include "gemm_epilogue.h"

namespace bkg {

typedef float x32_acc_t __attribute__((ext_vector_type(16)));

constexpr int X32_BM = 64, X32_K = 256, X32_N = 256, X32_NS = 2;
constexpr int X32_STAGE = X32_BM * X32_K * 4;   // 64 KB

struct Xw32Args {
  const float* A; int64_t lda;
  const float* W; int64_t ldw; int trans_w;   // trans_w: W is [N, K] (C = A W^T); else [K, N]
  float* C; int64_t ldc;
  int M;
};

__global__ __launch_bounds__(512) void gemm_xw32_kernel(Xw32Args g, GemmEpilogue ep) {
  __shared__ __attribute__((aligned(1024))) unsigned char ring[X32_NS * X32_STAGE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, kh = lane >> 5;
  const int n0 = wave * 32;
  const int m_tiles = (g.M + X32_BM - 1) / X32_BM, stride = gridDim.x;
  // Work items: the tiles of the full rounds, then -- when the last round would leave more than half of the blocks idle --
  // its tiles as two 32-row halves each (803 tiles on 256 blocks: 3 rounds + 35 tiles -> 70 halves: 3.5 tile-times
  // instead of 4).  Item t -> first row and whether it is a half (block-uniform).
  const int full_items = (m_tiles / stride) * stride, rest = m_tiles - full_items;
  const bool split_rest = 2 * rest <= stride;
  const int n_items = full_items + (split_rest ? 2 * rest : rest);
  auto row0_of = [&](int t) { return t < full_items || !split_rest ? t * X32_BM : full_items * X32_BM + (t - full_items) * 32; };

  // DMA pieces of this lane: rows wave * 8 + j of a tile (one row per wave-instruction), slot lane
  int t_issue = blockIdx.x;
  auto issue = [&](int slot) {
    unsigned char* base = ring + slot * X32_STAGE + wave * (8 * 1024);
    const int m0 = row0_of(t_issue);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = wave * 8 + j;
      const int r = max(0, min(m0 + row, g.M - 1));   // past the end: valid memory, never stored
      const float* src = g.A + (int64_t)r * g.lda + ((lane ^ (row & 31)) * 4);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(base + j * 1024), 16, 0, 0);
    }
    t_issue += stride;
  };
  issue(0);
  // the weights (and the bias) travel together with the first tile
  float w[X32_K / 2];
  if (g.trans_w && g.ldw % 4 == 0 && (reinterpret_cast<uintptr_t>(g.W) & 15) == 0) {
    // W[n][k]: this lane's row in 16-byte pieces (k, k + 2 -> two operands per piece; scalar loads would be 128 requests
    // of 32 x 4 bytes each)
    const float* row = g.W + (int64_t)(n0 + l31) * g.ldw;
#pragma unroll
    for (int q = 0; q < X32_K / 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(row + 4 * q);
      w[2 * q] = kh ? v.y : v.x;
      w[2 * q + 1] = kh ? v.w : v.z;
    }
  } else {
#pragma unroll
    for (int ks = 0; ks < X32_K / 2; ++ks)
      w[ks] = g.trans_w ? g.W[(int64_t)(n0 + l31) * g.ldw + 2 * ks + kh] : g.W[(int64_t)(2 * ks + kh) * g.ldw + n0 + l31];
  }
  // acc register r of a lane: row (r & 3) + 8 (r >> 2) + 4 kh of the 32-row block, column n0 + l31
  const float bias = ep.bias ? ep.bias[n0 + l31] : 0.f;

  int slot = 0;
  for (int t = blockIdx.x; t < n_items; t += stride) {
    const bool half = split_rest && t >= full_items;          // a 32-row item: the second row block is not computed
    const int m_row0 = row0_of(t), m_end = min(g.M, m_row0 + (half ? 32 : X32_BM));
    // this wave's DMAs of tile t have landed (older stores too); everyone's after the barrier, and the other slot is free
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue(slot ^ 1);
    const unsigned char* tile = ring + slot * X32_STAGE;
    x32_acc_t acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // A[row][k] sits at row * 1024 + ((k / 4) ^ (row & 31)) * 16 + (k % 4) * 4; row & 31 = l31 for both row blocks
    const unsigned char* arow = tile + l31 * 1024;
    if (!half) {
#pragma unroll
      for (int ks = 0; ks < X32_K / 2; ++ks) {
        const int k = 2 * ks;                                 // this lane reads k + kh: same 16-byte chunk (k even)
        const int off = (((k >> 2) ^ l31) << 4) + ((k & 3) + kh) * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float a = *reinterpret_cast<const float*>(arow + j * (32 * 1024) + off);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w[ks], acc[j], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < X32_K / 2; ++ks) {
        const int k = 2 * ks;
        const int off = (((k >> 2) ^ l31) << 4) + ((k & 3) + kh) * 4;
        const float a = *reinterpret_cast<const float*>(arow + off);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w[ks], acc[0], 0, 0, 0);
      }
    }
    // D = A_tile x W: lane holds column n0 + l31 of rows (r & 3) + 8 (r >> 2) + 4 kh
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m_row0 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (row < m_end) {
          float v = ep.alpha * acc[j][r] + bias;
          if (ep.act == 1) v = fmaxf(v, 0.f);
          else if (ep.act == 2) v = v > 0.f ? v : expm1f(v);
          g.C[(int64_t)row * g.ldc + n0 + l31] = v;
        }
      }
    slot ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the look-ahead DMA past the last tile
}

bool gemm_xw32_ok(int64_t M, int64_t N, int64_t K, const void* A, int64_t lda, bool plain_or_bias_act) {
  // tall inputs only: a block spends ~10 us loading its 128 weight registers and its first tile, which a mini-batch
  // (a few dozen tiles: the sampled regime) never earns back -- measured 877 -> 797 replayed steps/s when those took it
  return plain_or_bias_act && N == X32_N && K == X32_K && M >= 16384 && lda % 4 == 0 && (uintptr_t)A % 16 == 0;
}

void launch_gemm_xw32(int64_t M, const float* A, int64_t lda, const float* W, int64_t ldw, int trans_w, float* C, int64_t ldc,
                      const GemmEpilogue& ep, hipStream_t s) {
  Xw32Args g{A, lda, W, ldw, trans_w, C, ldc, (int)M};
  const unsigned blocks = (unsigned)std::min<int64_t>(256, ceil_div(M, X32_BM));
  hipLaunchKernelGGL(gemm_xw32_kernel, dim3(blocks), dim3(512), 0, s, g, ep);
}

}  // namespace bkg
