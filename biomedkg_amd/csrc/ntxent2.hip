// NT-Xent / GRACE InfoNCE with intra-view negatives -- v2: register-resident anchor rows, symmetric forward.
//
// Reference: PyGCL DualBranchContrast(InfoNCE(tau), "L2L", intraview_negs=True) called at
// /root/reference/biomedkg/gcl_module.py:171-173,186-190.  With Z = [normalize(h1); normalize(h2)] (M = 2N rows) that
// loss is exactly
//     L = 1/M * sum_i ( -z_i.z_p(i)/tau + log D_i ),   D_i = sum_{j != i} E_ij,  E_ij = exp(z_i.z_j / tau),
// p(i) = (i + N) mod M, and  dL/dz_i = 1/(M tau) * ( sum_{j != i} E_ij (1/D_i + 1/D_j) z_j  -  2 z_p(i) ).
// PyGCL materialises ~8 [N,2N] fp32 temporaries per direction; here the Gram tiles live only in MFMA accumulators.
//
// Structure (one workgroup = NW waves = a PANEL of 32*NW anchor rows i; the other side streams by in 64-row j-tiles):
//   * each wave keeps ITS 32 anchor rows as the B-operand fragments of the Gram product in registers for the whole
//     panel (fp32: pre-scaled by log2(e)/tau so that E = exp2(S) needs no multiply; bf16: unscaled, the scale is applied
//     to the fp32 accumulator), so the only LDS traffic is the j-tile, shared by all NW waves;
//   * j-tiles are double-buffered in LDS (row-major, padded / XOR-swizzled rows) with the next tile's global loads
//     in flight under the current tile's MFMAs: ONE barrier per tile;
//   * the tile is computed transposed, S^T[j][i] (rows j in the accumulator registers, anchor i on the lane), so that
//     (forward) the row sums over j are lane-local adds, and (backward) W^T = E (1/D_i + 1/D_j) is already the A operand
//     of the second product dZ_I += W Z_J (cdna_hip_programming.md "An accumulator tile as the next MFMA's operand"):
//     no LDS round trip for W; Z_J's B fragments come from the same LDS image (fp32: ds_read_b32 along the row;
//     bf16: ds_read_b64_tr_b16 on the T10 (b) image that also serves the row reads of the first product);
//   * FORWARD uses E's symmetry: only tiles at or right of the panel's own columns are computed (M^2 h instead of
//     2 M^2 h flops); a tile strictly right of the panel contributes its row sums to D_i and its COLUMN sums to D_j
//     (32-lane transpose-reduce butterfly, 31 shuffles per 32 values).  Work = the flattened upper-triangular list of
//     (panel, j-tile) items, cut into equal contiguous ranges for a resident grid; row sums are written per
//     (block, panel) segment, column sums per item, and a finishing kernel adds them in a FIXED order: no atomics,
//     bitwise repeatable;
//   * BACKWARD recomputes the tile (no symmetry: the transposed contribution would need an M x h float-atomic sum per
//     tile, which is bound by the 1.3 TB/s atomic rate and is not repeatable); grid = panels x j-splits, partial dZ
//     per split summed in split order by the finishing kernel, which also applies the -2 z_p(i) term and projects
//     through the row normalisation.
//   fp32 = v_mfma_f32_32x32x2_f32 (exact fp32, the 1e-5 loss bar of BASELINE configs[1]); bf16 operands =
//   v_mfma_f32_32x32x16_bf16 with fp32 exp / sums / accumulators (BASELINE configs[2] scale).
// No max-subtraction is needed (|cos| <= 1 so exp(s/tau) <= e^5 at tau = 0.2), matching PyGCL.
#include "common.h"

namespace bkg {

typedef float nx_acc_t __attribute__((ext_vector_type(16)));
typedef float nx_acc4_t __attribute__((ext_vector_type(4)));
typedef __bf16 nx_bf16x8_t __attribute__((ext_vector_type(8)));
typedef short nx_s4_t __attribute__((ext_vector_type(4)));
typedef short nx_s8_t __attribute__((ext_vector_type(8)));

constexpr int NX_BJ = 64;   // rows of a j-tile
#ifndef NX_FWD16_NW
#define NX_FWD16_NW 8        // waves per workgroup of the bf16 forward at h > 128 (experiment knob)
#endif
#ifndef NX_FWD16_X2
#define NX_FWD16_X2 1        // bf16 forward: 64 anchor rows per wave, LDS-DMA staging (ntx2_fwd16x2_kernel); 0: the generic kernel
#endif
#ifndef NX_X2_RD
#define NX_X2_RD 2           // fragment reads in flight ahead of their MFMAs in the x2 forward at h = 256 (2: 24 B of scratch)
#endif
#ifndef NX_X2_ASM
#define NX_X2_ASM 0          // 1: x2 forward with hand-issued fragment reads (asm ds_read_b128 + counted lgkmcnt) and the NEXT tile's
#endif                       // four LDS-DMAs issued BETWEEN the MFMAs of the matrix phase.  Round 5, N = 130k: 16.1 ms against 15.5 with
                             // hipcc's own schedule (0) -- correct, slower: the pinned order costs more than the DMA issue it hides
#ifndef NX_X2_ODD
#define NX_X2_ODD 0          // the x2 forward's barrier between a tile's two half-steps: 0 none (15.8 ms at N = 130k), 1 raw s_barrier (16.5), 2 __syncthreads (16.3)
#endif
#ifndef NX_BWD16_PIPE
#define NX_BWD16_PIPE 1        // bf16 backward at h > 128: the three-tile software pipeline (ntx2_bwd16p_kernel); 0: ntx2_kernel
#endif
#ifndef NX_PIPE_DS           // ntx2_bwd16p_kernel: instructions of each class issued behind every MFMA (sweep: docs/experiments.md 9-3)
#define NX_PIPE_DS 4
#endif
#ifndef NX_PIPE_VALU
#define NX_PIPE_VALU 4
#endif
#ifndef NX_PIPE_DMA_INSIDE
#define NX_PIPE_DMA_INSIDE 0     // (1: the tile DMAs issued between the MFMAs of the region: 79 vs 66 ms -- hipcc then drains them early)
#endif
#ifndef NX_PIPE_TRANS
#define NX_PIPE_TRANS 0         // (1: hipcc then puts the whole vector phase in front of the MFMAs: 69 -> slower)
#endif
#ifndef NX_PRIO_GRAM          // s_setprio at the start of a wave's matrix / vector phase in the x2 forward (-1: none)
#define NX_PRIO_GRAM -1
#endif
#ifndef NX_PRIO_VP
#define NX_PRIO_VP -1
#endif
#ifndef NX_FWD16_Q
#define NX_FWD16_Q 0          // 1: the bf16 forward on 16x16x32 tiles (ntx2_fwd16q_kernel: 16.1 - 16.3 ms at N = 130k against 15.7 - 15.9)
#endif
#ifndef NX_Q_RD
#define NX_Q_RD 1
#endif
#ifndef NX_LATE_RULE
#define NX_LATE_RULE 0
#endif
#ifndef NX_PINGPONG
#define NX_PINGPONG 1        // 8-wave bf16 forward: the two waves of a SIMD run half a tile apart (0: all waves in step; 2: fp32 too)
#endif
#ifndef NX_PP_PRIO
#define NX_PP_PRIO 0         // s_setprio of the later-dispatched half in the ping-pong forward (experiment knob)
#endif

struct NxArgs {
  const void* zn;      // [M][HPAD] normalised rows, zero padded columns (fp32 or bf16)
  int M;
  float scale;         // log2(e) / tau
  int n_panels, n_jtiles;
  const int* anchors;  // anchor a -> row of zn (NULL: a itself); the panels run over the n_anchor anchors
  int n_anchor;
  int rect;            // forward over an anchor SUBSET (row-partitioned multi-GPU): every j-tile, row sums only
  // forward
  long long n_items;   // flattened (panel, j-tile) items
  float* rowseg;       // [(grid + n_panels)][PR]   row sums per (block, panel) segment, slot = block + panel
  float* colitem;      // [n_items][64]             column sums per item (diagonal-panel items: not written, not read)
  unsigned int* ticket;  // nullable: zeroed by the forward launch for the loss reduce inside the finish launch
  // backward
  const float* rden;   // [M] 1 / D
  float* dzpart;       // [splits][M][HPAD]
  int splits, jt_per_split;
};

// ---- work list of the symmetric forward: panel P owns j-tiles [P*q, n_jtiles), q = PR / 64
__device__ __host__ __forceinline__ long long nx_start(long long P, int n_jtiles, int q) {
  return P * n_jtiles - (long long)q * (P * (P - 1) / 2);
}
__device__ __forceinline__ int nx_panel_of(long long x, int n_panels, int n_jtiles, int q) {
  int lo = 0, hi = n_panels - 1;   // largest P with start(P) <= x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (nx_start(mid, n_jtiles, q) <= x) lo = mid; else hi = mid - 1;
  }
  return lo;
}
// block that owns item x when the n_items are cut as lo(b) = floor(b * n_items / grid)
__device__ __host__ __forceinline__ long long nx_block_of(long long x, long long n_items, long long grid) {
  return ((x + 1) * grid - 1) / n_items;
}

// 32 values per lane (index v = 16 t + r of e[t][r]) -> lane q of each 32-lane half returns the half's sum of value q.
// Transpose-reduce: at every level a lane keeps the half of its values selected by one bit of its lane id and receives
// the partner's copy of that half.  Level "16" is one v_permlane16_swap (odd rows of e0 <-> even rows of e1) + one add
// per value pair, no select; the levels inside a 16-lane row use DPP partners (row_ror:8, row_half_mirror, quad_perm):
// no LDS crossbar round trips, so nothing waits on lgkmcnt (the ds_bpermute form spent ~600 exposed cycles per tile).
template <int CTRL>
__device__ __forceinline__ float nx_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// (TWO: the values are e0 + f0 / e1 + f1, the tiles of a wave's two anchor blocks, added pair by pair on the way in)
template <bool TWO>
__device__ __forceinline__ float nx_transpose_reduce32_of(const nx_acc_t& e0, const nx_acc_t& e1, const nx_acc_t& f0,
                                                          const nx_acc_t& f1, int l31) {
  float a[16], b[8], c[4], d[2];
#pragma unroll
  for (int r0 = 0; r0 < 16; r0 += 4) {
    // x' = {x.row0, y.row0, x.row2, y.row2}, y' = {x.row1, y.row1, x.row3, y.row3}.  Inline asm: on ROCm 7.2 the builtin
    // __builtin_amdgcn_permlane16_swap hands back the new first operand in BOTH result slots (scripts/dev_lanes.hip);
    // hipcc pads nothing around an asm statement, hence the s_nop on either side (cdna_hip_programming.md 5.7) -- of a
    // group of four swaps on independent registers, not of each one (32 s_nop per tile less)
    float x[4], y[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      x[q] = TWO ? e0[r0 + q] + f0[r0 + q] : e0[r0 + q];
      y[q] = TWO ? e1[r0 + q] + f1[r0 + q] : e1[r0 + q];
    }
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\t"
                 "v_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7\n\ts_nop 1"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
#pragma unroll
    for (int q = 0; q < 4; ++q) a[r0 + q] = x[q] + y[q];       // rows 0,2: value r; rows 1,3: value 16 + r
  }
  {
    const bool hi = l31 & 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float keep = hi ? a[r + 8] : a[r], send = hi ? a[r] : a[r + 8];
      b[r] = keep + nx_dpp<0x128>(send);                       // row_ror:8  (lane i <-> i ^ 8)
    }
  }
  {
    const bool hi = l31 & 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float keep = hi ? b[r + 4] : b[r], send = hi ? b[r] : b[r + 4];
      c[r] = keep + nx_dpp<0x141>(send);                       // row_half_mirror (i <-> i ^ 7: the other quad of the 8)
    }
  }
  {
    const bool hi = l31 & 2;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const float keep = hi ? c[r + 2] : c[r], send = hi ? c[r] : c[r + 2];
      d[r] = keep + nx_dpp<0x4E>(send);                        // quad_perm [2,3,0,1]
    }
  }
  const bool hi = l31 & 1;
  const float keep = hi ? d[1] : d[0], send = hi ? d[0] : d[1];
  return keep + nx_dpp<0xB1>(send);                            // quad_perm [1,0,3,2]
}
// 16 values per lane -> lane q of each 16-lane row returns the row's sum of value q (the levels 8, 4, 2, 1 of the above)
__device__ __forceinline__ float nx_transpose_reduce16(const float (&a)[16], int l15) {
  float b[8], c[4], d[2];
  {
    const bool hi = l15 & 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float keep = hi ? a[r + 8] : a[r], send = hi ? a[r] : a[r + 8];
      b[r] = keep + nx_dpp<0x128>(send);                       // row_ror:8
    }
  }
  {
    const bool hi = l15 & 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float keep = hi ? b[r + 4] : b[r], send = hi ? b[r] : b[r + 4];
      c[r] = keep + nx_dpp<0x141>(send);                       // row_half_mirror
    }
  }
  {
    const bool hi = l15 & 2;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const float keep = hi ? c[r + 2] : c[r], send = hi ? c[r] : c[r + 2];
      d[r] = keep + nx_dpp<0x4E>(send);                        // quad_perm [2,3,0,1]
    }
  }
  const bool hi = l15 & 1;
  const float keep = hi ? d[1] : d[0], send = hi ? d[0] : d[1];
  return keep + nx_dpp<0xB1>(send);                            // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float nx_transpose_reduce32(const nx_acc_t& e0, const nx_acc_t& e1, int l31) {
  return nx_transpose_reduce32_of<false>(e0, e1, e0, e1, l31);
}

__device__ __forceinline__ int nx_acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// T10 image (b): [slab of 128 columns][row][256 B], 16-byte chunk ch of a row XORed with the row's tag
__device__ __forceinline__ int nx_off16(int row, int ch) {
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

// ------------------------------------------------------------------------------------------------ main kernel
// BF16: operand type; HP = padded width / 32; NW waves; BWD.
template <bool BF16, int HP, int NW, bool BWD>
__global__ __launch_bounds__(NW * 64) void ntx2_kernel(NxArgs a) {
  if (a.ticket && blockIdx.x == 0 && threadIdx.x == 0) *a.ticket = 0u;   // (forward launches; the backward leaves it NULL)
  constexpr int HPAD = 32 * HP, PR = 32 * NW, NT = 64 * NW, Q = PR / NX_BJ;
  constexpr int HS = HPAD + 4;                                   // fp32 image: padded row (floats)
  // bf16 images: the forward only reads rows (ds_read_b128): rows padded by 16 B (conflict-free, every address a lane
  // constant + an immediate); the backward also reads columns (ds_read_b64_tr_b16): the T10 (b) XOR image
  constexpr bool PAD16 = BF16 && !BWD;
  constexpr int ROWB16 = HPAD * 2 + 16;
  constexpr int TILE_BYTES = BF16 ? (PAD16 ? NX_BJ * ROWB16 : NX_BJ * HPAD * 2) : NX_BJ * HS * 4;
  constexpr int CH_ROW = BF16 ? HPAD / 8 : HPAD / 4;             // 16-byte chunks per row
  constexpr int NPC = (NX_BJ * CH_ROW) / NT;                     // chunks staged per thread
  static_assert((NX_BJ * CH_ROW) % NT == 0, "tile chunks must divide over the threads");
  static_assert(!BF16 || HP % 4 == 0, "bf16 image is made of 128-column slabs");
  constexpr int KF = BF16 ? HPAD / 16 : HPAD / 8;                // anchor fragments per lane (uint4 / float4 each)

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  auto tile_at = [&](int b) { return smem + b * TILE_BYTES; };
  // Forward at 8 waves (two per SIMD): every tile is a matrix phase (the Gram product) followed by a vector phase (exp2,
  // row sums, the column-sum butterfly) of about the same length, and the per-tile barrier keeps the two waves of a SIMD
  // in the SAME phase -- they queue for the matrix pipe, then for the vector issue port (MI355X_MICROARCH.md "Two waves
  // per SIMD": both are per-SIMD resources).  PP: waves NW/2.. ("late" half, the second wave of each SIMD) finish tile
  // t-1's vector phase BEFORE tile t's matrix phase, so between two barriers a SIMD always has one wave on each pipe.
  constexpr bool PP = (NX_PINGPONG == 2 || (NX_PINGPONG == 1 && BF16)) && !BWD && NW == 8;   // (2: the fp32 forward too)
  constexpr int NSLOT = PP ? 3 : 2;                              // column-sum buffers in flight (PP: the late half writes a tile later)
  float* aux = reinterpret_cast<float*>(smem + 2 * TILE_BYTES);  // FWD: colbuf[NSLOT][NW][64]   BWD: rdj[2][64]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int M = a.M;
  const unsigned char* zn = static_cast<const unsigned char*>(a.zn);
  constexpr int ROW_BYTES = BF16 ? HPAD * 2 : HPAD * 4;

  // ---- this block's work
  long long x, x_end;
  int P, jt;
  if (BWD) {
    const int split = blockIdx.x % a.splits;
    P = blockIdx.x / a.splits;
    jt = split * a.jt_per_split;
    x = jt;
    x_end = min(jt + a.jt_per_split, a.n_jtiles);
    if (x >= x_end) return;
  } else {
    x = (long long)blockIdx.x * a.n_items / gridDim.x;
    x_end = (long long)(blockIdx.x + 1) * a.n_items / gridDim.x;
    if (x >= x_end) return;
    if (a.rect) {
      P = (int)(x / a.n_jtiles);
      jt = (int)(x % a.n_jtiles);
    } else {
      P = nx_panel_of(x, a.n_panels, a.n_jtiles, Q);
      jt = P * Q + (int)(x - nx_start(P, a.n_jtiles, Q));
    }
  }

  // ---- staging registers: the j-tiles in flight (and, backward, their 1/D values).  fp32 tiles take ~16x longer to
  //      consume than to fetch: one tile ahead; bf16 tiles are consumed in about a microsecond, less than an L2-miss round
  //      trip under load: TWO tiles ahead, in two register sets used alternately
  //      (not for the widest backward, whose accumulators leave no registers for a second set)
  //      -- measured: no gain from the second set (27 vs 28 ms forward at N = 130k, h = 256; the bf16 tile is bound by its
  //      exp / butterfly VALU work, not by the fetch), so DEPTH stays 1 everywhere; the two-set form is kept for a chip
  //      where the balance differs
  constexpr int DEPTH = 1;
  uint4 stA[NPC], stB[NPC];
  float rdA = 0.f, rdB = 0.f;
  auto prefetch = [&](int jt_, uint4 (&st)[NPC], float& st_rd) {
    const int j0 = jt_ * NX_BJ;
    if (j0 + NX_BJ <= M) {      // an interior tile (all but the last): unconditional loads, no per-chunk branch -- the
#pragma unroll                  // loads then sit in one scheduling region with the tile's MFMAs
      for (int i = 0; i < NPC; ++i) {
        const int p = tid + i * NT, row = p / CH_ROW, ch = p % CH_ROW;
        st[i] = *reinterpret_cast<const uint4*>(zn + (size_t)(j0 + row) * ROW_BYTES + ch * 16);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NPC; ++i) {
        const int p = tid + i * NT, row = p / CH_ROW, ch = p % CH_ROW;
        st[i] = make_uint4(0u, 0u, 0u, 0u);
        if (j0 + row < M) st[i] = *reinterpret_cast<const uint4*>(zn + (size_t)(j0 + row) * ROW_BYTES + ch * 16);
      }
    }
    if (BWD && tid < NX_BJ) st_rd = (j0 + tid < M) ? a.rden[j0 + tid] : 0.f;
  };
  auto commit = [&](int buf, const uint4 (&st)[NPC], float st_rd) {
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
      const int p = tid + i * NT, row = p / CH_ROW, ch = p % CH_ROW;
      if (PAD16) *reinterpret_cast<uint4*>(tile_at(buf) + row * ROWB16 + ch * 16) = st[i];
      else if (BF16) *reinterpret_cast<uint4*>(tile_at(buf) + (ch >> 4) * (NX_BJ * 256) + nx_off16(row, ch & 15)) = st[i];
      else *reinterpret_cast<uint4*>(tile_at(buf) + (row * HS + ch * 4) * 4) = st[i];
    }
    if (BWD && tid < NX_BJ) aux[buf * NX_BJ + tid] = st_rd;
  };

  // ---- anchor fragments of this wave's 32 rows
  uint4 zi[KF];
  float rd_i = 0.f;
  int i_glob = -1, i_lo = 0, i_hi = 0;      // this lane's anchor row (-1: none); the wave's row range
  auto load_anchor = [&](int P_) {
    const int a_idx = P_ * PR + wave * 32 + l31;
    i_glob = a_idx < a.n_anchor ? (a.anchors ? a.anchors[a_idx] : a_idx) : -1;
    int mn = i_glob < 0 ? 0x7fffffff : i_glob, mx = i_glob;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      mn = min(mn, __shfl_xor(mn, o, 64));
      mx = max(mx, __shfl_xor(mx, o, 64));
    }
    i_lo = __builtin_amdgcn_readfirstlane(mn);
    i_hi = __builtin_amdgcn_readfirstlane(mx);
#pragma unroll
    for (int k = 0; k < KF; ++k) {
      zi[k] = make_uint4(0u, 0u, 0u, 0u);
      if (i_glob >= 0) {
        // fp32: k values 8k + 4 lh + {0..3}; bf16: 16k + 8 lh + {0..7}  (16 bytes either way)
        zi[k] = *reinterpret_cast<const uint4*>(zn + (size_t)i_glob * ROW_BYTES + (2 * k + lh) * 16);
        if (!BF16) {
          float4 f = __builtin_bit_cast(float4, zi[k]);
          f.x *= a.scale; f.y *= a.scale; f.z *= a.scale; f.w *= a.scale;
          zi[k] = __builtin_bit_cast(uint4, f);
        }
      }
    }
    if (BWD) rd_i = (i_glob >= 0) ? a.rden[i_glob] : 0.f;
  };

  // T10-image addresses split into a lane constant, an XOR with a lane constant, and compile-time offsets, so that
  // a fragment read costs at most two VALU operations (the generic off(row, ch) form costs five):
  //   row read   (first product, row 32 t + l31, chunk 2k + lh):  rowA + ((32 (k & 7)) ^ rowT) + [8192 t + 16384 (k >> 3)]
  //   tr read    (second product, see below):                     trC[second] + ((64 (ct & 3)) ^ trQ) + [...]
  const int tag31 = ((l31 & 3) << 2) | ((l31 >> 2) & 3);
  const int rowA = 256 * l31 + 16 * (lh ^ (tag31 & 1)), rowT = 16 * (tag31 & 14);
  const int grp = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = grp & 1, g2 = grp >> 1;
  const int trL = 2 * g1 + (p4 >> 1), trQ = 64 * q4;
  const int trC[2] = {256 * (4 * g2 + q4) + 16 * (trL ^ (g2 & 3)) + 8 * (p4 & 1),
                      256 * (4 * g2 + q4) + 16 * (trL ^ ((g2 + 2) & 3)) + 8 * (p4 & 1)};
  const int rowP = l31 * ROWB16 + 16 * lh;                         // padded image: row 32 t + l31, chunk 2k + lh

  nx_acc_t dz[BWD ? HP : 1];
  if (BWD) {
#pragma unroll
    for (int t = 0; t < HP; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) dz[t][r] = 0.f;
  }
  float rowsum = 0.f;
  bool col_pending = false;     // forward: colbuf[buf ^ 1] holds the previous tile's column sums
  long long col_item = 0;

  auto item_after = [&](int& P_, int& jt_) {          // (panel, j-tile) of the next work item
    ++jt_;
    if (!BWD && jt_ == a.n_jtiles) { ++P_; jt_ = a.rect ? 0 : P_ * Q; }
  };
  prefetch(jt, stA, rdA);
  if (DEPTH == 2 && x + 1 < x_end) {
    int P1 = P, jt1 = jt;
    item_after(P1, jt1);
    prefetch(jt1, stB, rdB);
  }
  load_anchor(P);
  int buf = 0;

  // ---- S^T[j][i] for the two 32-row halves of the j-tile in LDS buffer buf_
  auto gram = [&](int buf_, nx_acc_t (&s)[2]) {
    if (!BF16) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
    }
    if (BF16) {
      // fragment reads run RD steps ahead of their MFMAs (an LDS read returns after ~100+ cycles, a 32x32x16 MFMA
      // issues every 32): the compiler's own schedule kept one step in flight and the matrix pipe waited on lgkmcnt
      constexpr int RD = (BWD && HP <= 4) ? 1 : 3, RING = RD + 1;
      uint4 ring[RING][2];
      auto read_k = [&](int k, uint4 (&dst)[2]) {
        const int at = PAD16 ? rowP + 32 * k : rowA + ((32 * (k & 7)) ^ rowT) + 16384 * (k >> 3);
#pragma unroll
        for (int t = 0; t < 2; ++t)
          dst[t] = *reinterpret_cast<const uint4*>(tile_at(buf_) + at + t * (PAD16 ? 32 * ROWB16 : 8192));
      };
#pragma unroll
      for (int k = 0; k < RD && k < KF; ++k) read_k(k, ring[k % RING]);
      const nx_acc_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KF; ++k) {
        if (k + RD < KF) read_k(k + RD, ring[(k + RD) % RING]);
        const nx_bf16x8_t b = __builtin_bit_cast(nx_bf16x8_t, zi[k]);
#pragma unroll
        for (int t = 0; t < 2; ++t)     // (k = 0: a literal zero C operand instead of 32 zeroed accumulator registers)
          s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(nx_bf16x8_t, ring[k % RING][t]), b,
                                                         k == 0 ? zero16 : s[t], 0, 0, 0);
      }
    } else {
      const float* zj = reinterpret_cast<const float*>(tile_at(buf_));
#pragma unroll
      for (int k = 0; k < KF; ++k) {
        const float4 b = __builtin_bit_cast(float4, zi[k]);
        const float4 a0 = *reinterpret_cast<const float4*>(zj + (l31)*HS + 8 * k + 4 * lh);
        const float4 a1 = *reinterpret_cast<const float4*>(zj + (32 + l31) * HS + 8 * k + 4 * lh);
        s[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, s[0], 0, 0, 0);
        s[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, s[1], 0, 0, 0);
        s[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, s[0], 0, 0, 0);
        s[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, s[1], 0, 0, 0);
        s[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, s[0], 0, 0, 0);
        s[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, s[1], 0, 0, 0);
        s[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, s[0], 0, 0, 0);
        s[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, s[1], 0, 0, 0);
      }
    }

  };
  // ---- E = exp2(S), masked where the tile touches the diagonal / the boundary; row sums (forward) or W (backward)
  auto expsum = [&](nx_acc_t (&s)[2], int j0, bool need_mask, int buf_) {
    const float* rdj = aux + buf_ * NX_BJ + 4 * lh;     // per-lane base: the row of register r is a compile-time offset from it
    // two copies of the loop, chosen ONCE per tile by the wave-uniform predicate: interior tiles (all but the few that
    // touch the diagonal or the boundary) run without the per-element compare / select / branch
    if (need_mask) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int jl = 32 * t + nx_acc_row(r, lh);
          float e = __builtin_amdgcn_exp2f(s[t][r]);     // both images carry the log2(e)/tau scale (bf16: sqrt on each side)
          const int jg = j0 + jl;
          e = (jg < M && i_glob >= 0 && jg != i_glob) ? e : 0.f;
          if (BWD) e *= rd_i + rdj[32 * t + nx_acc_row(r, 0)];
          else rowsum += e;
          s[t][r] = e;
        }
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float e = __builtin_amdgcn_exp2f(s[t][r]);
          if (BWD) e *= rd_i + rdj[32 * t + nx_acc_row(r, 0)];
          else rowsum += e;
          s[t][r] = e;
        }
    }

  };
  // ---- forward: this wave's column sums of the tile -> colbuf[slot_][wave]
  auto colwrite = [&](const nx_acc_t (&s)[2], int slot_) {
    const float v = nx_transpose_reduce32(s[0], s[1], l31);
    aux[slot_ * (NW * NX_BJ) + wave * NX_BJ + 32 * (l31 >> 4) + nx_acc_row(l31 & 15, lh)] = v;
  };
  // the waves' column sums of one item, added in wave order by the first 64 threads
  auto colreduce = [&](int slot_, long long item_) {
    const float* cb = aux + slot_ * (NW * NX_BJ);
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += cb[w * NX_BJ + tid];
    a.colitem[item_ * NX_BJ + tid] = v;
  };
  auto rowseg_write = [&](int P_) {
    const float tot = rowsum + __shfl_xor(rowsum, 32, 64);
    if (lh == 0) a.rowseg[((size_t)blockIdx.x + P_) * PR + wave * 32 + l31] = tot;
    rowsum = 0.f;
  };
  auto body = [&](uint4 (&st)[NPC], float& st_rd) {
    commit(buf, st, st_rd);
    __syncthreads();
    if (!BWD && col_pending && tid < NX_BJ) colreduce(buf ^ 1, col_item);      // previous tile's column sums
    // next item; the registers just committed take the tile DEPTH items ahead
    int P_next = P, jt_next = jt;
    item_after(P_next, jt_next);
    if (x + DEPTH < x_end) {
      int Pa = P_next, jta = jt_next;
      if (DEPTH == 2) item_after(Pa, jta);
      prefetch(jta, st, st_rd);
    }

    const int j0 = jt * NX_BJ;
    const bool need_mask = (j0 + NX_BJ > M) || (P * PR + wave * 32 + 32 > a.n_anchor) || (j0 <= i_hi && j0 + NX_BJ > i_lo);

    nx_acc_t s[2];
    gram(buf, s);
    expsum(s, j0, need_mask, buf);

    if (!BWD) {
      // column sums, unless the tile lies inside the panel's own columns (those pairs are also computed transposed)
      col_pending = !a.rect && jt >= (P + 1) * Q;
      if (col_pending) {
        colwrite(s, buf);
        col_item = x;
      }
    } else if (BF16) {
      // dZ[i][c] += sum_j W[j][i] Z_J[j][c]: accumulator registers 8 s2 .. 8 s2 + 7 are the A fragment of k-step s2;
      // its k order is permuted (element e of lane half g2 = row 16 s2 + 8 (e >> 2) + 4 g2 + (e & 3)), so the B
      // fragment is two transposed 4-row reads, rows r0 .. r0 + 3 and r0 + 8 .. r0 + 11, r0 = 32 t + 16 s2 + 4 g2
      nx_bf16x8_t wf[4];                                    // the four A fragments of this tile (t, s2)
#pragma unroll
      for (int ts = 0; ts < 4; ++ts)
#pragma unroll
        for (int e = 0; e < 8; ++e) wf[ts][e] = (__bf16)s[ts >> 1][8 * (ts & 1) + e];
      constexpr int N2 = 4 * HP, RD2 = HP <= 4 ? 2 : 3, RING2 = RD2 + 1;     // steps (ts, ct), reads RD2 steps ahead
      nx_s8_t bring[RING2];
      auto read_b = [&](int i) {
        const int ts = i / HP, ct = i % HP;
        const int col = ((64 * (ct & 3)) ^ trQ) + 16384 * (ct >> 2) + 256 * (32 * (ts >> 1) + 16 * (ts & 1));
        const nx_s4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) nx_s4_t*)(tile_at(buf) + trC[0] + col));
        const nx_s4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) nx_s4_t*)(tile_at(buf) + trC[1] + col + 256 * 8));
        return __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
      };
#pragma unroll
      for (int i = 0; i < RD2; ++i) bring[i % RING2] = read_b(i);
#pragma unroll
      for (int i = 0; i < N2; ++i) {
        if (i + RD2 < N2) bring[(i + RD2) % RING2] = read_b(i + RD2);
        dz[i % HP] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i / HP], __builtin_bit_cast(nx_bf16x8_t, bring[i % RING2]),
                                                             dz[i % HP], 0, 0, 0);
      }
    } else {
      // register r of W^T is the A operand for k = {row(r, lh = 0), row(r, lh = 1)} of the j-half t
      const float* zj = reinterpret_cast<const float*>(tile_at(buf));
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float* zrow = zj + (32 * t + nx_acc_row(r, lh)) * HS + l31;
#pragma unroll
          for (int ct = 0; ct < HP; ++ct)
            dz[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(s[t][r], zrow[32 * ct], dz[ct], 0, 0, 0);
        }
    }

    // ---- advance; forward: leaving the panel -> write its row sums, take the next panel's anchors
    const bool leave = (P_next != P) || (x + 1 == x_end);
    if (!BWD && leave) {
      rowseg_write(P);
      if (x + 1 < x_end) load_anchor(P_next);
    }
    P = P_next;
    jt = jt_next;
    buf ^= 1;
    ++x;
  };
  if constexpr (PP) {
    const bool late = wave >= NW / 2;
    if (late && NX_PP_PRIO) __builtin_amdgcn_s_setprio(NX_PP_PRIO);
    nx_acc_t s[2];
    bool owed = false, d_mask = false, d_col = false;      // late half: the tile whose vector phase is still owed
    int d_P = 0, d_j0 = 0, d_slot = 0;
    bool p1 = false, p2 = false;                           // column sums of items x-1 / x-2 still to be added over the waves
    long long it1 = 0, it2 = 0;
    int slot = 0;                                          // colbuf slot of item x (x mod 3 from this block's first item)
    while (x < x_end) {
      commit(buf, stA, rdA);
      __syncthreads();
      if (p2 && tid < NX_BJ) colreduce(slot == 2 ? 0 : slot + 1, it2);       // item x-2: both halves have written it
      int P_next = P, jt_next = jt;
      item_after(P_next, jt_next);
      if (x + 1 < x_end) prefetch(jt_next, stA, rdA);
      const int j0 = jt * NX_BJ;
      const bool col_now = !a.rect && jt >= (P + 1) * Q;
      if (!late) {
        const bool need_mask = (j0 + NX_BJ > M) || (P * PR + wave * 32 + 32 > a.n_anchor) || (j0 <= i_hi && j0 + NX_BJ > i_lo);
        gram(buf, s);
        expsum(s, j0, need_mask, buf);
        if (col_now) colwrite(s, slot);
        if (P_next != P || x + 1 == x_end) {
          rowseg_write(P);
          if (x + 1 < x_end) load_anchor(P_next);
        }
      } else {
        if (owed) {
          expsum(s, d_j0, d_mask, buf);
          if (d_col) colwrite(s, d_slot);
          if (d_P != P) {
            rowseg_write(d_P);
            load_anchor(P);
          }
        }
        d_mask = (j0 + NX_BJ > M) || (P * PR + wave * 32 + 32 > a.n_anchor) || (j0 <= i_hi && j0 + NX_BJ > i_lo);
        gram(buf, s);
        owed = true; d_P = P; d_j0 = j0; d_col = col_now; d_slot = slot;
      }
      p2 = p1; it2 = it1;
      p1 = col_now; it1 = x;
      P = P_next; jt = jt_next;
      buf ^= 1;
      slot = slot == 2 ? 0 : slot + 1;
      ++x;
    }
    if (late && owed) {
      expsum(s, d_j0, d_mask, buf);
      if (d_col) colwrite(s, d_slot);
      rowseg_write(d_P);
    }
    __syncthreads();
    if (tid < NX_BJ) {                                     // the last two items (slot = the slot the next item would take)
      if (p2) colreduce(slot == 2 ? 0 : slot + 1, it2);
      if (p1) colreduce(slot == 0 ? 2 : slot - 1, it1);
    }
    return;
  }

  while (x < x_end) {
    body(stA, rdA);
    if (DEPTH == 2) {
      if (x >= x_end) break;
      body(stB, rdB);
    }
  }

  if (!BWD) {
    __syncthreads();
    if (col_pending && tid < NX_BJ) colreduce(buf ^ 1, col_item);
  } else {
    // dz[ct][r]: row i = wave*32 + row(r, lh), column 32 ct + l31 of this split's partial
    const int split = blockIdx.x % a.splits;
    const int ib = (blockIdx.x / a.splits) * PR + wave * 32;
    float* out = a.dzpart + (size_t)split * a.n_anchor * HPAD;
#pragma unroll
    for (int ct = 0; ct < HP; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ia = ib + nx_acc_row(r, lh);
        if (ia < a.n_anchor) out[(size_t)ia * HPAD + 32 * ct + l31] = dz[ct][r];
      }
  }
}

template <int OFF> __device__ __forceinline__ uint4 nx_frag_asm(uint32_t addr) {
  uint4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N> __device__ __forceinline__ void nx_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);      // (an MFMA touches no memory: the clobber alone does not keep it behind the wait)
}

// ------------------------------------------------------------------------------------------------ bf16 forward, 64 anchors per wave
// The forward kernel above reads one 1 KB A fragment from LDS per 32x32x16 MFMA (the B fragment is the wave's 32 anchor
// rows in registers): 32 B per cycle and SIMD = 128 B per cycle and CU, exactly what the LDS delivers -- the Gram phase is
// LDS-bound as much as matrix-bound, and the exp / sum phase queues behind it (MFMA busy 0.41, 0.34 of the bf16 peak at
// N = 130k).  Here every wave keeps TWO 32-row anchor blocks in registers (128 VGPRs at h = 256), so each fragment read
// feeds two MFMAs: half the LDS traffic per flop, 512-row panels.  What pays for the registers: the j-tile is written to
// LDS by global_load_lds_dwordx4 (no staging registers; the T10 (b) XOR image, the swizzle applied to the SOURCE address
// since an LDS-DMA's destination is lane-linear -- cdna_hip_programming.md section 5).  The two waves of a SIMD run half
// a tile apart (PP above): one in its matrix phase (64 MFMAs), the other in its vector phase (64 exp2, the row sums, one
// column-sum butterfly over the SUM of its two blocks' tiles).
template <int HP>
__global__ __launch_bounds__(512) void ntx2_fwd16x2_kernel(NxArgs a) {
  if (a.ticket && blockIdx.x == 0 && threadIdx.x == 0) *a.ticket = 0u;   // (forward launches; the backward leaves it NULL)
  constexpr int NW = 8, AB = 2, HPAD = 32 * HP, PR = 32 * NW * AB, Q = PR / NX_BJ;
  constexpr int ROW_BYTES = HPAD * 2, TILE_BYTES = NX_BJ * ROW_BYTES, KF = HPAD / 16;
  constexpr int NG = TILE_BYTES / 1024 / NW;                     // LDS-DMA instructions per wave and tile (4 rows of a slab each)
  static_assert(HP % 4 == 0 && NG >= 1, "bf16 image is made of 128-column slabs");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  auto tile_at = [&](int b) { return smem + b * TILE_BYTES; };
  float* aux = reinterpret_cast<float*>(smem + 2 * TILE_BYTES);  // colbuf[3][NW][64]

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = tid >> 6;      // (left in a VGPR on purpose: with scalar control flow hipcc's schedule of this loop spills 400 B)
  const int M = a.M;
  const unsigned char* zn = static_cast<const unsigned char*>(a.zn);

  long long x = (long long)blockIdx.x * a.n_items / gridDim.x;
  const long long x_end = (long long)(blockIdx.x + 1) * a.n_items / gridDim.x;
  if (x >= x_end) return;
  int P, jt;
  if (a.rect) {
    P = (int)(x / a.n_jtiles);
    jt = (int)(x % a.n_jtiles);
  } else {
    P = nx_panel_of(x, a.n_panels, a.n_jtiles, Q);
    jt = P * Q + (int)(x - nx_start(P, a.n_jtiles, Q));
  }
  auto item_after = [&](int& P_, int& jt_) {
    ++jt_;
    if (jt_ == a.n_jtiles) { ++P_; jt_ = a.rect ? 0 : P_ * Q; }
  };

  // ---- j-tile -> LDS image [slab][row][256 B], chunk c of a row at position c ^ tag(row); rows past M: row M - 1 (masked).
  // DMA q = wave * NG + g of a tile writes rows 4 (q & 15) .. + 3 of slab q >> 4; lane: row 4 (q & 15) + (lane >> 4) -- whose
  // tag is ((lane >> 4) << 2) | (q & 3) -- position lane & 15.  Per lane that is ONE offset, XORed with 16 (q & 3) per DMA
  // (opaque to the optimiser on purpose: hoisted out of the tile loop, the NG variants would cost registers this kernel
  // does not have)
  const int st_row = lane >> 4;
  const unsigned st_lane = (unsigned)(st_row * ROW_BYTES + 16 * ((lane & 15) ^ (st_row << 2)));
  auto stage = [&](int jt_, int buf_) {
    const int j0 = jt_ * NX_BJ;
    unsigned off = st_lane;
    asm volatile("" : "+v"(off));
    const int ws = __builtin_amdgcn_readfirstlane(wave);     // scalar here: bases and LDS targets are SGPR arithmetic
    if (j0 + NX_BJ <= M) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int q = ws * NG + g, slab = q >> 4;
        const unsigned char* base = zn + (size_t)(j0 + 4 * (q & 15)) * ROW_BYTES + slab * 256;     // wave-uniform
        unsigned char* dst = tile_at(buf_) + slab * (NX_BJ * 256) + (q & 15) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (off ^ (16u * (q & 3)))),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int q = ws * NG + g, slab = q >> 4, row = 4 * (q & 15) + st_row;
        const int jr = min(j0 + row, M - 1);
        const unsigned char* src = zn + (size_t)jr * ROW_BYTES + slab * 256 + ((off ^ (16u * (q & 3))) & 255u);
        unsigned char* dst = tile_at(buf_) + slab * (NX_BJ * 256) + (q & 15) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
  };

  // ---- anchor fragments: block b of this wave = anchors P * PR + wave * 64 + 32 b + l31 (their row ids are recomputed
  //      where a masked tile needs them: this kernel has no registers to keep them in)
  uint4 zi[AB][KF];
  auto anchor_row = [&](int P_, int b) {
    const int a_idx = P_ * PR + wave * 64 + 32 * b + l31;
    return a_idx < a.n_anchor ? (a.anchors ? a.anchors[a_idx] : a_idx) : -1;
  };
  auto load_anchor = [&](int P_) {
#pragma unroll
    for (int b = 0; b < AB; ++b) {
      const int ig = anchor_row(P_, b);
#pragma unroll
      for (int k = 0; k < KF; ++k) {
        zi[b][k] = make_uint4(0u, 0u, 0u, 0u);
        if (ig >= 0) zi[b][k] = *reinterpret_cast<const uint4*>(zn + (size_t)ig * ROW_BYTES + (2 * k + lh) * 16);
      }
    }
  };

  // fragment k (chunk 2 (k & 7) + lh of slab k >> 3) of row 32 t + l31 sits at position (2 (k & 7)) ^ (tag ^ lh): ONE lane
  // constant, XORed with 32 (k & 7) per read (again kept from being hoisted into eight registers)
  const int tag31 = ((l31 & 3) << 2) | ((l31 >> 2) & 3);
  const int row_lc = 256 * l31 + 16 * (tag31 ^ lh);

  nx_acc_t s[AB][2];
  float rowsum[AB] = {0.f, 0.f};

  // (NX_X2_ASM) one of the NG DMAs of a FULL tile, issued from inside the matrix phase: the issue costs 100-185 cycles of
  // the wave's own chain when it stands in a block of its own (experiments 9-3); between two MFMAs the matrix pipe covers it
  const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  auto stage_one = [&](int jt_, int buf_, int g) {
    unsigned off = st_lane;
    asm volatile("" : "+v"(off));
    const int ws = __builtin_amdgcn_readfirstlane(wave);
    const int q = ws * NG + g, slab = q >> 4;
    const unsigned char* base = zn + (size_t)(jt_ * NX_BJ + 4 * (q & 15)) * ROW_BYTES + slab * 256;     // wave-uniform
    unsigned char* dst = tile_at(buf_) + slab * (NX_BJ * 256) + (q & 15) * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (off ^ (16u * (q & 3)))),
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  // the matrix phase with hand-issued fragment reads: RD k-steps (two reads each) ahead, a counted wait in front of every
  // k-step's four MFMAs; `jt_next >= 0`: the next tile's DMAs of this wave go out behind k-steps 1, 5, 9, 13
  auto gram_asm = [&](int buf_, int jt_next) {
    constexpr int RD = HP > 4 ? NX_X2_RD : 3, RING = RD + 1;
    uint4 ring[RING][2];
    uint32_t lc = (uint32_t)(row_lc + buf_ * TILE_BYTES);
    asm volatile("" : "+v"(lc));
    const nx_acc_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define NX_READ_K(K, DST)                                                                       \
    do {                                                                                        \
      const uint32_t at_ = smem_lds + (lc ^ (uint32_t)(32 * ((K) & 7)));                        \
      (DST)[0] = nx_frag_asm<16384 * ((K) >> 3)>(at_);                                          \
      (DST)[1] = nx_frag_asm<16384 * ((K) >> 3) + 8192>(at_);                                   \
    } while (0)
#define NX_STEP(K)                                                                              \
    do {                                                                                        \
      if constexpr ((K) + RD < KF) NX_READ_K((K) + RD, ring[((K) + RD) % RING]);                \
      nx_lgkm<2 * ((KF - 1 - (K)) < RD ? (KF - 1 - (K)) : RD)>();                               \
      _Pragma("unroll") for (int b = 0; b < AB; ++b)                                            \
        _Pragma("unroll") for (int t = 0; t < 2; ++t)                                           \
          s[b][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(nx_bf16x8_t, ring[(K) % RING][t]), \
                                                            __builtin_bit_cast(nx_bf16x8_t, zi[b][(K)]),           \
                                                            (K) == 0 ? zero16 : s[b][t], 0, 0, 0);                 \
      if constexpr (((K) & 3) == 1 && ((K) >> 2) < NG) { if (jt_next >= 0) stage_one(jt_next, buf_ ^ 1, (K) >> 2); } \
    } while (0)
    if constexpr (RD >= 1) NX_READ_K(0, ring[0]);
    if constexpr (RD >= 2) NX_READ_K(1, ring[1 % RING]);
    if constexpr (RD >= 3) NX_READ_K(2, ring[2 % RING]);
    static_assert(RD <= 3 && KF <= 16 && NG <= 4, "gram_asm: unrolled by hand for h <= 256");
    NX_STEP(0); NX_STEP(1); NX_STEP(2); NX_STEP(3);
    if constexpr (KF > 4) { NX_STEP(4); NX_STEP(5); NX_STEP(6); NX_STEP(7); }
    if constexpr (KF > 8) { NX_STEP(8); NX_STEP(9); NX_STEP(10); NX_STEP(11); }
    if constexpr (KF > 12) { NX_STEP(12); NX_STEP(13); NX_STEP(14); NX_STEP(15); }
#undef NX_STEP
#undef NX_READ_K
  };

  auto gram = [&](int buf_) {
    constexpr int RD = HP > 4 ? NX_X2_RD : 3, RING = RD + 1;      // (h = 256: the two anchor blocks leave little room for reads ahead)
    uint4 ring[RING][2];
    int lc = row_lc + buf_ * TILE_BYTES;
    asm volatile("" : "+v"(lc));
    auto read_k = [&](int k, uint4 (&dst)[2]) {
      const int at = (lc ^ (32 * (k & 7))) + 16384 * (k >> 3);
#pragma unroll
      for (int t = 0; t < 2; ++t) dst[t] = *reinterpret_cast<const uint4*>(smem + at + t * 8192);
    };
#pragma unroll
    for (int k = 0; k < RD && k < KF; ++k) read_k(k, ring[k % RING]);
    const nx_acc_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KF; ++k) {
      if (k + RD < KF) read_k(k + RD, ring[(k + RD) % RING]);
#pragma unroll
      for (int b = 0; b < AB; ++b)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          s[b][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(nx_bf16x8_t, ring[k % RING][t]),
                                                            __builtin_bit_cast(nx_bf16x8_t, zi[b][k]),
                                                            k == 0 ? zero16 : s[b][t], 0, 0, 0);
    }
  };
  // E = exp2(S) (both images carry sqrt(log2(e) / tau)), masked on the diagonal / at the boundary; row sums; with `col`
  // the column sums over the wave's 64 anchors -> colbuf[slot_][wave]
  auto vector_phase = [&](int P_, int j0, bool need_mask, bool col, int slot_) {
#pragma unroll
    for (int b = 0; b < AB; ++b) {
      if (need_mask) {
        const int ig = anchor_row(P_, b);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int jg = j0 + 32 * t + nx_acc_row(r, lh);
            float e = __builtin_amdgcn_exp2f(s[b][t][r]);
            e = (jg < M && ig >= 0 && jg != ig) ? e : 0.f;
            rowsum[b] += e;
            s[b][t][r] = e;
          }
      } else {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
#if defined(NX_EXP_NOEXP)
            const float e = s[b][t][r] * 1.0001f;
#else
            const float e = __builtin_amdgcn_exp2f(s[b][t][r]);
#endif
#if !defined(NX_EXP_NOROWSUM)
            rowsum[b] += e;
#endif
            s[b][t][r] = e;
          }
      }
    }
    if (col) {
      const float v = nx_transpose_reduce32_of<true>(s[0][0], s[0][1], s[1][0], s[1][1], l31);
      aux[slot_ * (NW * NX_BJ) + wave * NX_BJ + 32 * (l31 >> 4) + nx_acc_row(l31 & 15, lh)] = v;
    }
  };
  auto colreduce = [&](int slot_, long long item_) {
    const float* cb = aux + slot_ * (NW * NX_BJ);
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += cb[w * NX_BJ + tid];
    a.colitem[item_ * NX_BJ + tid] = v;
  };
  auto rowseg_write = [&](int P_) {
#pragma unroll
    for (int b = 0; b < AB; ++b) {
      const float tot = rowsum[b] + __shfl_xor(rowsum[b], 32, 64);
      if (lh == 0) a.rowseg[((size_t)blockIdx.x + P_) * PR + wave * 64 + 32 * b + l31] = tot;
      rowsum[b] = 0.f;
    }
  };
  // does the tile touch the boundary (rows past M, anchors past n_anchor) or the wave's own diagonal?  (an anchor LIST
  // -- the rectangular forward of a row-partitioned run -- has no range to test: always masked)
  auto mask_of = [&](int P_, int j0) {
    const int i0 = P_ * PR + wave * 64;
    return (j0 + NX_BJ > M) || (i0 + 64 > a.n_anchor) || a.anchors != nullptr || (j0 < i0 + 64 && j0 + NX_BJ > i0);
  };

  // ---- the loop runs over HALF-steps h: a wave is in its matrix phase (gram) on the even ones and in its vector phase
  //      on the odd ones of ITS OWN count hh = h - late, so that the two waves of a SIMD are always in different phases.
  //      Block-level duties sit on the even half-steps h = 2k: the column sums of item k - 2 (early waves wrote them at
  //      h = 2k - 3, late ones at 2k - 2) are added over the waves, and tile k + 1 is sent to the buffer tile k - 1 was
  //      last read from at h = 2k - 1.
  const int n_it = (int)(x_end - x);
  const bool late = NX_LATE_RULE == 0 ? wave >= NW / 2 : (NX_LATE_RULE == 1 ? (wave & 1) != 0 : ((wave >> 1) & 1) != 0);
  if (late && NX_PP_PRIO) __builtin_amdgcn_s_setprio(NX_PP_PRIO);
  int Pe = P, jte = jt;                                    // block timeline: item k at h = 2k
  bool p0 = false, p1 = false, p2 = false;                 // column-sum flags of items k, k - 1, k - 2
  int slot_e = 0;
  int kw = 0;                                              // this wave's own item (P, jt; the kw-th of the block), its buffer and colbuf slot
  int buf = 0, slot = 0;
  stage(jt, 0);
  load_anchor(P);
  for (int h = 0; h <= 2 * n_it; ++h) {
    // even h: the full barrier (the compiler drains this wave's LDS-DMAs in front of it: tile k is complete); odd h: the
    // waves only change phase -- a raw s_barrier, so that tile k + 1's DMAs stay in flight across it (NX_X2_ODD: 0 none,
    // 1 raw, 2 full)
    if ((h & 1) == 0 || NX_X2_ODD == 2) __syncthreads();
    else if (NX_X2_ODD == 1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if ((h & 1) == 0) {
      const int k = h >> 1;
      p2 = p1; p1 = p0;
      p0 = k < n_it && !a.rect && jte >= (Pe + 1) * Q;
      if (p2 && tid < NX_BJ) colreduce(slot_e == 0 ? 1 : (slot_e == 1 ? 2 : 0), x + k - 2);   // slot of k - 2 = (k + 1) mod 3
      if (k + 1 < n_it) {
        item_after(Pe, jte);
#if !defined(NX_EXP_NODMA)
        // (NX_X2_ASM: every wave sends ITS pieces of tile k + 1 from inside its own matrix phase of tile k -- the early
        // waves in this half-step, the late ones in the next -- unless the tile is a partial one: the row-clamped form)
        if (!NX_X2_ASM || (jte + 1) * NX_BJ > M) stage(jte, (int)((k + 1) & 1));
#endif
      }
      slot_e = slot_e == 2 ? 0 : slot_e + 1;
    }
    const int hh = h - (late ? 1 : 0);
    if (hh < 0 || hh >= 2 * n_it) continue;
    const int j0 = jt * NX_BJ;
    if ((hh & 1) == 0) {
      if (NX_PRIO_GRAM >= 0) __builtin_amdgcn_s_setprio(NX_PRIO_GRAM);
#if !defined(NX_EXP_NOGRAM)
      if constexpr (NX_X2_ASM != 0) {
        int jt_n = -1;
        if (kw + 1 < n_it) {
          int P_n = P;
          jt_n = jt;
          item_after(P_n, jt_n);
          if ((jt_n + 1) * NX_BJ > M) jt_n = -1;          // a partial tile: staged by the block's row-clamped form
        }
        gram_asm(buf, jt_n);
      } else {
        gram(buf);
      }
#endif
    } else {
      if (NX_PRIO_VP >= 0) __builtin_amdgcn_s_setprio(NX_PRIO_VP);
#if defined(NX_EXP_NOVP)
      rowsum[0] += s[0][0][0] + s[1][1][3];
#elif defined(NX_EXP_NOCOL)
      vector_phase(P, j0, mask_of(P, j0), false, slot);
#else
      vector_phase(P, j0, mask_of(P, j0), !a.rect && jt >= (P + 1) * Q, slot);
#endif
      int P_next = P, jt_next = jt;
      item_after(P_next, jt_next);
      const bool last = kw + 1 == n_it;
      if (P_next != P || last) {
        rowseg_write(P);
        if (!last) load_anchor(P_next);
      }
      P = P_next; jt = jt_next;
      ++kw;
      buf ^= 1;
      slot = slot == 2 ? 0 : slot + 1;
    }
  }
  __syncthreads();
  // the last item (early waves wrote it at h = 2 n - 1, late ones at 2 n); slot_e = (n + 1) mod 3, the item's = (n - 1) mod 3
  if (p1 && tid < NX_BJ) colreduce(slot_e == 0 ? 1 : (slot_e == 1 ? 2 : 0), x + n_it - 1);
}

// ------------------------------------------------------------------------------------------------ bf16 forward, 16x16x32 tiles
// ntx2_fwd16x2_kernel with the Gram tile of a wave (64 j x 64 anchors) made of 4 x 4 v_mfma_f32_16x16x32_bf16 tiles instead
// of 2 x 2 32x32x16 ones: the same flops and cycles, a quarter of the LDS bytes per MFMA cycle, a column butterfly over 16
// lanes instead of 32 -- and the clock: on random data the chip holds a higher one under the 16x16x32 shape
// (MI355X_MICROARCH.md, "DVFS give-back" item 7).  Everything around the tile (work list, LDS-DMA staging, half-step loop,
// column-sum buffers) is the kernel above.
template <int HP>
__global__ __launch_bounds__(512) void ntx2_fwd16q_kernel(NxArgs a) {
  if (a.ticket && blockIdx.x == 0 && threadIdx.x == 0) *a.ticket = 0u;   // (forward launches; the backward leaves it NULL)
  constexpr int NW = 8, IB = 4, JB = 4, HPAD = 32 * HP, PR = 16 * IB * NW, Q = PR / NX_BJ;
  constexpr int ROW_BYTES = HPAD * 2, TILE_BYTES = NX_BJ * ROW_BYTES, KS = HPAD / 32;
  constexpr int NG = TILE_BYTES / 1024 / NW;                     // LDS-DMA instructions per wave and tile (4 rows of a slab each)
  static_assert(HP % 4 == 0 && NG >= 1, "bf16 image is made of 128-column slabs");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  auto tile_at = [&](int b) { return smem + b * TILE_BYTES; };
  float* aux = reinterpret_cast<float*>(smem + 2 * TILE_BYTES);  // colbuf[3][NW][64]

  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
  const int wave = tid >> 6;      // (left in a VGPR on purpose: with scalar control flow hipcc's schedule of this loop spills 400 B)
  const int M = a.M;
  const unsigned char* zn = static_cast<const unsigned char*>(a.zn);

  long long x = (long long)blockIdx.x * a.n_items / gridDim.x;
  const long long x_end = (long long)(blockIdx.x + 1) * a.n_items / gridDim.x;
  if (x >= x_end) return;
  int P, jt;
  if (a.rect) {
    P = (int)(x / a.n_jtiles);
    jt = (int)(x % a.n_jtiles);
  } else {
    P = nx_panel_of(x, a.n_panels, a.n_jtiles, Q);
    jt = P * Q + (int)(x - nx_start(P, a.n_jtiles, Q));
  }
  auto item_after = [&](int& P_, int& jt_) {
    ++jt_;
    if (jt_ == a.n_jtiles) { ++P_; jt_ = a.rect ? 0 : P_ * Q; }
  };

  // ---- j-tile -> LDS image [slab][row][256 B], chunk c of a row at position c ^ tag(row); rows past M: row M - 1 (masked).
  // DMA q = wave * NG + g of a tile writes rows 4 (q & 15) .. + 3 of slab q >> 4; lane: row 4 (q & 15) + (lane >> 4) -- whose
  // tag is ((lane >> 4) << 2) | (q & 3) -- position lane & 15.  Per lane that is ONE offset, XORed with 16 (q & 3) per DMA
  // (opaque to the optimiser on purpose: hoisted out of the tile loop, the NG variants would cost registers this kernel
  // does not have)
  const int st_row = lane >> 4;
  const unsigned st_lane = (unsigned)(st_row * ROW_BYTES + 16 * ((lane & 15) ^ (st_row << 2)));
  auto stage = [&](int jt_, int buf_) {
    const int j0 = jt_ * NX_BJ;
    unsigned off = st_lane;
    asm volatile("" : "+v"(off));
    const int ws = __builtin_amdgcn_readfirstlane(wave);     // scalar here: bases and LDS targets are SGPR arithmetic
    if (j0 + NX_BJ <= M) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int q = ws * NG + g, slab = q >> 4;
        const unsigned char* base = zn + (size_t)(j0 + 4 * (q & 15)) * ROW_BYTES + slab * 256;     // wave-uniform
        unsigned char* dst = tile_at(buf_) + slab * (NX_BJ * 256) + (q & 15) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + (off ^ (16u * (q & 3)))),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int q = ws * NG + g, slab = q >> 4, row = 4 * (q & 15) + st_row;
        const int jr = min(j0 + row, M - 1);
        const unsigned char* src = zn + (size_t)jr * ROW_BYTES + slab * 256 + ((off ^ (16u * (q & 3))) & 255u);
        unsigned char* dst = tile_at(buf_) + slab * (NX_BJ * 256) + (q & 15) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
  };

  // ---- anchor fragments (B operands of v_mfma_f32_16x16x32_bf16: lane = column n = anchor, k = 8 (lane >> 4) + 0..7):
  //      block ib of this wave = anchors P * PR + wave * 64 + 16 ib + l15
  uint4 zi[IB][KS];
  auto anchor_row = [&](int P_, int ib) {
    const int a_idx = P_ * PR + wave * 64 + 16 * ib + l15;
    return a_idx < a.n_anchor ? (a.anchors ? a.anchors[a_idx] : a_idx) : -1;
  };
  auto load_anchor = [&](int P_) {
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) {
      const int ig = anchor_row(P_, ib);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        zi[ib][ks] = make_uint4(0u, 0u, 0u, 0u);
        if (ig >= 0) zi[ib][ks] = *reinterpret_cast<const uint4*>(zn + (size_t)ig * ROW_BYTES + (4 * ks + lg) * 16);
      }
    }
  };

  // A fragment of k-step ks, j-block jb: row 16 jb + l15 (its tag = the tag of l15), chunk 4 (ks & 3) + lg of slab ks >> 2 at
  // position chunk ^ tag: ONE lane constant, XORed with 64 (ks & 3) per read; 16 rows x 4 chunks per read = all 64 banks
  const int tag15 = ((l15 & 3) << 2) | ((l15 >> 2) & 3);
  const int row_lc = 256 * l15 + 16 * (lg ^ tag15);

  nx_acc4_t s[IB][JB];
  float rowsum[IB] = {0.f, 0.f, 0.f, 0.f};

  auto gram = [&](int buf_) {
    constexpr int RD = NX_Q_RD, RING = RD + 1;
    uint4 ring[RING][JB];
    int lc = row_lc + buf_ * TILE_BYTES;
    asm volatile("" : "+v"(lc));
    auto read_k = [&](int ks, uint4 (&dst)[JB]) {
      const int at = (lc ^ (64 * (ks & 3))) + 16384 * (ks >> 2);
#pragma unroll
      for (int jb = 0; jb < JB; ++jb) dst[jb] = *reinterpret_cast<const uint4*>(smem + at + jb * 4096);
    };
#pragma unroll
    for (int ks = 0; ks < RD && ks < KS; ++ks) read_k(ks, ring[ks % RING]);
    const nx_acc4_t zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + RD < KS) read_k(ks + RD, ring[(ks + RD) % RING]);
#pragma unroll
      for (int ib = 0; ib < IB; ++ib)
#pragma unroll
        for (int jb = 0; jb < JB; ++jb)
          s[ib][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(nx_bf16x8_t, ring[ks % RING][jb]),
                                                              __builtin_bit_cast(nx_bf16x8_t, zi[ib][ks]),
                                                              ks == 0 ? zero4 : s[ib][jb], 0, 0, 0);
    }
  };
  // E = exp2(S) (both images carry sqrt(log2(e) / tau)), masked on the diagonal / at the boundary; row sums; with `col`
  // the column sums over the wave's 64 anchors -> colbuf[slot_][wave].  s[ib][jb][r]: j = 16 jb + 4 lg + r, anchor 16 ib + l15
  auto vector_phase = [&](int P_, int j0, bool need_mask, bool col, int slot_) {
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) {
      if (need_mask) {
        const int ig = anchor_row(P_, ib);
#pragma unroll
        for (int jb = 0; jb < JB; ++jb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int jg = j0 + 16 * jb + 4 * lg + r;
            float e = __builtin_amdgcn_exp2f(s[ib][jb][r]);
            e = (jg < M && ig >= 0 && jg != ig) ? e : 0.f;
            rowsum[ib] += e;
            s[ib][jb][r] = e;
          }
      } else {
#pragma unroll
        for (int jb = 0; jb < JB; ++jb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(s[ib][jb][r]);
            rowsum[ib] += e;
            s[ib][jb][r] = e;
          }
      }
    }
    if (col) {
      float c[16];                                         // value v = 4 jb + r: the tile's column j = 16 jb + 4 lg + r
#pragma unroll
      for (int jb = 0; jb < JB; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) c[4 * jb + r] = (s[0][jb][r] + s[1][jb][r]) + (s[2][jb][r] + s[3][jb][r]);
      const float v = nx_transpose_reduce16(c, l15);       // lane l15 of each 16-lane row: the row's sum of value l15
      aux[slot_ * (NW * NX_BJ) + wave * NX_BJ + 16 * (l15 >> 2) + 4 * lg + (l15 & 3)] = v;
    }
  };
  auto colreduce = [&](int slot_, long long item_) {
    const float* cb = aux + slot_ * (NW * NX_BJ);
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += cb[w * NX_BJ + tid];
    a.colitem[item_ * NX_BJ + tid] = v;
  };
  auto rowseg_write = [&](int P_) {
#pragma unroll
    for (int ib = 0; ib < IB; ++ib) {
      float tot = rowsum[ib] + __shfl_xor(rowsum[ib], 16, 64);
      tot += __shfl_xor(tot, 32, 64);
      if (lg == 0) a.rowseg[((size_t)blockIdx.x + P_) * PR + wave * 64 + 16 * ib + l15] = tot;
      rowsum[ib] = 0.f;
    }
  };
  // does the tile touch the boundary (rows past M, anchors past n_anchor) or the wave's own diagonal?  (an anchor LIST
  // -- the rectangular forward of a row-partitioned run -- has no range to test: always masked)
  auto mask_of = [&](int P_, int j0) {
    const int i0 = P_ * PR + wave * 64;
    return (j0 + NX_BJ > M) || (i0 + 64 > a.n_anchor) || a.anchors != nullptr || (j0 < i0 + 64 && j0 + NX_BJ > i0);
  };

  // ---- the loop runs over HALF-steps h: a wave is in its matrix phase (gram) on the even ones and in its vector phase
  //      on the odd ones of ITS OWN count hh = h - late, so that the two waves of a SIMD are always in different phases.
  //      Block-level duties sit on the even half-steps h = 2k: the column sums of item k - 2 (early waves wrote them at
  //      h = 2k - 3, late ones at 2k - 2) are added over the waves, and tile k + 1 is sent to the buffer tile k - 1 was
  //      last read from at h = 2k - 1.
  const int n_it = (int)(x_end - x);
  const bool late = NX_LATE_RULE == 0 ? wave >= NW / 2 : (NX_LATE_RULE == 1 ? (wave & 1) != 0 : ((wave >> 1) & 1) != 0);
  if (late && NX_PP_PRIO) __builtin_amdgcn_s_setprio(NX_PP_PRIO);
  int Pe = P, jte = jt;                                    // block timeline: item k at h = 2k
  bool p0 = false, p1 = false, p2 = false;                 // column-sum flags of items k, k - 1, k - 2
  int slot_e = 0;
  int kw = 0;                                              // this wave's own item (P, jt; the kw-th of the block), its buffer and colbuf slot
  int buf = 0, slot = 0;
  stage(jt, 0);
  load_anchor(P);
  for (int h = 0; h <= 2 * n_it; ++h) {
    // even h: the full barrier (the compiler drains this wave's LDS-DMAs in front of it: tile k is complete); odd h: the
    // waves only change phase -- a raw s_barrier, so that tile k + 1's DMAs stay in flight across it (NX_X2_ODD: 0 none,
    // 1 raw, 2 full)
    if ((h & 1) == 0 || NX_X2_ODD == 2) __syncthreads();
    else if (NX_X2_ODD == 1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if ((h & 1) == 0) {
      const int k = h >> 1;
      p2 = p1; p1 = p0;
      p0 = k < n_it && !a.rect && jte >= (Pe + 1) * Q;
      if (p2 && tid < NX_BJ) colreduce(slot_e == 0 ? 1 : (slot_e == 1 ? 2 : 0), x + k - 2);   // slot of k - 2 = (k + 1) mod 3
      if (k + 1 < n_it) {
        item_after(Pe, jte);
#if !defined(NX_EXP_NODMA)
        stage(jte, (int)((k + 1) & 1));
#endif
      }
      slot_e = slot_e == 2 ? 0 : slot_e + 1;
    }
    const int hh = h - (late ? 1 : 0);
    if (hh < 0 || hh >= 2 * n_it) continue;
    const int j0 = jt * NX_BJ;
    if ((hh & 1) == 0) {
      if (NX_PRIO_GRAM >= 0) __builtin_amdgcn_s_setprio(NX_PRIO_GRAM);
#if !defined(NX_EXP_NOGRAM)
      gram(buf);
#endif
    } else {
      if (NX_PRIO_VP >= 0) __builtin_amdgcn_s_setprio(NX_PRIO_VP);
#if defined(NX_EXP_NOVP)
      rowsum[0] += s[0][0][0] + s[1][1][3];
#elif defined(NX_EXP_NOCOL)
      vector_phase(P, j0, mask_of(P, j0), false, slot);
#else
      vector_phase(P, j0, mask_of(P, j0), !a.rect && jt >= (P + 1) * Q, slot);
#endif
      int P_next = P, jt_next = jt;
      item_after(P_next, jt_next);
      const bool last = kw + 1 == n_it;
      if (P_next != P || last) {
        rowseg_write(P);
        if (!last) load_anchor(P_next);
      }
      P = P_next; jt = jt_next;
      ++kw;
      buf ^= 1;
      slot = slot == 2 ? 0 : slot + 1;
    }
  }
  __syncthreads();
  // the last item (early waves wrote it at h = 2 n - 1, late ones at 2 n); slot_e = (n + 1) mod 3, the item's = (n - 1) mod 3
  if (p1 && tid < NX_BJ) colreduce(slot_e == 0 ? 1 : (slot_e == 1 ? 2 : 0), x + n_it - 1);
}

// ------------------------------------------------------------------------------------------------ bf16 backward, software-pipelined
// The backward of ntx2_kernel at h = 256 runs one wave per SIMD (444 registers), and a tile is a chain: Gram product (32
// MFMAs) -> exp2 / (1/D_i + 1/D_j) / bf16 convert (~430 vector instructions) -> dZ product (32 MFMAs): nothing overlaps,
// the matrix pipe is busy 0.43 of the time.  Here the chain is cut across THREE tiles: iteration t issues, as one
// straight-line region of mutually independent work,
//     V(t)    the vector phase of tile t          S(t) -> W(t) (bf16 A fragments)
//     G(t+1)  the Gram product of tile t + 1      -> S(t + 1)
//     D(t-1)  the dZ product of tile t - 1        W(t - 1), tile t - 1 -> dZ
// so every MFMA has vector work of another tile to issue beside it.  Four j-tiles live in LDS (t - 1 for D, t, t + 1 for G,
// t + 2 arriving), written by global_load_lds_dwordx4 (the T10 (b) image, swizzle on the source address) together with
// their 1/D_j values; two S and two W register sets alternate (the loop body is instantiated twice).  The ends are not
// peeled: D(-1) runs with W = 0 over tile 0, G(n) recomputes tile n - 1; the last D runs after the loop.
template <int HP>
__global__ __launch_bounds__(256) void ntx2_bwd16p_kernel(NxArgs a) {
  constexpr int NW = 4, HPAD = 32 * HP, PR = 32 * NW, NB = 4;
  constexpr int ROW_BYTES = HPAD * 2, TILE_BYTES = NX_BJ * ROW_BYTES, KF = HPAD / 16;
  constexpr int NG = TILE_BYTES / 1024 / NW;                     // LDS-DMAs per wave and tile
  static_assert(HP % 4 == 0, "bf16 image is made of 128-column slabs");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* rdj_ring = reinterpret_cast<float*>(smem + NB * TILE_BYTES);   // [NB][64]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int M = a.M;
  const unsigned char* zn = static_cast<const unsigned char*>(a.zn);

  const int split = blockIdx.x % a.splits, P = blockIdx.x / a.splits;
  const int jt0 = split * a.jt_per_split, jt1 = min(jt0 + a.jt_per_split, a.n_jtiles), n = jt1 - jt0;
  if (n <= 0) return;

  // ---- tile jt_ (clamped to this block's last one) -> LDS buffer b_, its 1/D_j values -> rdj_ring[b_]
  const int st_row = lane >> 4;
  const unsigned st_lane = (unsigned)(st_row * ROW_BYTES + 16 * ((lane & 15) ^ (st_row << 2)));
  // (branch-free: the row index is clamped per lane, so that the DMAs can sit INSIDE the pipelined region, between its
  // MFMAs -- issued in front of it they cost the wave 100-185 cycles each, MI355X_MICROARCH.md "LDS-DMA piece issue cost")
  auto stage_tile = [&](int jt_, int b_) {
    const int j0 = min(jt_, jt1 - 1) * NX_BJ;
    unsigned off = st_lane;
    asm("" : "+v"(off));
    const int ws = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int q = ws * NG + g, slab = q >> 4, row = 4 * (q & 15) + st_row;
      const int jr = min(j0 + row, M - 1);
      const unsigned char* src = zn + (size_t)jr * ROW_BYTES + slab * 256 + ((off ^ (16u * (q & 3))) & 255u);
      unsigned char* dst = smem + b_ * TILE_BYTES + slab * (NX_BJ * 256) + (q & 15) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  auto stage_rd = [&](int jt_, int b_) {                       // 64 floats: one 4-byte DMA of wave 0 (rows past M: masked later)
    const int j0 = min(jt_, jt1 - 1) * NX_BJ;
    if (__builtin_amdgcn_readfirstlane(wave) == 0) {
      const float* src = a.rden + min(j0 + lane, M - 1);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(rdj_ring + b_ * NX_BJ), 4, 0, 0);
    }
  };
  auto stage = [&](int jt_, int b_) {
    stage_tile(jt_, b_);
    stage_rd(jt_, b_);
  };

  // ---- anchor fragments of this wave's 32 rows
  const int a_idx = P * PR + wave * 32 + l31;
  const int i_glob = a_idx < a.n_anchor ? (a.anchors ? a.anchors[a_idx] : a_idx) : -1;
  int mn = i_glob < 0 ? 0x7fffffff : i_glob, mx = i_glob;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    mn = min(mn, __shfl_xor(mn, o, 64));
    mx = max(mx, __shfl_xor(mx, o, 64));
  }
  const int i_lo = __builtin_amdgcn_readfirstlane(mn), i_hi = __builtin_amdgcn_readfirstlane(mx);
  uint4 zi[KF];
#pragma unroll
  for (int k = 0; k < KF; ++k) {
    zi[k] = make_uint4(0u, 0u, 0u, 0u);
    if (i_glob >= 0) zi[k] = *reinterpret_cast<const uint4*>(zn + (size_t)i_glob * ROW_BYTES + (2 * k + lh) * 16);
  }
  const float rd_i = (i_glob >= 0) ? a.rden[i_glob] : 0.f;

  const int tag31 = ((l31 & 3) << 2) | ((l31 >> 2) & 3);
  const int row_lc = 256 * l31 + 16 * (tag31 ^ lh);
  const int grp = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3, g1 = grp & 1, g2 = grp >> 1;
  const int trL = 2 * g1 + (p4 >> 1), trQ = 64 * q4;
  const int trC0 = 256 * (4 * g2 + q4) + 16 * (trL ^ (g2 & 3)) + 8 * (p4 & 1);
  const int trC1 = 256 * (4 * g2 + q4) + 16 * (trL ^ ((g2 + 2) & 3)) + 8 * (p4 & 1) + 256 * 8;

  nx_acc_t dz[HP];
#pragma unroll
  for (int t = 0; t < HP; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dz[t][r] = 0.f;

  // S^T of the tile in buffer b_ (row reads of the T10 image)
  auto gram = [&](int b_, nx_acc_t (&s)[2]) {
    constexpr int RD = 3, RING = RD + 1;
    uint4 ring[RING][2];
    int lc = row_lc + b_ * TILE_BYTES;
    asm("" : "+v"(lc));                                    // (not volatile: a volatile statement would fence the scheduler's region)
    auto read_k = [&](int k, uint4 (&dst)[2]) {
      const int at = (lc ^ (32 * (k & 7))) + 16384 * (k >> 3);
#pragma unroll
      for (int t = 0; t < 2; ++t) dst[t] = *reinterpret_cast<const uint4*>(smem + at + t * 8192);
    };
#pragma unroll
    for (int k = 0; k < RD && k < KF; ++k) read_k(k, ring[k % RING]);
    const nx_acc_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KF; ++k) {
      if (k + RD < KF) read_k(k + RD, ring[(k + RD) % RING]);
#pragma unroll
      for (int t = 0; t < 2; ++t)
        s[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(nx_bf16x8_t, ring[k % RING][t]),
                                                       __builtin_bit_cast(nx_bf16x8_t, zi[k]), k == 0 ? zero16 : s[t], 0, 0, 0);
    }
  };
  // W^T = exp2(S) (1/D_i + 1/D_j) as the four bf16 A fragments of the dZ product
  auto weights = [&](const nx_acc_t (&s)[2], int slot_, nx_bf16x8_t (&wf)[4]) {
    const float* rdj = rdj_ring + slot_ * NX_BJ + 4 * lh;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[t][r]) * (rd_i + rdj[32 * t + nx_acc_row(r, 0)]);
        wf[2 * t + (r >> 3)][r & 7] = (__bf16)e;
      }
  };
  // a tile that touches the diagonal or the boundary: the excluded pairs get S = -inf (exp2 -> 0) in a pass of their own,
  // so that the pipelined region below exists once, without a per-element predicate
  auto exclude = [&](nx_acc_t (&s)[2], int j0) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int jg = j0 + 32 * t + nx_acc_row(r, lh);
        if (!(jg < M && i_glob >= 0 && jg != i_glob)) s[t][r] = -__builtin_inff();
      }
  };
  // dZ += W Z_J over the tile in buffer b_ (transposed reads)
  auto dzacc = [&](int b_, const nx_bf16x8_t (&wf)[4]) {
    constexpr int N2 = 4 * HP, RD2 = 3, RING2 = RD2 + 1;
    nx_s8_t bring[RING2];
    int c0 = trC0 + b_ * TILE_BYTES, c1 = trC1 + b_ * TILE_BYTES;
    asm("" : "+v"(c0), "+v"(c1));
    auto read_b = [&](int i) {
      const int ts = i / HP, ct = i % HP;
      const int col = ((64 * (ct & 3)) ^ trQ) + 16384 * (ct >> 2) + 256 * (32 * (ts >> 1) + 16 * (ts & 1));
      const nx_s4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) nx_s4_t*)(smem + c0 + col));
      const nx_s4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) nx_s4_t*)(smem + c1 + col));
      return __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
#pragma unroll
    for (int i = 0; i < RD2; ++i) bring[i % RING2] = read_b(i);
#pragma unroll
    for (int i = 0; i < N2; ++i) {
      if (i + RD2 < N2) bring[(i + RD2) % RING2] = read_b(i + RD2);
      dz[i % HP] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i / HP], __builtin_bit_cast(nx_bf16x8_t, bring[i % RING2]),
                                                           dz[i % HP], 0, 0, 0);
    }
  };
  auto mask_of = [&](int jt_) {
    const int j0 = jt_ * NX_BJ;
    return (j0 + NX_BJ > M) || (P * PR + wave * 32 + 32 > a.n_anchor) || (j0 <= i_hi && j0 + NX_BJ > i_lo);
  };

  nx_acc_t sA[2], sB[2];
  nx_bf16x8_t wA[4], wB[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < 8; ++e) wA[q][e] = wB[q][e] = (__bf16)0.f;

  // one iteration: tile t = jt0 + k; S(t) in sc, W(t - 1) in wp  ->  S(t + 1) in sn, W(t) in wc
  auto step = [&](int k, nx_acc_t (&sc)[2], nx_acc_t (&sn)[2], nx_bf16x8_t (&wp)[4], nx_bf16x8_t (&wc)[4]) {
    __syncthreads();                                       // tile t + 1 has landed; everybody is done with tile t - 2's buffer
    stage_rd(jt0 + k + 2, (k + 2) & (NB - 1));
    if (!NX_PIPE_DMA_INSIDE) stage_tile(jt0 + k + 2, (k + 2) & (NB - 1));
    const int jt = jt0 + k, bD = k == 0 ? 0 : (k - 1) & (NB - 1);
    if (mask_of(jt)) exclude(sc, jt * NX_BJ);
    __builtin_amdgcn_sched_barrier(0);
    if (NX_PIPE_DMA_INSIDE) stage_tile(jt0 + k + 2, (k + 2) & (NB - 1));
    weights(sc, k & (NB - 1), wc);
    gram((k + 1) & (NB - 1), sn);
    dzacc(bD, wp);
    // the issue order of the region: every MFMA followed by its share of the LDS reads and of the vector work (left to
    // itself hipcc puts one step's 32 exp2 in front of the MFMAs and leaves the other step's MFMAs bare)
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  // 1 MFMA
      if (NX_PIPE_DMA_INSIDE && (i & 7) == 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // one of the 8 LDS-DMAs
      __builtin_amdgcn_sched_group_barrier(0x100, NX_PIPE_DS, 0);         // LDS reads
      if (NX_PIPE_TRANS) __builtin_amdgcn_sched_group_barrier(0x400, NX_PIPE_TRANS, 0);   // exp2 (not part of the VALU class)
      __builtin_amdgcn_sched_group_barrier(0x002, NX_PIPE_VALU, 0);       // vector instructions
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  stage(jt0, 0);
  stage(jt0 + 1, 1);
  __syncthreads();
  gram(0, sA);
  int k = 0;
  for (; k + 1 < n; k += 2) {
    step(k, sA, sB, wA, wB);
    step(k + 1, sB, sA, wB, wA);
  }
  if (k < n) {                                             // odd count: one more step, then the last dZ from wB
    step(k, sA, sB, wA, wB);
    __syncthreads();
    dzacc((n - 1) & (NB - 1), wB);
  } else {
    __syncthreads();
    dzacc((n - 1) & (NB - 1), wA);
  }

  const int ib = P * PR + wave * 32;
  float* out = a.dzpart + (size_t)split * a.n_anchor * HPAD;
#pragma unroll
  for (int ct = 0; ct < HP; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ia = ib + nx_acc_row(r, lh);
      if (ia < a.n_anchor) out[(size_t)ia * HPAD + 32 * ct + l31] = dz[ct][r];
    }
}

// ------------------------------------------------------------------------------------------------ small kernels
// One wave per positive pair (i, i + N): F.normalize (x / max(|x|, 1e-12)), padded fp32 (+ bf16) images, positive dot.
__global__ __launch_bounds__(256) void nx_prep_kernel(const float* __restrict__ h1, int64_t ld1,
                                                      const float* __restrict__ h2, int64_t ld2, int N, int h, int hpad,
                                                      float* __restrict__ zn, __bf16* __restrict__ zn16,
                                                      float zn16_scale, float* __restrict__ norm,
                                                      float* __restrict__ posdot) {
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const int c = lane * 4;
  float u[4], v[4];
  float su = 0.f, sv = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    u[q] = (c + q < h) ? h1[(int64_t)i * ld1 + c + q] : 0.f;
    v[q] = (c + q < h) ? h2[(int64_t)i * ld2 + c + q] : 0.f;
    su += u[q] * u[q];
    sv += v[q] * v[q];
  }
  const float nu = fmaxf(sqrtf(wave_sum(su)), 1e-12f), nv = fmaxf(sqrtf(wave_sum(sv)), 1e-12f);
  float dot = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    u[q] /= nu;
    v[q] /= nv;
    dot += u[q] * v[q];
  }
  dot = wave_sum(dot);
  if (c < hpad) {
    *reinterpret_cast<float4*>(zn + (size_t)i * hpad + c) = make_float4(u[0], u[1], u[2], u[3]);
    *reinterpret_cast<float4*>(zn + (size_t)(i + N) * hpad + c) = make_float4(v[0], v[1], v[2], v[3]);
    if (zn16) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        zn16[(size_t)i * hpad + c + q] = (__bf16)(u[q] * zn16_scale);
        zn16[(size_t)(i + N) * hpad + c + q] = (__bf16)(v[q] * zn16_scale);
      }
    }
  }
  if (lane == 0) {
    norm[i] = nu;
    norm[i + N] = nv;
    posdot[i] = dot;
  }
}

// D_i = (row segments of i's panel, in block order) + (column items of i's j-tile, in panel order); 1/D; loss term.
// rect (anchor subset): row segments only.  With `loss`: the terms are added in a fixed order inside this launch -- a
// tree per block (double), then the block that takes the last ticket adds the blocks' sums in block order and writes
// loss = sum / n_anchor (the ticket was zeroed by the Gram launch in front; slots: one double per block).
__global__ __launch_bounds__(256) void nx_fwd_finish_kernel(const float* __restrict__ rowseg, const float* __restrict__ colitem,
                                                            const float* __restrict__ posdot, int n_anchor, int N, int PR,
                                                            int n_panels, int n_jtiles, long long n_items, long long grid,
                                                            int rect, float inv_tau, float* __restrict__ den,
                                                            float* __restrict__ rden, double* __restrict__ term,
                                                            double* __restrict__ slots, unsigned int* __restrict__ ticket,
                                                            float* __restrict__ loss) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  double t = 0.0;
  if (i < n_anchor) {
    const int q = PR / NX_BJ, P = i / PR, jt = i / NX_BJ;
    const long long s0 = rect ? (long long)P * n_jtiles : nx_start(P, n_jtiles, q);
    const long long s1 = rect ? (long long)(P + 1) * n_jtiles : nx_start(P + 1, n_jtiles, q);
    const long long b0 = nx_block_of(s0, n_items, grid), b1 = nx_block_of(s1 - 1, n_items, grid);
    // the same left-to-right sums as ever, the loads four ahead of the adds (a thread's chain is 20-odd dependent loads
    // at the small sizes where this launch shows)
    float d = 0.f;
    const float* rs = rowseg + (size_t)(b0 + P) * PR + i % PR;
    const int nb = (int)(b1 - b0) + 1;
    int b = 0;
    for (; b + 4 <= nb; b += 4) {
      const float v0 = rs[(size_t)b * PR], v1 = rs[(size_t)(b + 1) * PR], v2 = rs[(size_t)(b + 2) * PR], v3 = rs[(size_t)(b + 3) * PR];
      d += v0; d += v1; d += v2; d += v3;
    }
    for (; b < nb; ++b) d += rs[(size_t)b * PR];
    if (!rect) {
      const int p_end = min(n_panels, jt / q);          // panels strictly left of this j-tile's own panel columns
      const float* ci = colitem + i % NX_BJ;
      int Pc = 0;
      for (; Pc + 4 <= p_end; Pc += 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ci[(size_t)(nx_start(Pc + u, n_jtiles, q) + jt - (Pc + u) * q) * NX_BJ];
        d += v[0]; d += v[1]; d += v[2]; d += v[3];
      }
      for (; Pc < p_end; ++Pc) d += ci[(size_t)(nx_start(Pc, n_jtiles, q) + jt - Pc * q) * NX_BJ];
    }
    den[i] = d;
    if (rden) rden[i] = 1.0f / d;
    if (term || loss) t = log((double)d) - (i < N ? 2.0 * (double)posdot[i] * (double)inv_tau : 0.0);
    if (term) term[i] = t;
  }
  if (!loss) return;
  __shared__ double red[256];
  __shared__ bool last;
  red[threadIdx.x] = t;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    slots[blockIdx.x] = red[0];
    __threadfence();
    last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  double acc = 0.0;                                    // thread k: blocks k, k + 256, ... in order; then the same tree
  for (unsigned bb = threadIdx.x; bb < gridDim.x; bb += 256) acc += __builtin_nontemporal_load(slots + bb);
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss[0] = (float)(red[0] / (double)n_anchor);
    *ticket = 0u;
  }
}

// dh = ((sum_s part_s - 2 z_p(i)) * g * inv_m_tau  projected through x / |x|) for one anchor row per wave.
// anchors (nullable): anchor a is row anchors[a] of zn; anchor a < n_half goes to dh1[a], the rest to dh2[a - n_half].
__global__ __launch_bounds__(256) void nx_bwd_finish_kernel(const float* __restrict__ dzpart, int splits,
                                                            const float* __restrict__ zn, const float* __restrict__ norm,
                                                            const float* __restrict__ gout, const int* __restrict__ anchors,
                                                            int n_anchor, int n_half, int N, int h, int hpad,
                                                            float inv_m_tau, float part_scale, float* __restrict__ dh1,
                                                            int64_t ld1,
                                                            float* __restrict__ dh2, int64_t ld2) {
  const int lane = threadIdx.x & 63, a = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (a >= n_anchor) return;
  const int i = anchors ? anchors[a] : a;
  const int c = lane * 4;
  const float g = gout[0] * inv_m_tau;
  float4 d = make_float4(0.f, 0.f, 0.f, 0.f), z = d;
  if (c < hpad) {
    for (int s = 0; s < splits; ++s) {
      const float4 p = *reinterpret_cast<const float4*>(dzpart + ((size_t)s * n_anchor + a) * hpad + c);
      d.x += p.x; d.y += p.y; d.z += p.z; d.w += p.w;
    }
    d.x *= part_scale; d.y *= part_scale; d.z *= part_scale; d.w *= part_scale;   // bf16 image rows carry sqrt(log2e/tau)
    z = *reinterpret_cast<const float4*>(zn + (size_t)i * hpad + c);
    const int pi = i < N ? i + N : i - N;
    const float4 zp = *reinterpret_cast<const float4*>(zn + (size_t)pi * hpad + c);
    d.x = (d.x - 2.f * zp.x) * g; d.y = (d.y - 2.f * zp.y) * g; d.z = (d.z - 2.f * zp.z) * g; d.w = (d.w - 2.f * zp.w) * g;
  }
  const float zd = wave_sum(z.x * d.x + z.y * d.y + z.z * d.z + z.w * d.w);
  const float inv = 1.0f / norm[a];
  float* out = a < n_half ? dh1 + (int64_t)a * ld1 : dh2 + (int64_t)(a - n_half) * ld2;
  const float o[4] = {(d.x - z.x * zd) * inv, (d.y - z.y * zd) * inv, (d.z - z.z * zd) * inv, (d.w - z.w * zd) * inv};
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (c + q < h) out[c + q] = o[q];
}

// ------------------------------------------------------------------------------------------------ host side
struct NxPlan { int nw, smem, occ; };

template <bool BF16, int HP, int NW, bool BWD>
static NxPlan nx_plan() {
  constexpr int HPAD = 32 * HP;
  const int tile = BF16 ? (BWD ? NX_BJ * HPAD * 2 : NX_BJ * (HPAD * 2 + 16)) : NX_BJ * (HPAD + 4) * 4;
  const int smem = 2 * tile + (BWD ? 2 * NX_BJ : (((NX_PINGPONG == 2 || (NX_PINGPONG == 1 && BF16)) && NW == 8) ? 3 : 2) * NW * NX_BJ) * 4;
  static int occ_cache = -1;
  if (occ_cache < 0) {
    hipFuncSetAttribute((const void*)ntx2_kernel<BF16, HP, NW, BWD>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)ntx2_kernel<BF16, HP, NW, BWD>, NW * 64, smem) !=
            hipSuccess || occ < 1)
      occ = 1;
    occ_cache = occ;
  }
  return NxPlan{NW, smem, occ_cache};
}

static int nx_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipGetDevice(&dev);
    hipDeviceProp_t p;
    cus = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  return cus;
}

// waves per workgroup by (precision, direction, width): 8 waves (a 256-row panel, two waves per SIMD) wherever the
// kernel fits 256 registers per lane, 4 waves (one per SIMD, up to 512 registers) for the widest accumulators
static int nx_waves(bool bf16, int hp, bool bwd) { return ((bwd || (bf16 && NX_FWD16_NW == 4)) && hp > 4) ? 4 : 8; }

// (precision, direction, width) -> instantiation; the workgroup shape follows nx_waves()
#define NX_FWD_CASES(FN, ...)                                                                     \
  if (bf16) return hp <= 4 ? FN<true, 4, 8, false>(__VA_ARGS__) : FN<true, 8, NX_FWD16_NW, false>(__VA_ARGS__); \
  switch (hp) {                                                                                   \
    case 1: return FN<false, 1, 8, false>(__VA_ARGS__);                                           \
    case 2: return FN<false, 2, 8, false>(__VA_ARGS__);                                           \
    case 3: return FN<false, 3, 8, false>(__VA_ARGS__);                                           \
    case 4: return FN<false, 4, 8, false>(__VA_ARGS__);                                           \
    case 5: return FN<false, 5, 8, false>(__VA_ARGS__);                                           \
    case 6: return FN<false, 6, 8, false>(__VA_ARGS__);                                           \
    case 7: return FN<false, 7, 8, false>(__VA_ARGS__);                                           \
    default: return FN<false, 8, 8, false>(__VA_ARGS__);                                          \
  }
#define NX_BWD_CASES(FN, ...)                                                                     \
  if (bf16) return hp <= 4 ? FN<true, 4, 8, true>(__VA_ARGS__) : FN<true, 8, 4, true>(__VA_ARGS__);   \
  switch (hp) {                                                                                   \
    case 1: return FN<false, 1, 8, true>(__VA_ARGS__);                                            \
    case 2: return FN<false, 2, 8, true>(__VA_ARGS__);                                            \
    case 3: return FN<false, 3, 8, true>(__VA_ARGS__);                                            \
    case 4: return FN<false, 4, 8, true>(__VA_ARGS__);                                            \
    case 5: return FN<false, 5, 4, true>(__VA_ARGS__);                                            \
    case 6: return FN<false, 6, 4, true>(__VA_ARGS__);                                            \
    case 7: return FN<false, 7, 4, true>(__VA_ARGS__);                                            \
    default: return FN<false, 8, 4, true>(__VA_ARGS__);                                           \
  }

template <bool BF16, int HP, int NW, bool BWD>
static int nx_occ_of() { return nx_plan<BF16, HP, NW, BWD>().occ; }
template <bool BF16, int HP, int NW, bool BWD>
static int nx_launch(NxArgs a, int grid, hipStream_t s) {
  const NxPlan p = nx_plan<BF16, HP, NW, BWD>();
  hipLaunchKernelGGL((ntx2_kernel<BF16, HP, NW, BWD>), dim3(grid), dim3(NW * 64), p.smem, s, a);
  return check_launch(BWD ? "bkg_ntxent_backward" : "bkg_ntxent_forward");
}

template <int HP>
static NxPlan nx_plan_x2() {
  const int smem = 2 * NX_BJ * 64 * HP + 3 * 8 * NX_BJ * 4;
  static int occ_cache = -1;
  if (occ_cache < 0) {
    const void* fn = NX_FWD16_Q ? (const void*)ntx2_fwd16q_kernel<HP> : (const void*)ntx2_fwd16x2_kernel<HP>;
    hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 512, smem) != hipSuccess || occ < 1) occ = 1;
    occ_cache = occ;
  }
  return NxPlan{8, smem, occ_cache};
}
template <int HP>
static int nx_launch_x2(NxArgs a, int grid, hipStream_t s) {
  const NxPlan p = nx_plan_x2<HP>();
  if (NX_FWD16_Q) hipLaunchKernelGGL((ntx2_fwd16q_kernel<HP>), dim3(grid), dim3(512), p.smem, s, a);
  else hipLaunchKernelGGL((ntx2_fwd16x2_kernel<HP>), dim3(grid), dim3(512), p.smem, s, a);
  return check_launch("bkg_ntxent_forward");
}

static int nx_occ(bool bf16, int hp, bool bwd) {
  if (bwd) { NX_BWD_CASES(nx_occ_of) }
  if (bf16 && NX_FWD16_X2) return hp <= 4 ? nx_plan_x2<4>().occ : nx_plan_x2<8>().occ;
  NX_FWD_CASES(nx_occ_of)
}
static int nx_run_fwd(bool bf16, int hp, NxArgs a, int grid, hipStream_t s) {
  if (bf16 && NX_FWD16_X2) return hp <= 4 ? nx_launch_x2<4>(a, grid, s) : nx_launch_x2<8>(a, grid, s);
  NX_FWD_CASES(nx_launch, a, grid, s)
}
static NxPlan nx_plan_bwdp() {
  const int smem = 4 * NX_BJ * 512 + 4 * NX_BJ * 4;
  static bool done = false;
  if (!done) {
    hipFuncSetAttribute((const void*)ntx2_bwd16p_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    done = true;
  }
  return NxPlan{4, smem, 1};
}
static int nx_run_bwd(bool bf16, int hp, NxArgs a, int grid, hipStream_t s) {
  if (bf16 && hp > 4 && NX_BWD16_PIPE) {
    const NxPlan p = nx_plan_bwdp();
    hipLaunchKernelGGL((ntx2_bwd16p_kernel<8>), dim3(grid), dim3(256), p.smem, s, a);
    return check_launch("bkg_ntxent_backward");
  }
  NX_BWD_CASES(nx_launch, a, grid, s)
}

struct NxShape { int hpad, hp, pr, n_panels, n_jtiles, grid, splits, jt_per_split; long long n_items; };

// n_anchor < 0: the whole (symmetric) problem; else a rectangular one over n_anchor anchor rows
static NxShape nx_shape(int64_t M, int64_t n_anchor, int h, bool bf16, bool bwd) {
  NxShape sh;
  const bool rect = n_anchor >= 0;
  if (!rect) n_anchor = M;
  sh.hpad = bf16 ? (h <= 128 ? 128 : 256) : ((h + 31) / 32) * 32;
  sh.hp = sh.hpad / 32;
  sh.pr = (bf16 && !bwd && NX_FWD16_X2) ? 512 : 32 * nx_waves(bf16, sh.hp, bwd);      // (x2 forward: 64 anchors per wave)
  sh.n_panels = (int)((n_anchor + sh.pr - 1) / sh.pr);
  sh.n_jtiles = (int)((M + NX_BJ - 1) / NX_BJ);
  sh.n_items = rect ? (long long)sh.n_panels * sh.n_jtiles : nx_start(sh.n_panels, sh.n_jtiles, sh.pr / NX_BJ);
  const int slots = nx_cus() * nx_occ(bf16, sh.hp, bwd);
  sh.grid = (int)std::min<long long>(sh.n_items, slots);
  // backward: enough j-splits to fill the chip once; none when the panels alone do (every workgroup then streams
  // the same j-tiles at the same time: the tile crosses the fabric once per XCD)
  sh.splits = sh.n_panels >= slots ? 1 : std::max(1, std::min(sh.n_jtiles, (slots + sh.n_panels / 2) / sh.n_panels));
  sh.jt_per_split = (sh.n_jtiles + sh.splits - 1) / sh.splits;
  sh.splits = (sh.n_jtiles + sh.jt_per_split - 1) / sh.jt_per_split;
  return sh;
}

}  // namespace bkg

using namespace bkg;

// ---- C ABI (declared in include/biomedkg_hip.h)
extern "C" int32_t bkg_ntxent_hpad(int32_t h, int32_t bf16) {
  if (h <= 0 || h > 256) return -1;
  return bf16 ? (h <= 128 ? 128 : 256) : ((h + 31) / 32) * 32;
}

extern "C" float bkg_ntxent_bf16_scale(float tau) { return sqrtf(1.4426950408889634f / tau); }

extern "C" int bkg_ntxent_prepare(const float* h1, int64_t ld1, const float* h2, int64_t ld2, int64_t N, int32_t h,
                                  int32_t hpad, float tau, float* zn, void* zn_bf16, float* norm, float* posdot,
                                  void* stream) {
  BKG_ARG(N > 0 && h > 0 && h <= 256 && hpad >= h && hpad % 32 == 0 && hpad <= 256, "bkg_ntxent_prepare: 0 < h <= hpad <= 256");
  BKG_ARG(tau > 0.f, "bkg_ntxent_prepare: tau > 0");
  hipLaunchKernelGGL(nx_prep_kernel, dim3(ceil_div(N, 4)), dim3(256), 0, (hipStream_t)stream, h1, ld1, h2, ld2, (int)N, h,
                     hpad, zn, (__bf16*)zn_bf16, bkg_ntxent_bf16_scale(tau), norm, posdot);
  return check_launch("bkg_ntxent_prepare");
}

static int64_t nx_loss_tail_floats(int64_t M) { return 2 * (ceil_div(M, 256) + 1) + 2; }
static bool nx_args_ok(int64_t M, int64_t n_anchor, int32_t h, float tau) {
  return M > 0 && M % 2 == 0 && M < (1ll << 30) && h > 0 && h <= 256 && tau > 0.f && n_anchor <= M;
}

extern "C" int64_t bkg_ntxent_rows_forward_workspace_floats(int64_t M, int64_t n_anchor, int32_t h, int32_t bf16) {
  if (M <= 0 || h <= 0 || h > 256 || n_anchor == 0) return -1;
  const NxShape sh = nx_shape(M, n_anchor, h, bf16 != 0, false);
  // + the loss reduce of the whole-problem forward: one double per finish block and the ticket (8-byte aligned tail)
  return (int64_t)(sh.grid + sh.n_panels) * sh.pr + (n_anchor < 0 ? sh.n_items * NX_BJ + nx_loss_tail_floats(M) : 0);
}
extern "C" int64_t bkg_ntxent_forward_workspace_floats(int64_t M, int32_t h, int32_t bf16) {
  return bkg_ntxent_rows_forward_workspace_floats(M, -1, h, bf16);
}

static int nx_forward(const void* zn, int32_t bf16, int64_t M, int32_t h, float tau, const int32_t* anchors,
                      int64_t n_anchor, const float* posdot, float* workspace, float* den, float* rden, double* term,
                      float* loss, hipStream_t s) {
  const NxShape sh = nx_shape(M, n_anchor, h, bf16 != 0, false);
  const bool rect = n_anchor >= 0;
  NxArgs a{};
  a.zn = zn; a.M = (int)M; a.scale = 1.4426950408889634f / tau;
  a.n_panels = sh.n_panels; a.n_jtiles = sh.n_jtiles; a.n_items = sh.n_items;
  a.anchors = anchors; a.n_anchor = (int)(rect ? n_anchor : M); a.rect = rect;
  a.rowseg = workspace;
  a.colitem = workspace + (size_t)(sh.grid + sh.n_panels) * sh.pr;
  double* slots = nullptr;
  if (loss) {                                         // the tail of the workspace, on an 8-byte boundary
    uintptr_t tail = reinterpret_cast<uintptr_t>(a.colitem + (size_t)sh.n_items * NX_BJ);
    slots = reinterpret_cast<double*>((tail + 7) & ~(uintptr_t)7);
    a.ticket = reinterpret_cast<unsigned int*>(slots + ceil_div(M, 256));
  }
  int rc = nx_run_fwd(bf16 != 0, sh.hp, a, sh.grid, s);
  if (rc) return rc;
  hipLaunchKernelGGL(nx_fwd_finish_kernel, dim3(ceil_div(a.n_anchor, 256)), dim3(256), 0, s, a.rowseg, a.colitem, posdot,
                     a.n_anchor, (int)(M / 2), sh.pr, sh.n_panels, sh.n_jtiles, sh.n_items, (long long)sh.grid, (int)rect,
                     1.0f / tau, den, rden, term, slots, a.ticket, loss);
  return check_launch("bkg_ntxent_forward(finish)");
}

// den[M], rden[M] = 1/D, term[M] (double, nullable) = log D_i - (i < N) * 2 z_i.z_(i+N) / tau, loss[1] (nullable) =
// sum(term) / M added in a fixed order inside the finish launch
extern "C" int bkg_ntxent_forward(const void* zn, int32_t bf16, int64_t M, int32_t h, float tau, const float* posdot,
                                  float* workspace, float* den, float* rden, double* term, float* loss, void* stream) {
  BKG_ARG(nx_args_ok(M, M, h, tau), "bkg_ntxent_forward: bad arguments (h <= 256)");
  return nx_forward(zn, bf16, M, h, tau, nullptr, -1, posdot, workspace, den, rden, term, loss, (hipStream_t)stream);
}

// den[n_anchor] = D of the anchor rows zn[anchors[a]] against ALL M rows (row-partitioned multi-GPU forward)
extern "C" int bkg_ntxent_rows_forward(const void* zn, int32_t bf16, int64_t M, int32_t h, float tau,
                                       const int32_t* anchors, int64_t n_anchor, float* workspace, float* den,
                                       void* stream) {
  BKG_ARG(nx_args_ok(M, n_anchor, h, tau) && n_anchor > 0 && anchors, "bkg_ntxent_rows_forward: bad arguments (h <= 256)");
  return nx_forward(zn, bf16, M, h, tau, anchors, n_anchor, nullptr, workspace, den, nullptr, nullptr, nullptr,
                    (hipStream_t)stream);
}

extern "C" int64_t bkg_ntxent_rows_backward_workspace_floats(int64_t M, int64_t n_anchor, int32_t h, int32_t bf16) {
  if (M <= 0 || h <= 0 || h > 256 || n_anchor == 0) return -1;
  const NxShape sh = nx_shape(M, n_anchor, h, bf16 != 0, true);
  return (int64_t)sh.splits * (n_anchor < 0 ? M : n_anchor) * sh.hpad;
}
extern "C" int64_t bkg_ntxent_backward_workspace_floats(int64_t M, int32_t h, int32_t bf16) {
  return bkg_ntxent_rows_backward_workspace_floats(M, -1, h, bf16);
}

static int nx_backward(const void* zn, int32_t bf16, const float* zn_f32, const float* norm, const float* rden,
                       const float* gout, int64_t M, int32_t h, float tau, const int32_t* anchors, int64_t n_anchor,
                       int64_t n_half, float inv_m_tau, float* workspace, float* dh1, int64_t ld1, float* dh2, int64_t ld2,
                       hipStream_t s) {
  const NxShape sh = nx_shape(M, n_anchor, h, bf16 != 0, true);
  NxArgs a{};
  a.zn = zn; a.M = (int)M; a.scale = 1.4426950408889634f / tau;
  a.n_panels = sh.n_panels; a.n_jtiles = sh.n_jtiles;
  a.anchors = anchors; a.n_anchor = (int)(n_anchor >= 0 ? n_anchor : M);
  a.rden = rden; a.dzpart = workspace; a.splits = sh.splits; a.jt_per_split = sh.jt_per_split;
  int rc = nx_run_bwd(bf16 != 0, sh.hp, a, sh.n_panels * sh.splits, s);
  if (rc) return rc;
  const float part_scale = bf16 ? 1.0f / sqrtf(a.scale) : 1.0f;
  hipLaunchKernelGGL(nx_bwd_finish_kernel, dim3(ceil_div(a.n_anchor, 4)), dim3(256), 0, s, workspace, sh.splits, zn_f32,
                     norm, gout, anchors, a.n_anchor, (int)n_half, (int)(M / 2), h, sh.hpad, inv_m_tau, part_scale, dh1, ld1,
                     dh2, ld2);
  return check_launch("bkg_ntxent_backward(finish)");
}

// dh1 / dh2 [N, h]: gradient of the loss wrt the UN-normalised inputs, times the upstream scalar *gout (device)
extern "C" int bkg_ntxent_backward(const void* zn, int32_t bf16, const float* zn_f32, const float* norm, const float* rden,
                                   const float* gout, int64_t M, int32_t h, float tau, float* workspace, float* dh1,
                                   int64_t ld1, float* dh2, int64_t ld2, void* stream) {
  BKG_ARG(nx_args_ok(M, M, h, tau), "bkg_ntxent_backward: bad arguments (h <= 256)");
  return nx_backward(zn, bf16, zn_f32, norm, rden, gout, M, h, tau, nullptr, -1, M / 2, 1.0f / ((float)M * tau), workspace,
                     dh1, ld1, dh2, ld2, (hipStream_t)stream);
}

// row-partitioned backward: anchors as in bkg_ntxent_rows_forward, norm[n_anchor] their norms, rden[M] = 1/D of ALL rows
// (all-gathered); anchor a < n_half -> dh1[a], else dh2[a - n_half]; the gradient is scaled by *gout * grad_scale
// (grad_scale = world / (M tau) for the whole-graph loss under gradient averaging)
extern "C" int bkg_ntxent_rows_backward(const void* zn, int32_t bf16, const float* zn_f32, const float* norm,
                                        const float* rden, const float* gout, int64_t M, int32_t h, float tau,
                                        const int32_t* anchors, int64_t n_anchor, int64_t n_half, float grad_scale,
                                        float* workspace, float* dh1, int64_t ld1, float* dh2, int64_t ld2, void* stream) {
  BKG_ARG(nx_args_ok(M, n_anchor, h, tau) && n_anchor > 0 && anchors && n_half >= 0 && n_half <= n_anchor,
          "bkg_ntxent_rows_backward: bad arguments (h <= 256)");
  return nx_backward(zn, bf16, zn_f32, norm, rden, gout, M, h, tau, anchors, n_anchor, n_half, grad_scale, workspace, dh1, ld1,
                     dh2, ld2, (hipStream_t)stream);
}
