// 16-byte pieces of a gather-table row, shared by the segment-reduce (spmm.hip) and KGE (kge*.hip)
// kernels.  A table is fp32 (exact mode) or bf16 (BASELINE config 3: bf16 storage, fp32 accumulate); a
// 16-byte load carries EPL = 4 or 8 columns and expands to Q = 1 or 2 float4 of accumulator width.
// Lane `lig` of an LPR-lane group owns, for piece v, columns EPL * (lig + v * LPR) + [0, EPL).
#pragma once
#include "common.h"

namespace bkg {

__device__ __forceinline__ void fma4(float4& a, float w, const float4& v) {
  a.x = fmaf(w, v.x, a.x); a.y = fmaf(w, v.y, a.y); a.z = fmaf(w, v.z, a.z); a.w = fmaf(w, v.w, a.w);
}

template <typename T> struct Piece;

template <> struct Piece<float> {
  static constexpr int Q = 1, EPL = 4;
  float4 v;
  __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
  __device__ __forceinline__ void expand(float4* out) const { out[0] = v; }
  __device__ __forceinline__ void fma(float4* acc, float w) const { fma4(acc[0], w, v); }
};

template <> struct Piece<__bf16> {
  static constexpr int Q = 2, EPL = 8;
  uint4 v;
  __device__ __forceinline__ void load(const __bf16* p) { v = *reinterpret_cast<const uint4*>(p); }
  // bf16 -> fp32 is a 16-bit shift: element 2i is the low half of word i
  __device__ __forceinline__ void expand(float4* out) const {
    out[0] = make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                         __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
    out[1] = make_float4(__uint_as_float(v.z << 16), __uint_as_float(v.z & 0xffff0000u),
                         __uint_as_float(v.w << 16), __uint_as_float(v.w & 0xffff0000u));
  }
  __device__ __forceinline__ void fma(float4* acc, float w) const {
    float4 e[2];
    expand(e);
    fma4(acc[0], w, e[0]);
    fma4(acc[1], w, e[1]);
  }
};

// column of accumulator chunk vq (= v * Q + q) of lane `lig`
template <typename T, int LPR>
__device__ __forceinline__ int piece_col(int lig, int vq) {
  constexpr int Q = Piece<T>::Q, EPL = Piece<T>::EPL;
  return EPL * (lig + (vq / Q) * LPR) + 4 * (vq % Q);
}

template <int N>
__device__ __forceinline__ void zero4(float4 (&acc)[N]) {
#pragma unroll
  for (int v = 0; v < N; ++v) acc[v] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- entry batches.  An item's entry range is wave-uniform (its bounds come through readfirstlane), so when the wave
// has one or two lane groups the indices and weights of a batch of NB consecutive entries are read with SCALAR loads
// into SGPRs: no vector registers, no texture-addresser cycles (a wave-wide 4-byte load with two distinct addresses
// occupies the addresser like a full one), and -- on their own counter (lgkmcnt) -- the NEXT batch's indices arrive
// while this batch's rows are in flight.  Lane group g of the G = 64 / LPR groups takes entries g, g + G, ... of the
// batch (a v_cndmask between two SGPRs).  With four or more groups a batch would not fit the scalar register file:
// those widths (d <= 128 bf16 / d <= 64 fp32) keep per-lane loads.
template <int G, int UT>
struct EntryIdx {
  static constexpr bool SCALAR = G <= 2;
  int v[SCALAR ? UT * G : UT];
  __device__ __forceinline__ void load(int e, int last, const int32_t* __restrict__ col, int grp) {
    if constexpr (SCALAR) {
#pragma unroll
      for (int j = 0; j < UT * G; ++j) v[j] = col[min(e + j, last)];
    } else {
#pragma unroll
      for (int u = 0; u < UT; ++u) v[u] = col[min(e + u * G + grp, last)];
    }
  }
  __device__ __forceinline__ int mine(int u, int grp) const {
    if constexpr (!SCALAR) return v[u];
    else if constexpr (G == 1) return v[u];
    else return grp ? v[u * G + 1] : v[u * G];
  }
};
template <int G, int UT, bool HAS_W>
struct EntryWt {
  static constexpr bool SCALAR = G <= 2;
  float v[HAS_W ? (SCALAR ? UT * G : UT) : 1];
  __device__ __forceinline__ void load(int e, int last, const float* __restrict__ w, int grp) {
    if constexpr (HAS_W) {
      if constexpr (SCALAR) {
#pragma unroll
        for (int j = 0; j < UT * G; ++j) v[j] = w[min(e + j, last)];
      } else {
#pragma unroll
        for (int u = 0; u < UT; ++u) v[u] = w[min(e + u * G + grp, last)];
      }
    }
  }
  __device__ __forceinline__ float mine(int u, int grp) const {
    if constexpr (!HAS_W) return 1.0f;
    else if constexpr (!SCALAR || G == 1) return v[u];
    else return grp ? v[u * G + 1] : v[u * G];
  }
};
constexpr int NO_CLAMP = 0x7fffffff;


// d -> (LPR, VPL) instantiations shared by the launchers: PIECES = d / EPL sixteen-byte pieces per row.
// (L32, L64: lanes per row for the two widths of the d = 256 tables -- 32 bf16 pieces / 64 fp32 pieces.  The
// segment-reduce SpMM uses one piece per lane; the KGE kernels do scalar work per triplet (score reduction, sigmoid,
// index arithmetic) and run faster with fewer lanes and more pieces per lane: measured 736 -> 558 us per 4 M triplets.)
#define BKG_PIECE_DISPATCH_(PIECES, L32, L64, ...)                        \
  switch (PIECES) {                                                       \
    case 2: { constexpr int LPR = 2, VPL = 1; __VA_ARGS__; } break;       \
    case 4: { constexpr int LPR = 4, VPL = 1; __VA_ARGS__; } break;       \
    case 6: { constexpr int LPR = 2, VPL = 3; __VA_ARGS__; } break;       \
    case 8: { constexpr int LPR = 8, VPL = 1; __VA_ARGS__; } break;       \
    case 12: { constexpr int LPR = 4, VPL = 3; __VA_ARGS__; } break;      \
    case 16: { constexpr int LPR = 16, VPL = 1; __VA_ARGS__; } break;     \
    case 24: { constexpr int LPR = 8, VPL = 3; __VA_ARGS__; } break;      \
    case 32: { constexpr int LPR = L32, VPL = 32 / (L32); __VA_ARGS__; } break;   \
    case 64: { constexpr int LPR = L64, VPL = 64 / (L64); __VA_ARGS__; } break;   \
    case 96: { constexpr int LPR = 32, VPL = 3; __VA_ARGS__; } break;     \
    case 128: { constexpr int LPR = 64, VPL = 2; __VA_ARGS__; } break;    \
    case 192: { constexpr int LPR = 64, VPL = 3; __VA_ARGS__; } break;    \
    default: bkg::set_error_msg("unsupported embedding width (d / 4 for fp32, d / 8 for bf16 tables must be one of 2,4,6,8,12,16,24,32,64,96,128,192)"); return -1; \
  }
#define BKG_PIECE_DISPATCH(PIECES, ...) BKG_PIECE_DISPATCH_(PIECES, 32, 64, __VA_ARGS__)

}  // namespace bkg
