// Shared device/host helpers for the biomedkg HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/biomedkg_hip.h"

#define BKG_WAVE 64

namespace bkg {

// Last error text for bkg_last_error(); set by the launch wrappers.
void set_error(const char* what, hipError_t e);
void set_error_msg(const char* what);

static inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(what, e);
    return (int)e;
  }
  return 0;
}

#define BKG_ARG(cond, msg)          \
  do {                              \
    if (!(cond)) {                  \
      bkg::set_error_msg(msg);      \
      return -1;                    \
    }                               \
  } while (0)

// In a split-K launcher: switch `ep` to plain stores into per-chunk slabs when the deterministic workspace is set.
#define BKG_DET_SLABS(det, ep, n_chunks, M, N, who)                                              \
  do {                                                                                           \
    const bkg::DetWorkspace dw_ = bkg::det_workspace();                                          \
    if (dw_.ptr) {                                                                               \
      if ((int64_t)(n_chunks) * (int64_t)(M) * (int64_t)(N) > dw_.floats) {                      \
        bkg::set_error_msg(who ": deterministic workspace too small (bkg_set_deterministic_workspace)"); \
        return -1;                                                                               \
      }                                                                                          \
      (det) = dw_.ptr;                                                                           \
      (ep).atomic = 0;                                                                           \
    }                                                                                            \
  } while (0)

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Deterministic mode (bkg_set_deterministic_workspace): the split-K weight-gradient products write one [M, N] slab per
// K chunk into this caller-owned buffer instead of adding with float atomics; splitk_reduce() then sums the slabs of
// each output in chunk order.  {nullptr, 0} = off.
struct DetWorkspace { float* ptr; int64_t floats; };
DetWorkspace det_workspace();
// C_g[m, n] = sum over the group's chunks (in chunk order) of ws[(first_chunk(g) + c) * M * N + m * N + n];
// group_ptr == nullptr: one group of `n_chunks` chunks.  Chunks of group g: ceil(rows_g / kchunk).
void splitk_reduce(const float* ws, float* C, int64_t ldc, int64_t strideC, int M, int N, const int32_t* group_ptr, int G,
                   int kchunk, int n_chunks, hipStream_t s);

// out[c] = sum over the n_blocks partial rows (in a fixed order) of partial[b * d + c]  (colsum.hip)
void colsum_final_launch(const float* partial, int n_blocks, int d, float* out, hipStream_t s);

// XCD-aware block -> work mapping.  MI355X dispatches workgroup b of a 1-D grid to XCD b % 8, and each of the 8
// XCDs has a private L2.  Work is indexed (outer, inner) where the `n_inner` items of one outer index re-read the
// same operand panel: they take consecutive slots of a single XCD, so the panel crosses the fabric once.  The grid
// is padded to a multiple of 8 outers; the kernel drops outer >= n_outer.
constexpr int BKG_XCDS = 8;
struct XcdSlot { int outer, inner; };
__device__ __forceinline__ XcdSlot xcd_slot(unsigned block, int n_inner) {
  const int xcd = block % BKG_XCDS, slot = block / BKG_XCDS;
  return XcdSlot{(slot / n_inner) * BKG_XCDS + xcd, slot % n_inner};
}
static inline unsigned xcd_grid(int64_t n_outer, int64_t n_inner) {
  return (unsigned)(((n_outer + BKG_XCDS - 1) / BKG_XCDS) * BKG_XCDS * n_inner);
}

// ---------------------------------------------------------------- Philox4x32-7 (counter RNG)
// Used for dropout / feature-mask / edge-mask keep decisions generated inside kernels so that
// the forward and the backward pass can regenerate the same decision from (seed, element id)
// without a mask ever touching HBM.
__device__ __forceinline__ uint4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                            uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return make_uint4(c0, c1, c2, c3);
}

// keep decision for 4 consecutive elements starting at (4-aligned) linear element id `e`:
// element kept iff u32 >= thresh, thresh = p * 2^32.
__device__ __forceinline__ uint4 philox_keep4(uint64_t seed, uint64_t e, uint32_t stream_id) {
  uint64_t q = e >> 2;
  return philox4x32((uint32_t)q, (uint32_t)(q >> 32), stream_id, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// The same kind of keep decisions at a fifth of the cost, for ReDAF's dropout behind a K = 768 product (reference
// biomedkg/utils/fusion.py:84; round 6): Philox4x32-7 is 28 quarter-rate 32-bit multiplies per four decisions, six here --
// 0.49 -> 0.46 ms for the training call at the bench size; with no mixing at all it is 0.435 (docs/experiments.md).  Three rounds of
// murmur3's 32-bit finaliser over (element group, seed, stream) give two words = four 16-bit fields, returned in the TOP
// halves of four 32-bit values so that the callers' ``r >= p * 2^32`` comparisons stand (drop probability ceil(p * 2^16) /
// 2^16: 0.100006 for p = 0.1).  Counter-based like the Philox stream: every kernel that asks about an element gets the same
// answer.
__device__ __forceinline__ uint32_t keep_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu;
  x ^= x >> 13; x *= 0xc2b2ae35u;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint4 hash_keep4(uint64_t seed, uint64_t e, uint32_t stream_id) {
  const uint32_t q = (uint32_t)(e >> 2), qh = (uint32_t)(e >> 34);
  const uint32_t k0 = (uint32_t)seed ^ (stream_id * 0x9E3779B1u), k1 = (uint32_t)(seed >> 32) + qh * 0x85ebca6bu;
  const uint32_t a = keep_mix32(keep_mix32(q ^ k0) + k1);
  const uint32_t b = keep_mix32(a ^ ((q << 13) | (q >> 19)) ^ 0x68E31DA4u);
  return make_uint4(a << 16, a & 0xffff0000u, b << 16, b & 0xffff0000u);
}

// Device fills WITHOUT the runtime's memset: a hipMemsetAsync recorded into a hipGraph re-executes wrongly on this
// stack (ROCm 7.2: a 20 KB or 520 KB memset node leaves stale / non-zero words from the second replay on; 4 MB ones
// behave) -- every fill of the library is a kernel of its own, so that every entry point can be captured.
void fill_u32(void* p, uint32_t value, int64_t n_words, hipStream_t s);
void fill_zero_rows_f32(float* C, int64_t ldc, int64_t M, int64_t N, hipStream_t s);

// Replay epoch of captured (hipGraph) training steps (bkg_set_rng_epoch): a device counter the captured step bumps
// once per replay; every in-kernel Philox stream adds it (times an odd constant: a bijection of the 2^32 counter
// values, so a captured launch only meets the same stream again after 2^32 replays -- "x 65536" wrapped after 65,536
// steps, less than one long training run) to its stream id, so each replay of the same captured launch draws a fresh
// mask although its kernel arguments are frozen.  NULL outside graph mode.
const uint32_t* rng_epoch_ptr();
__device__ __forceinline__ uint32_t with_epoch(uint32_t stream_id, const uint32_t* __restrict__ epoch) {
  return epoch ? stream_id + epoch[0] * 0x9E3779B1u : stream_id;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace bkg
