// Device-side mini-batch samplers: the step before the hot path in the reference's own regime
// (SURVEY.md 8f rank 1).  The reference uses PyG loaders that sample on the host in C++ (pyg-lib /
// torch-sparse) and ship every batch over PCIe:
//   * NeighborLoader(num_neighbors=[30]*3)                       /root/reference/biomedkg/data_module.py:89-99
//   * GraphSAINTRandomWalkSampler(walk_length=10, num_steps=...) /root/reference/biomedkg/data_module.py:100-107
// Here the graph's CSRs already live in HBM, so sampling is two small kernels and the batch never leaves
// the device.  Randomness: Philox4x32-7 keyed by (seed, draw id) -- reproducible per seed, not bit-equal to
// the host RNG streams of the upstream loaders (their streams cannot be reproduced on a device).
#include "common.h"

namespace bkg {

__device__ __forceinline__ uint32_t rng_u32(uint64_t seed, uint64_t ctr, uint32_t lane) {
  const uint4 r = philox4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), lane, 0x5a17u, (uint32_t)seed, (uint32_t)(seed >> 32));
  return r.x;
}

// uniform integer in [0, n) without modulo bias worth speaking of (n <= 2^31): multiply-shift
__device__ __forceinline__ uint32_t rng_below(uint64_t seed, uint64_t ctr, uint32_t lane, uint32_t n) {
  return (uint32_t)(((uint64_t)rng_u32(seed, ctr, lane) * n) >> 32);
}

// walks[w, 0] = start[w]; walks[w, s+1] = a uniformly chosen out-neighbour of walks[w, s] (stays put on a sink),
// over the SOURCE-sorted CSR (rowptr/col).  torch-sparse random_walk semantics.
__global__ void random_walk_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                   const int64_t* __restrict__ start, int64_t n_walks, int walk_length, uint64_t seed,
                                   int64_t* __restrict__ walks) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_walks) return;
  int64_t cur = start[w];
  walks[w * (walk_length + 1)] = cur;
  for (int s = 0; s < walk_length; ++s) {
    const int beg = rowptr[cur], deg = rowptr[cur + 1] - beg;
    if (deg > 0) cur = col[beg + rng_below(seed, (uint64_t)w, (uint32_t)s, (uint32_t)deg)];
    walks[w * (walk_length + 1) + s + 1] = cur;
  }
}

// For every frontier node f: up to `fanout` DISTINCT in-edges chosen uniformly (all of them when deg <= fanout),
// over the DESTINATION-sorted CSR.  out_pos[f, j] = CSR position of the j-th pick, or -1.
// One thread per frontier node; Floyd's algorithm (exactly `fanout` draws, no rejection loop).
template <int MAXF>
__global__ void sample_neighbors_kernel(const int32_t* __restrict__ rowptr, const int64_t* __restrict__ frontier,
                                        int64_t n_frontier, int fanout, uint64_t seed, uint32_t hop,
                                        int32_t* __restrict__ out_pos) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_frontier) return;
  const int64_t v = frontier[f];
  const int beg = rowptr[v], deg = rowptr[v + 1] - beg;
  int32_t* out = out_pos + f * fanout;
  if (deg <= fanout) {
    for (int j = 0; j < fanout; ++j) out[j] = j < deg ? beg + j : -1;
    return;
  }
  int picked[MAXF];
  int n = 0;
  for (int j = deg - fanout; j < deg; ++j) {   // Floyd: t uniform in [0, j]; take t, or j if t was taken
    int t = (int)rng_below(seed, (uint64_t)f * 64 + (uint64_t)(j - (deg - fanout)), hop, (uint32_t)(j + 1));
    bool seen = false;
    for (int q = 0; q < n; ++q) seen |= picked[q] == t;
    picked[n++] = seen ? j : t;
  }
  for (int j = 0; j < fanout; ++j) out[j] = beg + picked[j];
}

}  // namespace bkg

using namespace bkg;

extern "C" int bkg_random_walk(const int32_t* rowptr, const int32_t* col, const int64_t* start, int64_t n_walks,
                               int32_t walk_length, uint64_t seed, int64_t* walks, void* stream) {
  BKG_ARG(n_walks >= 0 && walk_length >= 0, "bkg_random_walk: bad sizes");
  if (n_walks == 0) return 0;
  hipLaunchKernelGGL(random_walk_kernel, dim3(ceil_div(n_walks, 256)), dim3(256), 0, (hipStream_t)stream, rowptr, col,
                     start, n_walks, walk_length, seed, walks);
  return check_launch("bkg_random_walk");
}

extern "C" int bkg_sample_neighbors(const int32_t* rowptr, const int64_t* frontier, int64_t n_frontier, int32_t fanout,
                                    uint64_t seed, uint32_t hop, int32_t* out_pos, void* stream) {
  BKG_ARG(n_frontier >= 0 && fanout > 0 && fanout <= 64, "bkg_sample_neighbors: 1 <= fanout <= 64");
  if (n_frontier == 0) return 0;
  hipLaunchKernelGGL(sample_neighbors_kernel<64>, dim3(ceil_div(n_frontier, 256)), dim3(256), 0, (hipStream_t)stream,
                     rowptr, frontier, n_frontier, fanout, seed, hop, out_pos);
  return check_launch("bkg_sample_neighbors");
}
