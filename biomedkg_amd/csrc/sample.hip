// Device-side mini-batch samplers: the step before the hot path in the reference's own regime
// (SURVEY.md 8f rank 1).  The reference uses PyG loaders that sample on the host in C++ (pyg-lib /
// torch-sparse) and ship every batch over PCIe:
//   * NeighborLoader(num_neighbors=[30]*3)                       /root/reference/biomedkg/data_module.py:89-99
//   * GraphSAINTRandomWalkSampler(walk_length=10, num_steps=...) /root/reference/biomedkg/data_module.py:100-107
// Here the graph's CSRs already live in HBM, so sampling is two small kernels and the batch never leaves
// the device.  Randomness: Philox4x32-7 keyed by (seed, draw id) -- reproducible per seed, not bit-equal to
// the host RNG streams of the upstream loaders (their streams cannot be reproduced on a device).
// Replay (parity tests, oracle/samplers.py): every entry point takes an optional TAPE of raw 32-bit words that replaces
// the Philox stream -- walks: word [w, step]; neighbour picks: word [node id, Floyd step] (keyed by the NODE, not by its
// position in the frontier: a node is expanded once, so the tape means the same to a sampler that orders its frontier
// differently) -- mapped to an index by the same multiply-shift.  The CPU oracle consumes the same tape.
#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace bkg {

__device__ __forceinline__ uint32_t rng_u32(uint64_t seed, uint64_t ctr, uint32_t lane) {
  const uint4 r = philox4x32((uint32_t)ctr, (uint32_t)(ctr >> 32), lane, 0x5a17u, (uint32_t)seed, (uint32_t)(seed >> 32));
  return r.x;
}

// uniform integer in [0, n) without modulo bias worth speaking of (n <= 2^31): multiply-shift
__device__ __forceinline__ uint32_t below(uint32_t word, uint32_t n) { return (uint32_t)(((uint64_t)word * n) >> 32); }
__device__ __forceinline__ uint32_t rng_below(uint64_t seed, uint64_t ctr, uint32_t lane, uint32_t n) {
  return below(rng_u32(seed, ctr, lane), n);
}
constexpr int TAPE_STRIDE = 64;      // words per node of a neighbour-pick tape (= the largest fan-out)

// walks[w, 0] = start[w]; walks[w, s+1] = a uniformly chosen out-neighbour of walks[w, s] (stays put on a sink),
// over the SOURCE-sorted CSR (rowptr/col).  torch-sparse random_walk semantics.
__global__ void random_walk_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                   const int64_t* __restrict__ start, int64_t n_walks, int walk_length, uint64_t seed,
                                   const uint32_t* __restrict__ epoch, int64_t n_nodes, const uint32_t* __restrict__ tape,
                                   int64_t* __restrict__ walks) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_walks) return;
  if (epoch) seed += (uint64_t)epoch[0] * 0x9E3779B97F4A7C15ull;   // captured replays: a fresh stream per replay
  // roots: given, or uniform nodes from the same keyed stream (lane id no step uses)
  int64_t cur = start ? start[w] : (int64_t)rng_below(seed, (uint64_t)w, 0xFFFFFFFFu, (uint32_t)n_nodes);
  walks[w * (walk_length + 1)] = cur;
  for (int s = 0; s < walk_length; ++s) {
    const int beg = rowptr[cur], deg = rowptr[cur + 1] - beg;
    if (deg > 0)
      cur = col[beg + (tape ? below(tape[w * walk_length + s], (uint32_t)deg)
                            : rng_below(seed, (uint64_t)w, (uint32_t)s, (uint32_t)deg))];
    walks[w * (walk_length + 1) + s + 1] = cur;
  }
}

// For every frontier node f: up to `fanout` DISTINCT in-edges chosen uniformly (all of them when deg <= fanout),
// over the DESTINATION-sorted CSR.  out_pos[f, j] = CSR position of the j-th pick, or -1.
// One thread per frontier node; Floyd's algorithm (exactly `fanout` draws, no rejection loop).
// Floyd's draw of `fanout` distinct positions out of a row of `deg` > fanout entries: step s takes t uniform in [0, j],
// j = deg - fanout + s, or j itself if t was taken before (pyg-lib's neighbor_sample; oracle/samplers.py).  The picks live
// in REGISTERS: both loops are fully unrolled (a dynamically indexed array went to scratch -- 272 bytes per lane, up to 450
// dependent scratch reads per thread: the launch took 61 us for 25 k frontier rows, round 6: 8 us).  `fanout` is a kernel
// argument, so the guards are scalar branches.
template <int MAXF, typename Draw>
__device__ __forceinline__ void floyd_positions(int beg, int deg, int fanout, Draw draw, int32_t* __restrict__ out) {
  int picked[MAXF];
#pragma unroll
  for (int step = 0; step < MAXF; ++step) {
    if (step < fanout) {
      const int j = deg - fanout + step;
      const int t = draw(step, (uint32_t)(j + 1));
      bool taken = false;
#pragma unroll
      for (int q = 0; q < step; ++q) taken |= picked[q] == t;
      picked[step] = taken ? j : t;
    }
  }
#pragma unroll
  for (int step = 0; step < MAXF; ++step)
    if (step < fanout) out[step] = beg + picked[step];
}

template <int MAXF>
__global__ void sample_neighbors_kernel(const int32_t* __restrict__ rowptr, const int64_t* __restrict__ frontier,
                                        int64_t n_frontier, int fanout, uint64_t seed, uint32_t hop,
                                        const uint32_t* __restrict__ tape, int32_t* __restrict__ out_pos) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_frontier) return;
  const int64_t v = frontier[f];
  const int beg = rowptr[v], deg = rowptr[v + 1] - beg;
  int32_t* out = out_pos + f * fanout;
  if (deg <= fanout) {
    for (int j = 0; j < fanout; ++j) out[j] = j < deg ? beg + j : -1;
    return;
  }
  floyd_positions<MAXF>(beg, deg, fanout, [&](int step, uint32_t bound) {
    return (int)(tape ? below(tape[v * TAPE_STRIDE + step], bound) : rng_below(seed, (uint64_t)f * 64 + (uint64_t)step, hop, bound));
  }, out);
}

// ---- GraphSAINT batch = distinct walk nodes (ascending) + ALL edges among them, with no host round trip: every
// size stays in device memory (sizes[0] = nodes, sizes[1] = edges, sizes[2] = edges with head != tail), the outputs
// are padded to fixed capacities (node pads = n_nodes, i.e. the zero row appended to the feature table; edge pads =
// edges INTO the local dummy node `pad_node` -- never a real node -- from sources spread over the local ids and with
// types spread over [0, pad_types): no row of any structure built over them grows long), so a captured step can
// replay over them.
__global__ void saint_mark_kernel(const int64_t* __restrict__ walks, int64_t n, int32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag[walks[i]] = 1;
}

__global__ void saint_nodes_kernel(const int32_t* __restrict__ flag, const int32_t* __restrict__ offs, int64_t n_nodes,
                                   int64_t node_cap, int32_t* __restrict__ lut, int64_t* __restrict__ n_id,
                                   int64_t* __restrict__ sizes) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = offs[n_nodes - 1] + flag[n_nodes - 1];
  if (i < n_nodes) {
    const int p = flag[i] ? offs[i] : -1;
    lut[i] = p;
    if (p >= 0 && p < node_cap) n_id[p] = i;
  }
  if (i < node_cap && i >= total) n_id[i] = n_nodes;
  if (i == 0) { sizes[0] = total; sizes[2] = 0; }
}

__global__ void saint_edge_flags_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t n_edges,
                                        const int32_t* __restrict__ node_flag, int32_t* __restrict__ eflag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_edges) eflag[i] = node_flag[src[i]] & node_flag[dst[i]];
}

__global__ void saint_edges_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                   const int64_t* __restrict__ type, int64_t n_edges, const int32_t* __restrict__ eflag,
                                   const int32_t* __restrict__ offs, const int32_t* __restrict__ lut, int64_t edge_cap,
                                   int64_t pad_node, int64_t pad_types, int64_t* __restrict__ out_src,
                                   int64_t* __restrict__ out_dst,
                                   int64_t* __restrict__ out_type, int64_t* __restrict__ out_eid,
                                   int64_t* __restrict__ sizes) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = offs[n_edges - 1] + eflag[n_edges - 1];
  if (i < n_edges && eflag[i]) {
    const int64_t p = offs[i];
    if (p < edge_cap) {
      const int64_t s = src[i], d = dst[i];
      out_src[p] = lut[s];
      out_dst[p] = lut[d];
      out_type[p] = type ? type[i] : 0;
      out_eid[p] = i;
      if (s != d) atomicAdd(reinterpret_cast<unsigned long long*>(sizes + 2), 1ull);   // integer: order-independent
    }
  }
  if (i < edge_cap && i >= total) {
    out_src[i] = i % (pad_node + 1); out_dst[i] = pad_node; out_type[i] = i % pad_types; out_eid[i] = -1;
  }
  if (i == 0) sizes[1] = total;
}

// ---- NeighborLoader batch without a host round trip per hop: the frontier bounds and the edge count live in `ctr`
// (device: [0] = first frontier row in n_id, [1] = nodes so far, [2] = edges so far); every launch covers the capacity
// and exits past the counts.  Same draws, same node / edge order as the hop-by-hop host version (frontier-indexed
// Philox counters; new nodes appended in ascending id order; edges in (frontier row, pick) order).
__global__ void nb_begin_kernel(const int64_t* __restrict__ seeds, int64_t n_seeds, int32_t* __restrict__ seen,
                                int64_t* __restrict__ n_id, int64_t* __restrict__ ctr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_seeds) { n_id[i] = seeds[i]; seen[seeds[i]] = 1; }
  if (i == 0) { ctr[0] = 0; ctr[1] = n_seeds; ctr[2] = 0; }
}

template <int MAXF>
__global__ void nb_sample_kernel(const int32_t* __restrict__ rowptr, const int64_t* __restrict__ n_id,
                                 const int64_t* __restrict__ ctr, int fanout, uint64_t seed, uint32_t hop,
                                 const uint32_t* __restrict__ tape, int32_t* __restrict__ out_pos,
                                 int32_t* __restrict__ new_flag, int64_t n_nodes) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f < n_nodes) new_flag[f] = 0;      // (the launch covers n_nodes threads: the hop's "new node" marks start clear)
  const int64_t lo = ctr[0], cnt = ctr[1] - ctr[0];
  if (f >= cnt) return;
  const int64_t v = n_id[lo + f];
  const int beg = rowptr[v], deg = rowptr[v + 1] - beg;
  int32_t* out = out_pos + f * fanout;
  if (deg <= fanout) {
    for (int j = 0; j < fanout; ++j) out[j] = j < deg ? beg + j : -1;
    return;
  }
  floyd_positions<MAXF>(beg, deg, fanout, [&](int step, uint32_t bound) {      // (as sample_neighbors_kernel)
    return (int)(tape ? below(tape[v * TAPE_STRIDE + step], bound) : rng_below(seed, (uint64_t)f * 64 + (uint64_t)step, hop, bound));
  }, out);
}

// per candidate slot: valid pick?  (and mark the not-yet-seen sources)
__global__ void nb_flags_kernel(const int32_t* __restrict__ pos, const int32_t* __restrict__ col,
                                const int64_t* __restrict__ ctr, int fanout, int64_t slots,
                                const int32_t* __restrict__ seen, int32_t* __restrict__ new_flag,
                                int32_t* __restrict__ eflag) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= slots) return;
  const int64_t cnt = ctr[1] - ctr[0];
  int ok = 0;
  if (k / fanout < cnt) {
    const int32_t p = pos[k];
    if (p >= 0) {
      ok = 1;
      const int32_t s = col[p];
      if (!seen[s]) new_flag[s] = 1;
    }
  }
  eflag[k] = ok;
}

__global__ void nb_append_nodes_kernel(const int32_t* __restrict__ new_flag, const int32_t* __restrict__ offs,
                                       int64_t n_nodes, const int64_t* __restrict__ ctr, int32_t* __restrict__ seen,
                                       int64_t* __restrict__ n_id) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes || !new_flag[i]) return;
  n_id[ctr[1] + offs[i]] = i;
  seen[i] = 1;
}

__global__ void nb_append_edges_kernel(const int32_t* __restrict__ pos, const int32_t* __restrict__ eflag,
                                       const int32_t* __restrict__ offs, const int32_t* __restrict__ col,
                                       const int32_t* __restrict__ perm, const int64_t* __restrict__ n_id,
                                       const int64_t* __restrict__ ctr, int fanout, int64_t slots, int64_t edge_cap,
                                       int64_t* __restrict__ e_src, int64_t* __restrict__ e_dst,
                                       int64_t* __restrict__ e_id, const int32_t* __restrict__ bias) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= slots || !eflag[k]) return;
  const int64_t o = ctr[2] + offs[k] - (bias ? bias[0] : 0);      // (bias: the hop's new nodes, when ONE scan ran over both flag arrays)
  if (o >= edge_cap) return;
  const int32_t p = pos[k];
  e_src[o] = col[p];
  e_dst[o] = n_id[ctr[0] + k / fanout];
  e_id[o] = perm[p];
}

// after both appends: the new nodes become the frontier
__global__ void nb_advance_kernel(const int32_t* __restrict__ new_flag, const int32_t* __restrict__ node_offs,
                                  int64_t n_nodes, const int32_t* __restrict__ eflag, const int32_t* __restrict__ edge_offs,
                                  int64_t slots, int64_t* __restrict__ ctr, const int32_t* __restrict__ bias) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const int64_t new_nodes = node_offs[n_nodes - 1] + new_flag[n_nodes - 1];
  const int64_t new_edges = edge_offs[slots - 1] + eflag[slots - 1] - (bias ? bias[0] : 0);
  ctr[0] = ctr[1];
  ctr[1] += new_nodes;
  ctr[2] += new_edges;
}

__global__ void nb_lut_kernel(const int64_t* __restrict__ n_id, const int64_t* __restrict__ ctr, int64_t n_nodes,
                              int32_t* __restrict__ lut) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_nodes && i < ctr[1]) lut[n_id[i]] = (int32_t)i;
}

__global__ void nb_relabel_kernel(const int32_t* __restrict__ lut, const int64_t* __restrict__ ctr, int64_t edge_cap,
                                  int64_t* __restrict__ e_src, int64_t* __restrict__ e_dst) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= edge_cap || k >= ctr[2]) return;
  e_src[k] = lut[e_src[k]];
  e_dst[k] = lut[e_dst[k]];
}

}  // namespace bkg

using namespace bkg;

extern "C" int bkg_random_walk(const int32_t* rowptr, const int32_t* col, const int64_t* start, int64_t n_walks,
                               int32_t walk_length, uint64_t seed, const uint32_t* tape, int64_t* walks, void* stream) {
  BKG_ARG(n_walks >= 0 && walk_length >= 0, "bkg_random_walk: bad sizes");
  if (n_walks == 0) return 0;
  hipLaunchKernelGGL(random_walk_kernel, dim3(ceil_div(n_walks, 256)), dim3(256), 0, (hipStream_t)stream, rowptr, col,
                     start, n_walks, walk_length, seed, (const uint32_t*)nullptr, (int64_t)0, tape, walks);
  return check_launch("bkg_random_walk");
}

extern "C" int64_t bkg_saint_batch_workspace_bytes(int64_t n_nodes, int64_t n_edges) {
  size_t a = 0, b = 0;
  rocprim::exclusive_scan(nullptr, a, (int32_t*)nullptr, (int32_t*)nullptr, 0, (size_t)n_nodes, rocprim::plus<int32_t>());
  rocprim::exclusive_scan(nullptr, b, (int32_t*)nullptr, (int32_t*)nullptr, 0, (size_t)n_edges, rocprim::plus<int32_t>());
  return (int64_t)(a > b ? a : b) + 256;
}

// One GraphSAINT random-walk batch, entirely on the device (see the kernels above).  node_flag / node_offs / lut:
// int32[n_nodes] scratch (lut = local id or -1 survives the call); edge_flag / edge_offs: int32[n_edges] scratch.
extern "C" int bkg_saint_batch(const int32_t* rowptr, const int32_t* col, const int64_t* start, int64_t n_walks,
                               int32_t walk_length, uint64_t seed, const uint32_t* epoch, const uint32_t* tape,
                               int64_t* walks,
                               const int64_t* src, const int64_t* dst, const int64_t* type, int64_t n_nodes,
                               int64_t n_edges, int32_t* node_flag, int32_t* node_offs, int32_t* lut, int32_t* edge_flag,
                               int32_t* edge_offs, int64_t node_cap, int64_t edge_cap, int64_t pad_node, int64_t pad_types, int64_t* n_id,
                               int64_t* out_src, int64_t* out_dst, int64_t* out_type, int64_t* out_eid, int64_t* sizes,
                               void* workspace, int64_t workspace_bytes, void* stream) {
  BKG_ARG(n_walks > 0 && walk_length >= 0 && n_nodes > 0 && n_edges > 0 && node_cap > 0 && edge_cap > 0 && pad_types > 0 &&
              n_nodes < (1ll << 31) && n_edges < (1ll << 31), "bkg_saint_batch: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_entries = n_walks * (walk_length + 1);
  BKG_ARG(!tape || start, "bkg_saint_batch: a replayed tape needs explicit roots");
  hipLaunchKernelGGL(random_walk_kernel, dim3(ceil_div(n_walks, 256)), dim3(256), 0, s, rowptr, col, start, n_walks,
                     walk_length, seed, epoch, n_nodes, tape, walks);
  fill_u32(node_flag, 0u, n_nodes, s);
  hipError_t e = hipSuccess;
  hipLaunchKernelGGL(saint_mark_kernel, dim3(ceil_div(n_entries, 256)), dim3(256), 0, s, walks, n_entries, node_flag);
  size_t bytes = (size_t)workspace_bytes;
  e = rocprim::exclusive_scan(workspace, bytes, node_flag, node_offs, 0, (size_t)n_nodes, rocprim::plus<int32_t>(), s);
  if (e != hipSuccess) { set_error("bkg_saint_batch: node scan", e); return (int)e; }
  const int64_t tn = n_nodes > node_cap ? n_nodes : node_cap;
  hipLaunchKernelGGL(saint_nodes_kernel, dim3(ceil_div(tn, 256)), dim3(256), 0, s, node_flag, node_offs, n_nodes, node_cap,
                     lut, n_id, sizes);
  hipLaunchKernelGGL(saint_edge_flags_kernel, dim3(ceil_div(n_edges, 256)), dim3(256), 0, s, src, dst, n_edges, node_flag,
                     edge_flag);
  bytes = (size_t)workspace_bytes;
  e = rocprim::exclusive_scan(workspace, bytes, edge_flag, edge_offs, 0, (size_t)n_edges, rocprim::plus<int32_t>(), s);
  if (e != hipSuccess) { set_error("bkg_saint_batch: edge scan", e); return (int)e; }
  const int64_t te = n_edges > edge_cap ? n_edges : edge_cap;
  hipLaunchKernelGGL(saint_edges_kernel, dim3(ceil_div(te, 256)), dim3(256), 0, s, src, dst, type, n_edges, edge_flag,
                     edge_offs, lut, edge_cap, pad_node, pad_types, out_src, out_dst, out_type, out_eid, sizes);
  return check_launch("bkg_saint_batch");
}

extern "C" int bkg_sample_neighbors(const int32_t* rowptr, const int64_t* frontier, int64_t n_frontier, int32_t fanout,
                                    uint64_t seed, uint32_t hop, const uint32_t* tape, int32_t* out_pos, void* stream) {
  BKG_ARG(n_frontier >= 0 && fanout > 0 && fanout <= 64, "bkg_sample_neighbors: 1 <= fanout <= 64");
  if (n_frontier == 0) return 0;
  if (fanout <= 32)
    hipLaunchKernelGGL(sample_neighbors_kernel<32>, dim3(ceil_div(n_frontier, 256)), dim3(256), 0, (hipStream_t)stream,
                       rowptr, frontier, n_frontier, fanout, seed, hop, tape, out_pos);
  else
    hipLaunchKernelGGL(sample_neighbors_kernel<64>, dim3(ceil_div(n_frontier, 256)), dim3(256), 0, (hipStream_t)stream,
                       rowptr, frontier, n_frontier, fanout, seed, hop, tape, out_pos);
  return check_launch("bkg_sample_neighbors");
}

// ---- NeighborLoader batches with ONE host read at the end (bkg_neighbor_begin, one bkg_neighbor_hop per hop,
// bkg_neighbor_finish; ctr[1] = nodes, ctr[2] = edges).  Capacities: n_id / seen / lut / node scratch: n_nodes; pos /
// edge scratch: n_nodes * fanout slots; e_src / e_dst / e_id: edge_cap (every node is expanded at most once, so
// n_nodes * max fanout always suffices).
extern "C" int64_t bkg_neighbor_workspace_bytes(int64_t n_nodes, int32_t max_fanout) {
  size_t a = 0;
  rocprim::exclusive_scan(nullptr, a, (int32_t*)nullptr, (int32_t*)nullptr, 0, (size_t)(n_nodes * (max_fanout + 1)),
                          rocprim::plus<int32_t>());      // (+ 1: the node flags in front of the slots' when one scan takes both)
  return (int64_t)a + 256;
}

extern "C" int bkg_neighbor_begin(const int64_t* seeds, int64_t n_seeds, int64_t n_nodes, int32_t* seen, int64_t* n_id,
                                  int64_t* ctr, void* stream) {
  BKG_ARG(n_seeds > 0 && n_seeds <= n_nodes, "bkg_neighbor_begin: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  fill_u32(seen, 0u, n_nodes, s);
  hipLaunchKernelGGL(nb_begin_kernel, dim3(ceil_div(n_seeds, 256)), dim3(256), 0, s, seeds, n_seeds, seen, n_id, ctr);
  return check_launch("bkg_neighbor_begin");
}

extern "C" int bkg_neighbor_hop(const int32_t* rowptr, const int32_t* col, const int32_t* perm, int64_t n_nodes,
                                int32_t fanout, uint64_t seed, uint32_t hop, const uint32_t* tape, int32_t* seen,
                                int64_t* n_id, int64_t* ctr,
                                int32_t* pos, int32_t* new_flag, int32_t* node_offs, int32_t* eflag, int32_t* edge_offs,
                                int64_t edge_cap, int64_t* e_src, int64_t* e_dst, int64_t* e_id, void* workspace,
                                int64_t workspace_bytes, void* stream) {
  BKG_ARG(n_nodes > 0 && fanout > 0 && fanout <= 64 && n_nodes * fanout < (1ll << 31), "bkg_neighbor_hop: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  const int64_t slots = n_nodes * fanout;
  if (fanout <= 32)
    hipLaunchKernelGGL(nb_sample_kernel<32>, dim3(ceil_div(n_nodes, 256)), dim3(256), 0, s, rowptr, n_id, ctr, fanout, seed,
                       hop, tape, pos, new_flag, n_nodes);
  else
    hipLaunchKernelGGL(nb_sample_kernel<64>, dim3(ceil_div(n_nodes, 256)), dim3(256), 0, s, rowptr, n_id, ctr, fanout, seed,
                       hop, tape, pos, new_flag, n_nodes);
  hipError_t e = hipSuccess;
  hipLaunchKernelGGL(nb_flags_kernel, dim3(ceil_div(slots, 256)), dim3(256), 0, s, pos, col, ctr, fanout, slots, seen,
                     new_flag, eflag);
  // the slots' flags right behind the nodes' (and the offsets alike): ONE scan over both -- the slots' offsets then start at
  // the hop's number of new nodes, edge_offs[0], which the consumers take off (two launches less per hop, round 6)
  const bool merged = eflag == new_flag + n_nodes && edge_offs == node_offs + n_nodes && n_nodes + slots < (1ll << 31);
  size_t bytes = (size_t)workspace_bytes;
  if (merged) {
    e = rocprim::exclusive_scan(workspace, bytes, new_flag, node_offs, 0, (size_t)(n_nodes + slots), rocprim::plus<int32_t>(), s);
    if (e != hipSuccess) { set_error("bkg_neighbor_hop: scan", e); return (int)e; }
  } else {
    e = rocprim::exclusive_scan(workspace, bytes, new_flag, node_offs, 0, (size_t)n_nodes, rocprim::plus<int32_t>(), s);
    if (e != hipSuccess) { set_error("bkg_neighbor_hop: node scan", e); return (int)e; }
    bytes = (size_t)workspace_bytes;
    e = rocprim::exclusive_scan(workspace, bytes, eflag, edge_offs, 0, (size_t)slots, rocprim::plus<int32_t>(), s);
    if (e != hipSuccess) { set_error("bkg_neighbor_hop: edge scan", e); return (int)e; }
  }
  const int32_t* bias = merged ? edge_offs : nullptr;
  hipLaunchKernelGGL(nb_append_edges_kernel, dim3(ceil_div(slots, 256)), dim3(256), 0, s, pos, eflag, edge_offs, col, perm,
                     n_id, ctr, fanout, slots, edge_cap, e_src, e_dst, e_id, bias);
  hipLaunchKernelGGL(nb_append_nodes_kernel, dim3(ceil_div(n_nodes, 256)), dim3(256), 0, s, new_flag, node_offs, n_nodes,
                     ctr, seen, n_id);
  hipLaunchKernelGGL(nb_advance_kernel, dim3(1), dim3(64), 0, s, new_flag, node_offs, n_nodes, eflag, edge_offs, slots, ctr, bias);
  return check_launch("bkg_neighbor_hop");
}

extern "C" int bkg_neighbor_finish(int64_t n_nodes, const int64_t* n_id, const int64_t* ctr, int32_t* lut,
                                   int64_t edge_cap, int64_t* e_src, int64_t* e_dst, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(nb_lut_kernel, dim3(ceil_div(n_nodes, 256)), dim3(256), 0, s, n_id, ctr, n_nodes, lut);
  hipLaunchKernelGGL(nb_relabel_kernel, dim3(ceil_div(edge_cap, 256)), dim3(256), 0, s, lut, ctr, edge_cap, e_src, e_dst);
  return check_launch("bkg_neighbor_finish");
}
