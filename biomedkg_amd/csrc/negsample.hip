// Device negative sampling for the KGE step (reference biomedkg/kge_module.py:86-102: PyG 2.5.3 negative_sampling +
// a randomly permuted copy of the positives' relation ids).  PyG's population is the N*(N-1) OFF-DIAGONAL cells
// (edge_index_to_vector, non-bipartite directed form: self-loop positives are dropped, cell (r, c) has the code
// r*(N-1) + c - (r < c); vector_to_edge_index re-opens the diagonal with c += (r <= c)), so no negative is ever a
// self pair.  PyG draws `random.sample(range(N*(N-1)), m)` on the host and tests membership with np.isin; the first
// device version of this path did the same with sorts and binary searches (two 8-pass radix sorts + a 22-level
// searchsorted per step).  Nothing is sorted here:
//   * sampling WITHOUT replacement = a prefix of a random permutation: candidate i is pi(i) for a keyed bijection pi
//     of [0, N*(N-1)) (6-round Feistel network on the enclosing even power of two + cycle walking), so candidates are
//     distinct by construction -- random.sample's guarantee -- and later rounds just continue the prefix;
//   * membership in the positives is a lookup in an open-addressing hash table built once per graph
//     (bkg_hashset_insert) instead of a binary search;
//   * survivors are compacted in candidate order (flag -> rocPRIM exclusive scan -> scatter), so the result is a
//     pure function of (graph, seed): bitwise repeatable, no atomics;
//   * bkg_permute_gather_i64: out[i] = src[pi(i)] with the same kind of bijection over [0, n) -- the
//     `x[torch.randperm(n)]` of the reference without the sort behind randperm.
#include <cstring>
#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace bkg {

constexpr unsigned long long HS_EMPTY = ~0ull;

__device__ __host__ __forceinline__ uint64_t hs_mix(uint64_t x) {   // splitmix64 finaliser
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27; x *= 0x94d049bb133111ebull;
  x ^= x >> 31;
  return x;
}

// 0: inserted, 1: was present
__device__ __forceinline__ int hs_insert(unsigned long long* __restrict__ tab, uint64_t mask, uint64_t key) {
  uint64_t i = hs_mix(key) & mask;
  for (;;) {
    const unsigned long long prev = atomicCAS(tab + i, HS_EMPTY, (unsigned long long)key);
    if (prev == HS_EMPTY) return 0;
    if (prev == key) return 1;
    i = (i + 1) & mask;
  }
}

__device__ __forceinline__ bool hs_contains(const unsigned long long* __restrict__ tab, uint64_t mask, uint64_t key) {
  uint64_t i = hs_mix(key) & mask;
  for (;;) {
    const unsigned long long v = tab[i];
    if (v == key) return true;
    if (v == HS_EMPTY) return false;
    i = (i + 1) & mask;
  }
}

// Keyed bijection of [0, n): balanced Feistel network over 2*half bits (a permutation of [0, 4^half) whatever the
// round function is), walked until the image falls back into [0, n) -- that keeps it a bijection of [0, n); the
// enclosing domain is < 4 n, so the walk takes < 4 steps on average.
struct Feistel {
  uint64_t n, key[6];
  int half;
  __device__ __forceinline__ uint64_t operator()(uint64_t x) const {
    const uint64_t m = (1ull << half) - 1ull;
    do {
      uint64_t l = x >> half, r = x & m;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const uint64_t f = hs_mix(r ^ key[i]) >> 17;
        const uint64_t t = l ^ (f & m);
        l = r; r = t;
      }
      x = (l << half) | r;
    } while (x >= n);
    return x;
  }
};

__host__ __device__ static inline Feistel make_feistel(uint64_t n, uint64_t seed) {
  Feistel f;
  f.n = n;
  int bits = 2;
  while (bits < 64 && (1ull << bits) < n) bits += 2;
  f.half = bits / 2;
  for (int i = 0; i < 6; ++i) f.key[i] = hs_mix(seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1));
  return f;
}

// The same network with a 32-bit round function (murmur3's finaliser under a 32-bit round key: two v_mul_lo_u32 per round
// where the splitmix64 finaliser costs two 64-BIT multiplies, eight quarter-rate 32-bit ones) for the id shuffles
// (round 6: ``edge_type.repeat(10)[randperm]`` -- 40 M evaluations -- took 0.85 ms, now 0.37).  The halves have at most 32
// bits.  The samplers and the corruption permutation keep the network above (their streams are what the tests' training
// trajectories were recorded with).
__device__ __host__ __forceinline__ uint32_t fs_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu;
  x ^= x >> 13; x *= 0xc2b2ae35u;
  x ^= x >> 16;
  return x;
}
struct Feistel32 {
  uint64_t n;
  uint32_t key[6];
  int half;
  __device__ __forceinline__ uint64_t operator()(uint64_t x) const {
    const uint32_t m = half >= 32 ? 0xffffffffu : ((1u << half) - 1u);
    do {
      uint32_t l = (uint32_t)(x >> half), r = (uint32_t)x & m;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const uint32_t t = l ^ (fs_mix32(r ^ key[i]) & m);
        l = r; r = t;
      }
      x = ((uint64_t)l << half) | r;
    } while (x >= n);
    return x;
  }
};
__host__ __device__ static inline Feistel32 make_feistel32(uint64_t n, uint64_t seed) {
  Feistel32 f;
  f.n = n;
  int bits = 2;
  while (bits < 64 && (1ull << bits) < n) bits += 2;
  f.half = bits / 2;
  for (int i = 0; i < 6; ++i) f.key[i] = (uint32_t)(hs_mix(seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1)) >> 29);
  return f;
}

// PyG edge_index_to_vector (non-bipartite, directed): self loops are not positives of the off-diagonal population;
// cell (r, c), r != c, has the code r*(N-1) + c - (r < c).
__global__ __launch_bounds__(256) void hashset_insert_kernel(const int64_t* __restrict__ head, const int64_t* __restrict__ tail,
                                                             int64_t n_nodes, int64_t n, unsigned long long* __restrict__ tab,
                                                             uint64_t mask) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t r = head[i], c = tail[i];
  if (r != c) hs_insert(tab, mask, (uint64_t)(r * (n_nodes - 1) + c - (r < c ? 1 : 0)));
}

__global__ __launch_bounds__(256) void negative_candidates_kernel(Feistel pi, int64_t first, int64_t n_cand,
                                                                  const unsigned long long* __restrict__ pos_tab,
                                                                  uint64_t pos_mask, uint64_t* __restrict__ keys,
                                                                  int32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_cand) return;
  const uint64_t key = pi((uint64_t)(first + i));
  keys[i] = key;
  flag[i] = hs_contains(pos_tab, pos_mask, key) ? 0 : 1;
}

__global__ __launch_bounds__(256) void negative_scatter_kernel(const uint64_t* __restrict__ keys,
                                                               const int32_t* __restrict__ flag,
                                                               const int32_t* __restrict__ offs, int64_t n_cand,
                                                               uint64_t n_nodes, int64_t out_offset, int64_t max_out,
                                                               int64_t* __restrict__ out_head, int64_t* __restrict__ out_tail,
                                                               int32_t* __restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_cand) return;
  if (flag[i]) {
    const int64_t p = out_offset + offs[i];
    if (p < max_out) {                    // PyG vector_to_edge_index: row = code / (N-1), col = code % (N-1), col += (row <= col)
      const uint64_t r = keys[i] / (n_nodes - 1), c = keys[i] % (n_nodes - 1);
      out_head[p] = (int64_t)r;
      out_tail[p] = (int64_t)(c + (r <= c ? 1 : 0));
    }
  }
  // slots the survivors did not reach hold (0, 0) -- never whatever the buffer held (the one-round caller does not read
  // the count: a shortfall, ~1e-9 per call, must not leave stale ids); a later round of the caller writes over them
  const int64_t total = (int64_t)offs[n_cand - 1] + flag[n_cand - 1], q = out_offset + total + i;
  if (q < max_out) { out_head[q] = 0; out_tail[q] = 0; }
  if (i == n_cand - 1) *count = (int32_t)total;
}

__global__ __launch_bounds__(256) void permute_gather_kernel(const int64_t* __restrict__ src, int64_t n, Feistel32 pi,
                                                             int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = src[pi((uint64_t)i)];
}

// out[i] = src[pi(i) % n_src] over i < n_src * repeat: ``src.repeat(repeat)[randperm(n_src * repeat)]`` -- the negatives'
// relation ids of reference kge_module.py:97-100 -- without the repeated copy
template <typename S>
__global__ __launch_bounds__(256) void permute_gather_repeat_kernel(const S* __restrict__ src, int64_t n_src, int64_t n,
                                                                    Feistel32 pi, int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (int64_t)src[(int64_t)(pi((uint64_t)i) % (uint64_t)n_src)];
}
__global__ __launch_bounds__(256) void narrow_i64_u8_kernel(const int64_t* __restrict__ src, int64_t n, uint8_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (uint8_t)src[i];
}

// ---- the same sampler with every extent in DEVICE memory (captured training steps over padded mini-batches,
// biomedkg_amd/sampled.py): sizes[0] = N (nodes of the batch), sizes[1] = E (its edges), sizes[2] = E' (edges with
// head != tail); k = neg_ratio * E negatives are asked for.  One round of int(1.1 k / prob) candidates (PyG's first
// round; the later rounds only matter when fewer than k survive, which the caller rules out on the host: prob > 0.9).
// The permutation key mixes a device replay counter, so a replayed graph draws fresh negatives.
__device__ __forceinline__ uint64_t dev_seed(uint64_t seed, const uint32_t* __restrict__ epoch, uint64_t salt) {
  return hs_mix(seed ^ (salt * 0xD1B54A32D192ED03ull) ^ ((uint64_t)(epoch ? epoch[0] : 0u) << 32));
}

__global__ __launch_bounds__(256) void hashset_insert_dev_kernel(const int64_t* __restrict__ head,
                                                                 const int64_t* __restrict__ tail,
                                                                 const int64_t* __restrict__ sizes, int64_t rows,
                                                                 unsigned long long* __restrict__ tab, uint64_t mask) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows || i >= sizes[1]) return;
  const int64_t r = head[i], c = tail[i], n_nodes = sizes[0];
  if (r != c) hs_insert(tab, mask, (uint64_t)(r * (n_nodes - 1) + c - (r < c ? 1 : 0)));
}

__device__ __forceinline__ int64_t dev_candidates(const int64_t* __restrict__ sizes, int64_t neg_ratio, int64_t cap) {
  const int64_t n = sizes[0], k = neg_ratio * sizes[1];
  const double population = (double)(n * (n - 1));
  const double prob = 1.0 - (double)sizes[2] / population;
  int64_t m = (int64_t)(1.1 * (double)k / prob);                     // PyG: sample_size = int(1.1 * k / prob)
  if (m > n * (n - 1)) m = n * (n - 1);
  return m < cap ? m : cap;
}

__global__ __launch_bounds__(256) void negative_candidates_dev_kernel(const int64_t* __restrict__ sizes, int64_t neg_ratio,
                                                                      uint64_t seed, const uint32_t* __restrict__ epoch,
                                                                      int64_t cap,
                                                                      const unsigned long long* __restrict__ pos_tab,
                                                                      uint64_t pos_mask, uint64_t* __restrict__ keys,
                                                                      int32_t* __restrict__ flag) {
  __shared__ Feistel pi;          // the block's copy of the bijection and of the candidate count: built by one lane
  __shared__ int64_t n_cand;      // (the count takes a double division, the keys six mixes)
  if (threadIdx.x == 0) {
    const int64_t n = sizes[0];
    n_cand = n < 2 ? 0 : dev_candidates(sizes, neg_ratio, cap);
    if (n >= 2) pi = make_feistel((uint64_t)(n * (n - 1)), dev_seed(seed, epoch, 1));
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= cap) return;
  if (i >= n_cand) { flag[i] = 0; return; }
  const uint64_t key = pi((uint64_t)i);
  keys[i] = key;
  flag[i] = hs_contains(pos_tab, pos_mask, key) ? 0 : 1;
}

// survivors in candidate order, the first k of them; slots past the count get pairs spread over the batch's nodes
// (the caller gives them a relation id no scorer visits; spread, so that no node's incidence list grows long).
// count = min(survivors, k).
__global__ __launch_bounds__(256) void negative_scatter_dev_kernel(const uint64_t* __restrict__ keys,
                                                                   const int32_t* __restrict__ flag,
                                                                   const int32_t* __restrict__ offs, int64_t cap,
                                                                   const int64_t* __restrict__ sizes, int64_t neg_ratio,
                                                                   int64_t out_cap, int64_t* __restrict__ out_head,
                                                                   int64_t* __restrict__ out_tail, int64_t* __restrict__ count) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t k = min(neg_ratio * sizes[1], out_cap);
  const int64_t total = offs[cap - 1] + flag[cap - 1];
  const int64_t have = total < k ? total : k;
  if (i < cap && flag[i] && offs[i] < k) {
    const uint64_t n1 = (uint64_t)sizes[0] - 1;
    const uint64_t r = keys[i] / n1, c = keys[i] % n1;
    out_head[offs[i]] = (int64_t)r;
    out_tail[offs[i]] = (int64_t)(c + (r <= c ? 1 : 0));
  }
  if (i < out_cap && i >= have) {
    const int64_t n = sizes[0] > 1 ? sizes[0] : 2;
    out_head[i] = i % n; out_tail[i] = (i + 1) % n;
  }
  if (i == 0) *count = have;
}

// out[i] = src[pi(i) % E] for i < count (pi: keyed bijection of [0, neg_ratio * E)) -- the reference's
// `edge_type.repeat(neg_ratio)[randperm]` -- and `pad` beyond.
__global__ __launch_bounds__(256) void permute_gather_dev_kernel(const int64_t* __restrict__ src,
                                                                 const int64_t* __restrict__ sizes, int64_t neg_ratio,
                                                                 uint64_t seed, const uint32_t* __restrict__ epoch,
                                                                 const int64_t* __restrict__ count, int64_t pad,
                                                                 int64_t out_cap, int64_t* __restrict__ out) {
  __shared__ Feistel pi;
  const int64_t e = sizes[1];
  if (threadIdx.x == 0 && e > 0) pi = make_feistel((uint64_t)(neg_ratio * e), dev_seed(seed, epoch, 2));
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= out_cap) return;
  if (i >= count[0] || e <= 0) { out[i] = pad; return; }
  out[i] = src[pi((uint64_t)i) % (uint64_t)e];
}

// out[i] = pi(i) for i < n[0], pi a keyed bijection of [0, n[0]) (n in DEVICE memory), and i beyond: torch.randperm over
// the real rows of a padded batch, the pads staying where they are (GGD's corruption inside a captured mini-batch step).
__global__ __launch_bounds__(256) void randperm_dev_kernel(const int64_t* __restrict__ n_dev, int64_t cap, uint64_t seed,
                                                           const uint32_t* __restrict__ epoch, int64_t* __restrict__ out) {
  __shared__ Feistel pi;
  const int64_t n = n_dev[0] < cap ? n_dev[0] : cap;
  if (threadIdx.x == 0 && n > 0) pi = make_feistel((uint64_t)n, dev_seed(seed, epoch, 3));
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= cap) return;
  out[i] = i < n ? (int64_t)pi((uint64_t)i) : i;
}

// The loss-side view of a padded batch in one launch: pad positives get the relation id `pad_rel` (one past the real
// ones: no scorer wave visits it) and endpoints spread over the local ids (zero gradient records, no long incidence
// list); inv_total = 1 / (edges + negatives).
// The padded view of a device NeighborLoader batch (biomedkg_amd/sampled_gcl.py): node ids past the batch's node count
// read the zero row `pad_row`, edges past its edge count become self loops of `last_node` (weight 0 under gcn_norm; behind
// the real edges, whose targets ascend, so the list stays sorted by target), real[i] = i < nodes.  One launch for what
// was six torch expressions inside every captured step.
__global__ __launch_bounds__(256) void padded_gcl_batch_kernel(const int64_t* __restrict__ sizes,
                                                               const int64_t* __restrict__ n_id,
                                                               const int64_t* __restrict__ src,
                                                               const int64_t* __restrict__ dst, int64_t n_cap, int64_t e_b,
                                                               int64_t pad_row, int64_t last_node,
                                                               int64_t* __restrict__ out_n_id, int64_t* __restrict__ out_src,
                                                               int64_t* __restrict__ out_dst, uint8_t* __restrict__ real) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nodes = sizes[1], edges = sizes[2];
  if (i < n_cap) {
    const bool live = i < nodes;
    out_n_id[i] = live ? n_id[i] : pad_row;
    real[i] = live ? 1 : 0;
  }
  if (i < e_b) {
    const bool live = i < edges;
    out_src[i] = live ? src[i] : last_node;
    out_dst[i] = live ? dst[i] : last_node;
  }
}

__global__ __launch_bounds__(256) void padded_loss_inputs_kernel(const int64_t* __restrict__ sizes,
                                                                 const int64_t* __restrict__ count,
                                                                 const int64_t* __restrict__ src,
                                                                 const int64_t* __restrict__ dst,
                                                                 const int64_t* __restrict__ type, int64_t rows,
                                                                 int64_t node_cap, int64_t pad_rel,
                                                                 int64_t* __restrict__ out_src, int64_t* __restrict__ out_dst,
                                                                 int64_t* __restrict__ out_type, float* __restrict__ inv_total) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i == 0) inv_total[0] = 1.0f / (float)(sizes[1] + count[0]);
  if (i >= rows) return;
  const bool valid = i < sizes[1];
  out_src[i] = valid ? src[i] : i % node_cap;
  out_dst[i] = valid ? dst[i] : (i + 1) % node_cap;
  out_type[i] = valid ? type[i] : pad_rel;
}


// ---- block-structured sampling for node-partitioned training (biomedkg_amd/partition.py, both-owner scoring).  The
// off-diagonal cell grid is cut into world x world blocks (head owner a, tail owner b); block (a, b) draws its share
// k_ab = K * free_ab / free of the K negatives -- free = cells that are not positives, counted once per graph, so the
// union is K distinct uniform non-edges stratified by block with proportional allocation -- from ITS OWN keyed
// permutation of the block's cells, seeded by (seed, a, b).  Every rank can therefore produce any block: rank r draws the
// 2 world - 1 blocks that touch its nodes (row band r: it owns the heads; column band r: it owns the tails), the two
// owners of a block get identical triplets, and no triplet, score or gradient row ever crosses the wire.  The relation id
// of the negative at global position p (blocks laid out one after the other in block-id order) is
// type[sigma(p) % E], sigma a keyed bijection of [0, K): the reference's edge_type.repeat(k)[randperm(K)].
struct NegBlock {
  int64_t row_lo, n_rows, col_lo, n_cols;   // the block's cells; a diagonal block has row_lo == col_lo, n_rows == n_cols
  int64_t population;                       // n_rows * n_cols, or n (n - 1) on the diagonal
  int64_t cand0, n_cand;                    // its candidates: [cand0, cand0 + n_cand) of the launch, cand0 % 256 == 0
  int64_t out0, k;                          // its survivors go to out[out0 .. out0 + k)
  int64_t gpos0;                            // global position of its first negative (relation ids)
  int64_t id;                               // a * world + b (seeds its permutation)
  int64_t row_shift, col_shift;             // added to the head / tail id on the way out (node id -> slot of the gathered table)
};

__device__ __forceinline__ int neg_block_of(const NegBlock* __restrict__ blocks, int n_blocks, int64_t i) {
  int b = 0;
  for (int j = 1; j < n_blocks; ++j) b = (i >= blocks[j].cand0) ? j : b;     // (cand0 ascending; <= 2 world - 1 entries)
  return b;
}

__global__ __launch_bounds__(256) void negative_candidates_blocks_kernel(const NegBlock* __restrict__ blocks, int n_blocks,
                                                                         uint64_t seed, const uint32_t* __restrict__ epoch,
                                                                         uint64_t n_nodes,
                                                                         const unsigned long long* __restrict__ pos_tab,
                                                                         uint64_t pos_mask, uint64_t* __restrict__ keys,
                                                                         int32_t* __restrict__ flag) {
  __shared__ Feistel pi;
  __shared__ NegBlock blk;
  if (threadIdx.x == 0) {
    blk = blocks[neg_block_of(blocks, n_blocks, (int64_t)blockIdx.x * 256)];
    // (a captured step's kernel arguments are frozen: the library's device replay counter re-keys the draw -- the same
    // counter value on every rank, which all replay once per step, so the two owners of a block still agree)
    const uint64_t s_eff = epoch ? seed ^ ((uint64_t)epoch[0] * 0x9E3779B97F4A7C15ull) : seed;
    pi = make_feistel((uint64_t)blk.population, hs_mix(s_eff ^ ((uint64_t)(blk.id + 1) * 0xD1B54A32D192ED03ull)));
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t j = i - blk.cand0;
  if (j >= blk.n_cand) { flag[i] = 0; keys[i] = 0; return; }
  const uint64_t code = pi((uint64_t)j);
  uint64_t r, c;
  if (blk.row_lo == blk.col_lo) {          // diagonal block: its own off-diagonal cells, PyG's coding inside the block
    const uint64_t n1 = (uint64_t)blk.n_rows - 1;
    r = code / n1; c = code % n1; c += (r <= c ? 1 : 0);
  } else {
    r = code / (uint64_t)blk.n_cols; c = code % (uint64_t)blk.n_cols;
  }
  r += (uint64_t)blk.row_lo; c += (uint64_t)blk.col_lo;
  const uint64_t key = r * (n_nodes - 1) + c - (r < c ? 1 : 0);     // the graph-wide off-diagonal code (hash set, output)
  keys[i] = key;
  flag[i] = hs_contains(pos_tab, pos_mask, key) ? 0 : 1;
}

__global__ __launch_bounds__(256) void negative_scatter_blocks_kernel(const NegBlock* __restrict__ blocks, int n_blocks,
                                                                      const uint64_t* __restrict__ keys,
                                                                      const int32_t* __restrict__ flag,
                                                                      const int32_t* __restrict__ offs, uint64_t n_nodes,
                                                                      uint64_t seed, int64_t k_global,
                                                                      const int64_t* __restrict__ pos_type, int64_t n_pos,
                                                                      int64_t* __restrict__ out_head, int64_t* __restrict__ out_tail,
                                                                      int64_t* __restrict__ out_type, int32_t* __restrict__ count) {
  __shared__ Feistel sigma;
  __shared__ NegBlock blk;
  __shared__ int bi;
  if (threadIdx.x == 0) {
    bi = neg_block_of(blocks, n_blocks, (int64_t)blockIdx.x * 256);
    blk = blocks[bi];
    sigma = make_feistel((uint64_t)k_global, hs_mix(seed ^ 0x2545F4914F6CDD1Dull));
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t j = i - blk.cand0;
  const int64_t base = offs[blk.cand0];
  const int64_t last = blk.cand0 + blk.n_cand - 1;
  const int64_t survivors = (int64_t)offs[last] + flag[last] - base;
  const int64_t have = survivors < blk.k ? survivors : blk.k;
  auto put = [&](int64_t p, int64_t r, int64_t c) {
    out_head[blk.out0 + p] = r + blk.row_shift;
    out_tail[blk.out0 + p] = c + blk.col_shift;
    if (out_type) out_type[blk.out0 + p] = pos_type[sigma((uint64_t)(blk.gpos0 + p)) % (uint64_t)n_pos];
  };
  if (j < blk.n_cand && flag[i]) {
    const int64_t rank = (int64_t)offs[i] - base;
    if (rank < blk.k) {                    // vector_to_edge_index: re-open the diagonal
      const int64_t r = (int64_t)(keys[i] / (n_nodes - 1));
      int64_t c = (int64_t)(keys[i] % (n_nodes - 1));
      c += (r <= c ? 1 : 0);
      put(rank, r, c);
    }
  }
  if (j >= have && j < blk.k) {            // a shortfall (never, in practice: the candidate count carries six sigma; it is
    const int64_t r = blk.row_lo + j % blk.n_rows;     // reported through count): cells of the block all the same, so that
    int64_t c = blk.col_lo + (j + 1) % blk.n_cols;     // the incidence lists built over the block layout stay valid
    if (r == c) c = blk.col_lo + (j + 2) % blk.n_cols;
    put(j, r, c);
  }
  if (j == 0) count[bi] = (int32_t)have;
}

}  // namespace bkg

using namespace bkg;

static bool pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

extern "C" int bkg_hashset_insert_offdiag(const int64_t* head, const int64_t* tail, int64_t n_nodes, int64_t n,
                                          uint64_t* table, int64_t capacity, void* stream) {
  BKG_ARG(pow2(capacity) && capacity >= 2 * n && n_nodes > 1,
          "bkg_hashset_insert_offdiag: capacity must be a power of two >= 2 n, n_nodes > 1");
  if (n == 0) return 0;
  hipLaunchKernelGGL(hashset_insert_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, head, tail, n_nodes,
                     n, reinterpret_cast<unsigned long long*>(table), (uint64_t)capacity - 1);
  return check_launch("bkg_hashset_insert_offdiag");
}

extern "C" int64_t bkg_negative_sample_workspace_bytes(int64_t n_candidates) {
  size_t bytes = 0;
  rocprim::exclusive_scan(nullptr, bytes, (int32_t*)nullptr, (int32_t*)nullptr, 0, (size_t)n_candidates,
                          rocprim::plus<int32_t>());
  return (int64_t)bytes + 256;
}

extern "C" int bkg_negative_sample(int64_t n_nodes, int64_t first_candidate, int64_t n_candidates, uint64_t seed,
                                   const uint64_t* pos_table, int64_t pos_capacity, uint64_t* keys_tmp,
                                   int32_t* flag_tmp, int32_t* offs_tmp, int64_t* out_head, int64_t* out_tail,
                                   int64_t out_offset, int64_t max_out, int32_t* count, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
  BKG_ARG(n_nodes > 1 && n_nodes < (1ll << 31) && n_candidates >= 0 && n_candidates < (1ll << 31) && first_candidate >= 0,
          "bkg_negative_sample: bad sizes");
  BKG_ARG(first_candidate + n_candidates <= n_nodes * (n_nodes - 1), "bkg_negative_sample: more candidates than cells");
  BKG_ARG(pow2(pos_capacity), "bkg_negative_sample: table capacity must be a power of two");
  hipStream_t s = (hipStream_t)stream;
  if (n_candidates == 0) {
    fill_u32(count, 0u, 1, s);
    return check_launch("bkg_negative_sample(empty)");
  }
  const Feistel pi = make_feistel((uint64_t)(n_nodes * (n_nodes - 1)), seed);
  hipLaunchKernelGGL(negative_candidates_kernel, dim3(ceil_div(n_candidates, 256)), dim3(256), 0, s, pi, first_candidate,
                     n_candidates, reinterpret_cast<const unsigned long long*>(pos_table), (uint64_t)pos_capacity - 1,
                     keys_tmp, flag_tmp);
  size_t bytes = (size_t)workspace_bytes;
  hipError_t e = rocprim::exclusive_scan(workspace, bytes, flag_tmp, offs_tmp, 0, (size_t)n_candidates,
                                         rocprim::plus<int32_t>(), s);
  if (e != hipSuccess) { set_error("bkg_negative_sample: scan", e); return (int)e; }
  hipLaunchKernelGGL(negative_scatter_kernel, dim3(ceil_div(n_candidates, 256)), dim3(256), 0, s, keys_tmp, flag_tmp,
                     offs_tmp, n_candidates, (uint64_t)n_nodes, out_offset, max_out, out_head, out_tail, count);
  return check_launch("bkg_negative_sample");
}

extern "C" int64_t bkg_negative_blocks_struct_bytes(void) { return (int64_t)sizeof(NegBlock); }

// Block-structured negative sampling (see NegBlock): `blocks` is a DEVICE array of n_blocks descriptors (13 int64 each:
// row_lo, n_rows, col_lo, n_cols, population, cand0, n_cand, out0, k, gpos0, id, row_shift, col_shift; cand0 ascending
// multiples of 256; the shifts turn node ids into slots of a rank's gathered table on the way out),
// n_cand_total the padded candidate count (a multiple of 256).  out_head / out_tail / out_type: sum of k entries;
// pos_type: the n_pos positives' relation ids (out_type nullable); count[n_blocks]: survivors kept per block (== k unless
// a block fell short).
extern "C" int bkg_negative_sample_blocks(const void* blocks, int32_t n_blocks, int64_t n_cand_total, int64_t n_nodes,
                                          uint64_t seed, int64_t k_global, const uint64_t* pos_table, int64_t pos_capacity,
                                          const int64_t* pos_type, int64_t n_pos, uint64_t* keys_tmp, int32_t* flag_tmp,
                                          int32_t* offs_tmp, int64_t* out_head, int64_t* out_tail, int64_t* out_type,
                                          int32_t* count, void* workspace, int64_t workspace_bytes, void* stream) {
  BKG_ARG(n_blocks >= 0 && n_blocks <= 4096 && n_cand_total >= 0 && n_cand_total % 256 == 0 && n_cand_total < (1ll << 31) &&
              n_nodes > 1 && n_nodes < (1ll << 31) && k_global > 0, "bkg_negative_sample_blocks: bad sizes");
  BKG_ARG(pow2(pos_capacity) && (!out_type || (pos_type && n_pos > 0)), "bkg_negative_sample_blocks: bad tables");
  if (n_blocks == 0 || n_cand_total == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const NegBlock* b = reinterpret_cast<const NegBlock*>(blocks);
  hipLaunchKernelGGL(negative_candidates_blocks_kernel, dim3(n_cand_total / 256), dim3(256), 0, s, b, (int)n_blocks, seed,
                     rng_epoch_ptr(), (uint64_t)n_nodes, reinterpret_cast<const unsigned long long*>(pos_table), (uint64_t)pos_capacity - 1,
                     keys_tmp, flag_tmp);
  size_t bytes = (size_t)workspace_bytes;
  hipError_t e = rocprim::exclusive_scan(workspace, bytes, flag_tmp, offs_tmp, 0, (size_t)n_cand_total,
                                         rocprim::plus<int32_t>(), s);
  if (e != hipSuccess) { set_error("bkg_negative_sample_blocks: scan", e); return (int)e; }
  hipLaunchKernelGGL(negative_scatter_blocks_kernel, dim3(n_cand_total / 256), dim3(256), 0, s, b, (int)n_blocks, keys_tmp,
                     flag_tmp, offs_tmp, (uint64_t)n_nodes, seed, k_global, pos_type, n_pos, out_head, out_tail, out_type,
                     count);
  return check_launch("bkg_negative_sample_blocks");
}

extern "C" int bkg_permute_gather_i64(const int64_t* src, int64_t n, uint64_t seed, int64_t* out, void* stream) {
  BKG_ARG(n >= 0, "bkg_permute_gather_i64: bad n");
  if (n == 0) return 0;
  hipLaunchKernelGGL(permute_gather_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, src, n,
                     make_feistel32((uint64_t)n, seed), out);
  return check_launch("bkg_permute_gather_i64");
}

extern "C" int bkg_permute_gather_repeat_i64(const int64_t* src, int64_t n_src, int64_t repeat, uint64_t seed, int64_t* out,
                                             void* stream) {
  BKG_ARG(n_src >= 0 && repeat >= 0, "bkg_permute_gather_repeat_i64: bad sizes");
  const int64_t n = n_src * repeat;
  if (n == 0) return 0;
  hipLaunchKernelGGL(permute_gather_repeat_kernel<int64_t>, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, src, n_src,
                     n, make_feistel32((uint64_t)n, seed), out);
  return check_launch("bkg_permute_gather_repeat_i64");
}
// The same from a one-byte image of the ids (values < 256; bkg_narrow_i64_u8 makes it, once per id tensor): the random reads
// of a 4 M-entry source then stay in the L2 (4 MB instead of 32), which is what k = 10 repeats spend their time on.
extern "C" int bkg_narrow_i64_u8(const int64_t* src, int64_t n, uint8_t* out, void* stream) {
  BKG_ARG(n >= 0, "bkg_narrow_i64_u8: bad n");
  if (n == 0) return 0;
  hipLaunchKernelGGL(narrow_i64_u8_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, src, n, out);
  return check_launch("bkg_narrow_i64_u8");
}
extern "C" int bkg_permute_gather_repeat_u8(const uint8_t* src, int64_t n_src, int64_t repeat, uint64_t seed, int64_t* out,
                                            void* stream) {
  BKG_ARG(n_src >= 0 && repeat >= 0, "bkg_permute_gather_repeat_u8: bad sizes");
  const int64_t n = n_src * repeat;
  if (n == 0) return 0;
  hipLaunchKernelGGL(permute_gather_repeat_kernel<uint8_t>, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, src, n_src,
                     n, make_feistel32((uint64_t)n, seed), out);
  return check_launch("bkg_permute_gather_repeat_u8");
}

extern "C" int bkg_randperm_dev(const int64_t* n_dev, int64_t cap, uint64_t seed, const uint32_t* epoch, int64_t* out,
                                void* stream) {
  BKG_ARG(cap >= 0 && n_dev, "bkg_randperm_dev: bad arguments");
  if (cap == 0) return 0;
  hipLaunchKernelGGL(randperm_dev_kernel, dim3(ceil_div(cap, 256)), dim3(256), 0, (hipStream_t)stream, n_dev, cap, seed,
                     epoch ? epoch : rng_epoch_ptr(), out);      // (no counter given: the library-wide one of bkg_set_rng_epoch, if any)
  return check_launch("bkg_randperm_dev");
}

extern "C" int64_t bkg_negative_sample_dev_candidates(int64_t max_negatives) {
  return (int64_t)(1.1 * (double)max_negatives / 0.9) + 64;      // the callers guarantee prob > 0.9
}

// Device-extent form of bkg_hashset_insert_offdiag + bkg_negative_sample + bkg_permute_gather_i64 for one padded
// mini-batch: head / tail / type hold `rows` entries of which the first sizes[1] are edges.  table: int64[capacity]
// scratch (re-filled here); keys / flag / offs: scratch of bkg_negative_sample_dev_candidates(out_cap) entries.
extern "C" int bkg_negative_sample_dev(const int64_t* sizes, int64_t neg_ratio, const int64_t* head, const int64_t* tail,
                                       const int64_t* type, int64_t rows, uint64_t seed, const uint32_t* epoch,
                                       uint64_t* table, int64_t capacity, uint64_t* keys_tmp, int32_t* flag_tmp,
                                       int32_t* offs_tmp, int64_t out_cap, int64_t type_pad, int64_t* out_head,
                                       int64_t* out_tail, int64_t* out_type, int64_t* count, void* workspace,
                                       int64_t workspace_bytes, void* stream) {
  BKG_ARG(rows > 0 && neg_ratio > 0 && out_cap >= neg_ratio * rows && pow2(capacity) && capacity >= 2 * rows,
          "bkg_negative_sample_dev: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  const int64_t cap = bkg_negative_sample_dev_candidates(out_cap);
  fill_u32(table, 0xFFFFFFFFu, capacity * 2, s);
  hipError_t e = hipSuccess;
  hipLaunchKernelGGL(hashset_insert_dev_kernel, dim3(ceil_div(rows, 256)), dim3(256), 0, s, head, tail, sizes, rows,
                     reinterpret_cast<unsigned long long*>(table), (uint64_t)capacity - 1);
  hipLaunchKernelGGL(negative_candidates_dev_kernel, dim3(ceil_div(cap, 256)), dim3(256), 0, s, sizes, neg_ratio, seed,
                     epoch, cap, reinterpret_cast<const unsigned long long*>(table), (uint64_t)capacity - 1, keys_tmp,
                     flag_tmp);
  size_t bytes = (size_t)workspace_bytes;
  e = rocprim::exclusive_scan(workspace, bytes, flag_tmp, offs_tmp, 0, (size_t)cap, rocprim::plus<int32_t>(), s);
  if (e != hipSuccess) { set_error("bkg_negative_sample_dev: scan", e); return (int)e; }
  hipLaunchKernelGGL(negative_scatter_dev_kernel, dim3(ceil_div(cap, 256)), dim3(256), 0, s, keys_tmp, flag_tmp, offs_tmp, cap,
                     sizes, neg_ratio, out_cap, out_head, out_tail, count);
  if (out_type)
    hipLaunchKernelGGL(permute_gather_dev_kernel, dim3(ceil_div(out_cap, 256)), dim3(256), 0, s, type, sizes, neg_ratio, seed,
                       epoch, count, type_pad, out_cap, out_type);
  return check_launch("bkg_negative_sample_dev");
}

extern "C" int bkg_padded_loss_inputs(const int64_t* sizes, const int64_t* count, const int64_t* src, const int64_t* dst,
                                      const int64_t* type, int64_t rows, int64_t node_cap, int64_t pad_rel, int64_t* out_src,
                                      int64_t* out_dst, int64_t* out_type, float* inv_total, void* stream) {
  BKG_ARG(rows > 0 && node_cap > 1, "bkg_padded_loss_inputs: bad sizes");
  hipLaunchKernelGGL(padded_loss_inputs_kernel, dim3(ceil_div(rows, 256)), dim3(256), 0, (hipStream_t)stream, sizes, count,
                     src, dst, type, rows, node_cap, pad_rel, out_src, out_dst, out_type, inv_total);
  return check_launch("bkg_padded_loss_inputs");
}

extern "C" int bkg_padded_gcl_batch(const int64_t* sizes, const int64_t* n_id, const int64_t* src, const int64_t* dst,
                                    int64_t n_cap, int64_t e_b, int64_t pad_row, int64_t last_node, int64_t* out_n_id,
                                    int64_t* out_src, int64_t* out_dst, uint8_t* real, void* stream) {
  BKG_ARG(n_cap >= 0 && e_b >= 0 && sizes, "bkg_padded_gcl_batch: bad sizes");
  const int64_t span = n_cap > e_b ? n_cap : e_b;
  if (span == 0) return 0;
  hipLaunchKernelGGL(padded_gcl_batch_kernel, dim3(ceil_div(span, 256)), dim3(256), 0, (hipStream_t)stream, sizes, n_id, src, dst,
                     n_cap, e_b, pad_row, last_node, out_n_id, out_src, out_dst, real);
  return check_launch("bkg_padded_gcl_batch");
}
