// Ranking evaluation for the TransE decoder (BASELINE configs[4] quotes MRR for the DPI fine-tune):
// score(h, r, t) = -|| z^_h + r - z^_t ||_1 with z^ the L1-normalised rows (reference decoder.py:31-38).
// Tail prediction ranks candidates j by -|| q - z^_j ||_1 with q = z^_h + r; head prediction by the same
// distance with q = z^_t - r: both are "L1 distance of a query row to every row of the normalised table",
// which is VALU work, not a GEMM.  One wave owns one candidate row at a time (coalesced 16 B/lane load) and
// scores it against QB query rows held in registers; a query's true score is produced by the SAME device
// function on its target row, so ties compare bit-identically.  Build-defined evaluation (the reference has no
// ranking metric); checker: oracle/ranking.py.
#include "common.h"

namespace bkg {

constexpr int RL_QB = 8;   // query rows per block, kept in registers by every wave

// -sum_k |q_k - z_k| over a d-wide row spread over the 64 lanes (VPL float4 per lane), all lanes get the result
template <int VPL>
__device__ __forceinline__ float rl_neg_l1(const float4 (&q)[VPL], const float4 (&z)[VPL]) {
  float s = 0.f;
#pragma unroll
  for (int v = 0; v < VPL; ++v)
    s += fabsf(q[v].x - z[v].x) + fabsf(q[v].y - z[v].y) + fabsf(q[v].z - z[v].z) + fabsf(q[v].w - z[v].w);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  return -s;
}

template <int VPL>
__device__ __forceinline__ void rl_load(const float* __restrict__ row, int d, int lane, float4 (&out)[VPL]) {
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int c = 4 * (lane + 64 * v);
    out[v] = c < d ? *reinterpret_cast<const float4*>(row + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// true[i] = -||q_i - zn[target_i]||_1
template <int VPL>
__global__ __launch_bounds__(256) void rank_l1_true_kernel(const float* __restrict__ q, int64_t ldq,
                                                           const float* __restrict__ zn, int64_t ldz,
                                                           const int64_t* __restrict__ target, int64_t n_q, int d,
                                                           float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_q) return;
  float4 a[VPL], b[VPL];
  rl_load<VPL>(q + i * ldq, d, lane, a);
  rl_load<VPL>(zn + target[i] * ldz, d, lane, b);
  const float s = rl_neg_l1<VPL>(a, b);
  if (lane == 0) out[i] = s;
}

// block = RL_QB queries x a slice of the candidates; each wave walks candidates j = wave, wave + 4, ...
template <int VPL>
__global__ __launch_bounds__(256) void rank_l1_count_kernel(const float* __restrict__ q, int64_t ldq, int n_q,
                                                            const float* __restrict__ zn, int64_t ldz, int n_nodes,
                                                            int d, const float* __restrict__ true_score,
                                                            const int64_t* __restrict__ target, int cand_per_split,
                                                            int n_q_blocks, int32_t* __restrict__ gt,
                                                            int32_t* __restrict__ eq) {
  __shared__ int s_gt[4][RL_QB], s_eq[4][RL_QB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qb = blockIdx.x % n_q_blocks, split = blockIdx.x / n_q_blocks;
  const int i0 = qb * RL_QB;
  float4 qr[RL_QB][VPL];
  float t[RL_QB];
  int tg[RL_QB], cg[RL_QB], ce[RL_QB];
#pragma unroll
  for (int u = 0; u < RL_QB; ++u) {
    const int i = min(i0 + u, n_q - 1);
    rl_load<VPL>(q + (int64_t)i * ldq, d, lane, qr[u]);
    t[u] = true_score[i];
    tg[u] = (i0 + u < n_q) ? (int)target[i] : -2;
    cg[u] = ce[u] = 0;
  }
  const int j_beg = split * cand_per_split, j_end = min(j_beg + cand_per_split, n_nodes);
  for (int j = j_beg + wave; j < j_end; j += 4) {
    float4 z[VPL];
    rl_load<VPL>(zn + (int64_t)j * ldz, d, lane, z);
#pragma unroll
    for (int u = 0; u < RL_QB; ++u) {
      const float s = rl_neg_l1<VPL>(qr[u], z);
      const bool valid = tg[u] != -2 && j != tg[u];
      cg[u] += (valid && s > t[u]) ? 1 : 0;
      ce[u] += (valid && s == t[u]) ? 1 : 0;
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int u = 0; u < RL_QB; ++u) { s_gt[wave][u] = cg[u]; s_eq[wave][u] = ce[u]; }
  }
  __syncthreads();
  if (threadIdx.x < RL_QB && i0 + threadIdx.x < n_q) {
    const int u = threadIdx.x;
    gt[(int64_t)split * n_q + i0 + u] = s_gt[0][u] + s_gt[1][u] + s_gt[2][u] + s_gt[3][u];
    eq[(int64_t)split * n_q + i0 + u] = s_eq[0][u] + s_eq[1][u] + s_eq[2][u] + s_eq[3][u];
  }
}

// filtered setting: one wave per query removes its known true candidates from the summed counts
template <int VPL>
__global__ __launch_bounds__(256) void rank_l1_filter_kernel(const float* __restrict__ q, int64_t ldq,
                                                             const float* __restrict__ zn, int64_t ldz, int d,
                                                             const float* __restrict__ true_score,
                                                             const int64_t* __restrict__ target,
                                                             const int64_t* __restrict__ qptr,
                                                             const int64_t* __restrict__ cand, int64_t n_q,
                                                             int32_t* __restrict__ gt, int32_t* __restrict__ eq) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n_q) return;
  float4 a[VPL], b[VPL];
  rl_load<VPL>(q + i * ldq, d, lane, a);
  const float t = true_score[i];
  int dg = 0, de = 0;
  for (int64_t c = qptr[i]; c < qptr[i + 1]; ++c) {
    const int64_t j = cand[c];
    if (j == target[i]) continue;
    rl_load<VPL>(zn + j * ldz, d, lane, b);
    const float s = rl_neg_l1<VPL>(a, b);
    dg += s > t;
    de += s == t;
  }
  if (lane == 0) { gt[i] -= dg; eq[i] -= de; }
}

#define RL_DISPATCH(D, ...)                                     \
  if ((D) <= 256) { constexpr int VPL = 1; __VA_ARGS__; }       \
  else if ((D) <= 512) { constexpr int VPL = 2; __VA_ARGS__; }  \
  else if ((D) <= 768) { constexpr int VPL = 3; __VA_ARGS__; }  \
  else { bkg::set_error_msg("rank_l1: embedding width must be <= 768"); return -1; }

static bool rl_ok(const void* p, int64_t ld, int d) { return ((uintptr_t)p % 16 == 0) && ld % 4 == 0 && d % 4 == 0; }

}  // namespace bkg

using namespace bkg;

extern "C" int bkg_rank_l1_true_f32(const float* q, int64_t ldq, const float* zn, int64_t ldz, const int64_t* target,
                                    int64_t n_q, int32_t d, float* out, void* stream) {
  BKG_ARG(rl_ok(q, ldq, d) && rl_ok(zn, ldz, d), "bkg_rank_l1_true_f32: rows must be 16-byte aligned, d % 4 == 0");
  if (n_q == 0) return 0;
  RL_DISPATCH(d, hipLaunchKernelGGL((rank_l1_true_kernel<VPL>), dim3(ceil_div(n_q, 4)), dim3(256), 0,
                                    (hipStream_t)stream, q, ldq, zn, ldz, target, n_q, d, out))
  return check_launch("bkg_rank_l1_true_f32");
}

extern "C" int bkg_rank_l1_count_f32(const float* q, int64_t ldq, int64_t n_q, const float* zn, int64_t ldz,
                                     int64_t n_nodes, int32_t d, const float* true_score, const int64_t* target,
                                     int32_t splits, int32_t* gt_partial, int32_t* eq_partial, void* stream) {
  BKG_ARG(rl_ok(q, ldq, d) && rl_ok(zn, ldz, d), "bkg_rank_l1_count_f32: rows must be 16-byte aligned, d % 4 == 0");
  BKG_ARG(n_q > 0 && n_nodes > 0 && splits > 0, "bkg_rank_l1_count_f32: bad sizes");
  const int n_qb = ceil_div(n_q, RL_QB);
  const int per = ceil_div(n_nodes, splits);
  RL_DISPATCH(d, hipLaunchKernelGGL((rank_l1_count_kernel<VPL>), dim3(n_qb * splits), dim3(256), 0, (hipStream_t)stream,
                                    q, ldq, (int)n_q, zn, ldz, (int)n_nodes, d, true_score, target, per, n_qb,
                                    gt_partial, eq_partial))
  return check_launch("bkg_rank_l1_count_f32");
}

extern "C" int bkg_rank_l1_filter_f32(const float* q, int64_t ldq, const float* zn, int64_t ldz, int32_t d,
                                      const float* true_score, const int64_t* target, const int64_t* qptr,
                                      const int64_t* cand, int64_t n_q, int32_t* gt, int32_t* eq, void* stream) {
  BKG_ARG(rl_ok(q, ldq, d) && rl_ok(zn, ldz, d), "bkg_rank_l1_filter_f32: rows must be 16-byte aligned, d % 4 == 0");
  if (n_q == 0) return 0;
  RL_DISPATCH(d, hipLaunchKernelGGL((rank_l1_filter_kernel<VPL>), dim3(ceil_div(n_q, 4)), dim3(256), 0,
                                    (hipStream_t)stream, q, ldq, zn, ldz, d, true_score, target, qptr, cand, n_q, gt, eq))
  return check_launch("bkg_rank_l1_filter_f32");
}
