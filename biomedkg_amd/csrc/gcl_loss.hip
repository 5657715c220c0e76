// GGD's loss (reference biomedkg/gcl_module.py:229-234): binary_cross_entropy_with_logits(cat(pos, neg), cat(1, 0)) -- the
// mean over 2 n logits of softplus(-pos) and softplus(neg) -- as one partial-sum launch + a one-block finish, and its
// gradient as one launch: torch's form is two cats, ones / zeros, the elementwise loss, a mean and five backward launches
// over 2 n floats (a replayed mini-batch step is bound by its node count).  Masked form (padded mini-batches,
// biomedkg_amd/sampled_gcl.py): only rows below n_dev[0] (a DEVICE scalar) count, and the mean divides by 2 n_dev[0].
// Sums run in a fixed order: bitwise repeatable.
#include "common.h"

namespace bkg {

__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

__global__ __launch_bounds__(256) void ggd_bce_partial_kernel(const float* __restrict__ pos, int64_t sp,
                                                              const float* __restrict__ neg, int64_t sn, int64_t n,
                                                              const int64_t* __restrict__ n_dev, float* __restrict__ partial) {
  __shared__ float sh[4];
  const int64_t live = n_dev ? (n_dev[0] < n ? n_dev[0] : n) : n;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < live; i += (int64_t)gridDim.x * 256)
    acc += softplus_f(-pos[i * sp]) + softplus_f(neg[i * sn]);
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void ggd_bce_finish_kernel(const float* __restrict__ partial, int n_partial, int64_t n,
                                                             const int64_t* __restrict__ n_dev, float* __restrict__ out) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n_partial; i += 256) acc += partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int64_t live = n_dev ? (n_dev[0] < n ? n_dev[0] : n) : n;
    out[0] = live > 0 ? ((sh[0] + sh[1]) + (sh[2] + sh[3])) / (2.0f * (float)live) : 0.f;
  }
}

// d loss / d pos = -sigmoid(-pos) / (2 n), d loss / d neg = sigmoid(neg) / (2 n), times the upstream scalar
__global__ __launch_bounds__(256) void ggd_bce_bwd_kernel(const float* __restrict__ pos, int64_t sp,
                                                          const float* __restrict__ neg, int64_t sn, int64_t n,
                                                          const int64_t* __restrict__ n_dev, const float* __restrict__ gout,
                                                          float* __restrict__ dpos, int64_t dsp, float* __restrict__ dneg,
                                                          int64_t dsn) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t live = n_dev ? (n_dev[0] < n ? n_dev[0] : n) : n;
  float gp = 0.f, gn = 0.f;
  if (i < live) {
    const float g = gout[0] / (2.0f * (float)live);
    gp = -g / (1.0f + expf(pos[i * sp]));
    gn = g / (1.0f + expf(-neg[i * sn]));
  }
  dpos[i * dsp] = gp;
  dneg[i * dsn] = gn;
}

constexpr int GGD_BCE_BLOCKS = 256;

}  // namespace bkg

using namespace bkg;

extern "C" int64_t bkg_ggd_bce_partials(void) { return GGD_BCE_BLOCKS; }

extern "C" int bkg_ggd_bce_f32(const float* pos, int64_t stride_pos, const float* neg, int64_t stride_neg, int64_t n,
                               const int64_t* n_dev, float* partial, float* out, void* stream) {
  BKG_ARG(n >= 0 && stride_pos > 0 && stride_neg > 0 && partial && out, "bkg_ggd_bce_f32: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(GGD_BCE_BLOCKS, ceil_div(n, 256)));
  hipLaunchKernelGGL(ggd_bce_partial_kernel, dim3(blocks), dim3(256), 0, s, pos, stride_pos, neg, stride_neg, n, n_dev, partial);
  hipLaunchKernelGGL(ggd_bce_finish_kernel, dim3(1), dim3(256), 0, s, partial, blocks, n, n_dev, out);
  return check_launch("bkg_ggd_bce_f32");
}

extern "C" int bkg_ggd_bce_bwd_f32(const float* pos, int64_t stride_pos, const float* neg, int64_t stride_neg, int64_t n,
                                   const int64_t* n_dev, const float* gout, float* dpos, int64_t stride_dpos, float* dneg,
                                   int64_t stride_dneg, void* stream) {
  BKG_ARG(n >= 0 && stride_pos > 0 && stride_neg > 0 && stride_dpos > 0 && stride_dneg > 0 && gout,
          "bkg_ggd_bce_bwd_f32: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(ggd_bce_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, pos, stride_pos, neg,
                     stride_neg, n, n_dev, gout, dpos, stride_dpos, dneg, stride_dneg);
  return check_launch("bkg_ggd_bce_bwd_f32");
}
