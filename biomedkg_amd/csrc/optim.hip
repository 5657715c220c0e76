// Global-norm gradient clipping + Adam over a LIST of parameter tensors as two launches.
//
// The reference's step tail -- Lightning's gradient_clip_val = 1.0 (train_gcl.py:63-68, train_kge.py:44-50:
// torch.nn.utils.clip_grad_norm_) followed by torch.optim.Adam.step() (gcl_module.py:78-87, kge_module.py:208-217) -- is,
// on the GPU, a chain of small launches: per-tensor norms, a stack + norm, a clamp, a multiply per tensor, then the
// multi-tensor Adam (torch: FusedAdam + norm + mul + addcmul, ~3 % of the bench step and 12 of the ~150 nodes of a
// replayed mini-batch step).  Here:
//
//   launch 1  partial[b] = sum of g^2 over block b's 4096-element chunk (chunks of all tensors in one grid)
//   launch 2  every block re-adds the partials in index order (same total on every block, bitwise repeatable),
//             c = min(1, max_norm / (sqrt(total) + 1e-6));  g' = c g;  m = b1 m + (1 - b1) g';  v = b2 v + (1 - b2) g'^2;
//             p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)       [torch.optim.Adam, no amsgrad / decay]
//
// Tensor pointers travel BY VALUE in the kernel arguments (<= 48 tensors, 1.7 KB): gradients are re-allocated every step
// (zero_grad(set_to_none=True)), a device-side pointer table would need a host-to-device copy per step.  The step count
// and the learning rate are host scalars, or device scalars for captured (hipGraph) steps: `step_dev` is incremented by
// the kernel, `lr_dev` is what the LR scheduler fills between replays.
#include "common.h"

namespace bkg {

constexpr int OPT_MAX_TENSORS = 48;
constexpr int OPT_CHUNK = 4096;       // elements per block

struct OptTensors {
  float* p[OPT_MAX_TENSORS];
  const float* g[OPT_MAX_TENSORS];
  float* m[OPT_MAX_TENSORS];
  float* v[OPT_MAX_TENSORS];
  int64_t n[OPT_MAX_TENSORS];
  int32_t first_chunk[OPT_MAX_TENSORS + 1];   // chunk ranges per tensor (prefix sums)
  int32_t count;
};

// bf16 images of the weights that the bf16-mode products read (bkg_weight_to_bf16: [batch][rows][cols] as stored and its
// transpose [batch][cols][rows]), refreshed by the update itself: a weight changes exactly here, so the twelve conversion
// launches at the top of the next step (one per weight) disappear.  w16[k] == nullptr: tensor k has no image.
struct OptImages {
  __bf16* w16[OPT_MAX_TENSORS];
  __bf16* w16t[OPT_MAX_TENSORS];
  int32_t rows[OPT_MAX_TENSORS], cols[OPT_MAX_TENSORS];
};

__device__ __forceinline__ int opt_tensor_of(const OptTensors& t, int chunk) {
  int lo = 0, hi = t.count;                   // first_chunk[lo] <= chunk < first_chunk[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (t.first_chunk[mid] <= chunk) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const float out = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return out;
}

__global__ __launch_bounds__(256) void grad_sqnorm_kernel(OptTensors t, float* __restrict__ partial) {
  __shared__ float sh[4];
  const int chunk = blockIdx.x;
  const int k = opt_tensor_of(t, chunk);
  const int64_t base = (int64_t)(chunk - t.first_chunk[k]) * OPT_CHUNK;
  const int64_t n = t.n[k];
  const float* g = t.g[k];
  float acc = 0.f;
  if (((uintptr_t)g & 15) == 0) {
#pragma unroll
    for (int it = 0; it < OPT_CHUNK / 1024; ++it) {
      const int64_t i = base + (int64_t)it * 1024 + threadIdx.x * 4;
      if (i + 3 < n) {
        const float4 x = *reinterpret_cast<const float4*>(g + i);
        acc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
      } else {
        for (int64_t j = i; j < n && j < i + 4; ++j) acc += g[j] * g[j];
      }
    }
  } else {
    for (int64_t i = base + threadIdx.x; i < n && i < base + OPT_CHUNK; i += 256) acc += g[i] * g[i];
  }
  const float s = block_sum_256(acc, sh);
  if (threadIdx.x == 0) partial[chunk] = s;
}

// One element of the update (torch.optim.Adam, no amsgrad / decay) -- the same operations in the same order whichever
// path below loads it.
__device__ __forceinline__ float adam_element(float g, float clip, float& m, float& v, float p, float beta1, float beta2,
                                              float step_size, float inv_sqrt_bc2, float eps) {
  const float gi = g * clip;
  m = beta1 * m + (1.0f - beta1) * gi;
  v = beta2 * v + (1.0f - beta2) * gi * gi;
  return p - step_size * m / (sqrtf(v) * inv_sqrt_bc2 + eps);
}

constexpr int OPT_TILE_PAD = 2;          // bf16 elements between the rows of a chunk staged for its transposed image

template <bool IMAGES>
__global__ __launch_bounds__(256) void clip_adam_kernel(OptTensors t, OptImages im, const float* __restrict__ partial,
                                                        int n_partial, float max_norm, float lr,
                                                        const float* __restrict__ lr_dev, float beta1, float beta2,
                                                        float eps, float step, float* __restrict__ step_dev,
                                                        float* __restrict__ norm_out) {
  __shared__ float sh[4];
  __shared__ __bf16 tile[IMAGES ? OPT_CHUNK + 64 * OPT_TILE_PAD : 1];
  // the total in index order: every block computes the same sum (n_partial <= a few thousand floats, L2-resident)
  float acc = 0.f;
  for (int i = threadIdx.x; i < n_partial; i += 256) acc += partial[i];
  const float total = block_sum_256(acc, sh);
  const float norm = sqrtf(total);
  const float clip = max_norm > 0.f ? fminf(max_norm / (norm + 1e-6f), 1.0f) : 1.0f;
  float tstep = step;
  if (step_dev) tstep = step_dev[0] + 1.0f;     // (the counter itself is bumped by block 0 below, after everybody's read...)
  const float rate = lr_dev ? lr_dev[0] : lr;
  const float bc1 = 1.0f - powf(beta1, tstep), bc2 = 1.0f - powf(beta2, tstep);
  const float step_size = rate / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
  const int chunk = blockIdx.x;
  const int k = opt_tensor_of(t, chunk);
  const int64_t base = (int64_t)(chunk - t.first_chunk[k]) * OPT_CHUNK;
  const int64_t n = t.n[k];
  float* __restrict__ p = t.p[k];
  const float* __restrict__ g = t.g[k];
  float* __restrict__ m = t.m[k];
  float* __restrict__ v = t.v[k];
  __bf16* __restrict__ w16 = IMAGES ? im.w16[k] : nullptr;
  __bf16* __restrict__ w16t = IMAGES ? im.w16t[k] : nullptr;
  const int rows = IMAGES ? im.rows[k] : 1, cols = IMAGES ? im.cols[k] : 1;
  if (chunk == 0 && threadIdx.x == 0 && norm_out) norm_out[0] = norm;
  // A whole chunk of sixteen-byte aligned tensors: four elements per thread and pass (the R-GCN weights are 97 % of the
  // bench step's parameters).  With images, when the chunk is a band of whole rows of ONE matrix (cols divides the chunk,
  // the matrix a whole number of chunks): the band is staged in LDS and leaves for the transposed image with 4096 / cols
  // consecutive rows per column -- 32 bytes per 16 lanes at cols = 256 -- instead of one 2-byte store per cache line
  // (8 M scattered stores for the bench's R-GCN weights: the update ran at 1.1 TB/s, 228 us).
  const bool vec = base + OPT_CHUNK <= n && ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
  if (vec) {
    const bool band = IMAGES && w16 && cols >= 64 && OPT_CHUNK % cols == 0 && ((int64_t)rows * cols) % OPT_CHUNK == 0 &&
                      (((uintptr_t)w16 & 7) == 0);
    const int band_rows = band ? OPT_CHUNK / cols : 1;
#pragma unroll
    for (int it = 0; it < OPT_CHUNK / 1024; ++it) {
      const int local = it * 1024 + threadIdx.x * 4;
      const int64_t i = base + local;
      const float4 g4 = *reinterpret_cast<const float4*>(g + i);
      float4 m4 = *reinterpret_cast<const float4*>(m + i);
      float4 v4 = *reinterpret_cast<const float4*>(v + i);
      float4 p4 = *reinterpret_cast<const float4*>(p + i);
      p4.x = adam_element(g4.x, clip, m4.x, v4.x, p4.x, beta1, beta2, step_size, inv_sqrt_bc2, eps);
      p4.y = adam_element(g4.y, clip, m4.y, v4.y, p4.y, beta1, beta2, step_size, inv_sqrt_bc2, eps);
      p4.z = adam_element(g4.z, clip, m4.z, v4.z, p4.z, beta1, beta2, step_size, inv_sqrt_bc2, eps);
      p4.w = adam_element(g4.w, clip, m4.w, v4.w, p4.w, beta1, beta2, step_size, inv_sqrt_bc2, eps);
      *reinterpret_cast<float4*>(m + i) = m4;
      *reinterpret_cast<float4*>(v + i) = v4;
      *reinterpret_cast<float4*>(p + i) = p4;
      if (IMAGES && w16) {
        const __bf16 h[4] = {(__bf16)p4.x, (__bf16)p4.y, (__bf16)p4.z, (__bf16)p4.w};
        if (band) {
          *reinterpret_cast<uint2*>(w16 + i) = *reinterpret_cast<const uint2*>(h);
          const int r = local / cols, c = local - r * cols;          // (cols % 4 == 0: the four share a row)
#pragma unroll
          for (int e = 0; e < 4; ++e) tile[r * (cols + OPT_TILE_PAD) + c + e] = h[e];
        } else {
          const int64_t per = (int64_t)rows * cols;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int64_t ie = i + e, b = ie / per, rem = ie - b * per;
            const int r = (int)(rem / cols), c = (int)(rem - (int64_t)r * cols);
            w16[ie] = h[e];
            w16t[b * per + (int64_t)c * rows + r] = h[e];
          }
        }
      }
    }
    if (band) {
      __syncthreads();
      const int64_t per = (int64_t)rows * cols, b = base / per;
      const int r0 = (int)((base - b * per) / cols);
      __bf16* __restrict__ out = w16t + b * per + r0;
      for (int j = threadIdx.x; j < OPT_CHUNK; j += 256) {
        const int c = j / band_rows, dr = j - c * band_rows;
        out[(int64_t)c * rows + dr] = tile[dr * (cols + OPT_TILE_PAD) + c];
      }
    }
    return;
  }
  for (int64_t i = base + threadIdx.x; i < n && i < base + OPT_CHUNK; i += 256) {
    float mi = m[i], vi = v[i];
    const float pi = adam_element(g[i], clip, mi, vi, p[i], beta1, beta2, step_size, inv_sqrt_bc2, eps);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
    if (IMAGES && w16) {
      const __bf16 h = (__bf16)pi;
      w16[i] = h;
      const int64_t per = (int64_t)rows * cols, b = i / per, rem = i - b * per;
      const int r = (int)(rem / cols), c = (int)(rem - (int64_t)r * cols);
      w16t[b * per + (int64_t)c * rows + r] = h;           // (2-byte scattered stores into matrices that live in the L2)
    }
  }
}

// the device step counter is bumped by its own one-thread launch AFTER the update (a block of the update kernel doing it
// would race with the other blocks' reads of the old value)
__global__ void bump_step_kernel(float* __restrict__ step_dev) { step_dev[0] += 1.0f; }

}  // namespace bkg

using namespace bkg;

extern "C" int64_t bkg_clip_adam_max_tensors() { return OPT_MAX_TENSORS; }
extern "C" int64_t bkg_clip_adam_chunks(const int64_t* numel, int32_t count) {
  int64_t c = 0;
  for (int i = 0; i < count; ++i) c += (numel[i] + OPT_CHUNK - 1) / OPT_CHUNK;
  return c;
}

static int clip_adam_impl(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                          float* const* exp_avg_sq, const int64_t* numel, void* const* w16, void* const* w16t,
                          const int32_t* rows, const int32_t* cols, float max_norm, float lr, const float* lr_dev,
                          float beta1, float beta2, float eps, float step, float* step_dev, float* partial,
                          int64_t n_partial, float* norm_out, void* stream) {
  BKG_ARG(count >= 0 && count <= OPT_MAX_TENSORS, "bkg_clip_adam_f32: at most 48 tensors per call");
  BKG_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps > 0.f, "bkg_clip_adam_f32: bad hyper-parameters");
  if (count == 0) return 0;
  OptTensors t;
  int64_t chunks = 0;
  t.count = count;
  for (int i = 0; i < count; ++i) {
    BKG_ARG(numel[i] >= 0 && params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i], "bkg_clip_adam_f32: null tensor");
    t.p[i] = params[i]; t.g[i] = grads[i]; t.m[i] = exp_avg[i]; t.v[i] = exp_avg_sq[i]; t.n[i] = numel[i];
    t.first_chunk[i] = (int32_t)chunks;
    chunks += (numel[i] + OPT_CHUNK - 1) / OPT_CHUNK;
  }
  t.first_chunk[count] = (int32_t)chunks;
  BKG_ARG(chunks < (1 << 30) && n_partial >= chunks && partial, "bkg_clip_adam_f32: partial needs bkg_clip_adam_chunks floats");
  if (chunks == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3((unsigned)chunks), dim3(256), 0, s, t, partial);
  OptImages im;
  bool any = false;
  for (int i = 0; i < count; ++i) {
    im.w16[i] = w16 ? reinterpret_cast<__bf16*>(w16[i]) : nullptr;
    im.w16t[i] = w16t ? reinterpret_cast<__bf16*>(w16t[i]) : nullptr;
    im.rows[i] = rows ? rows[i] : 1;
    im.cols[i] = cols ? cols[i] : 1;
    if (im.w16[i]) {
      BKG_ARG(im.w16t[i] && im.rows[i] > 0 && im.cols[i] > 0 && numel[i] % ((int64_t)im.rows[i] * im.cols[i]) == 0,
              "bkg_clip_adam_images_f32: an image needs both orientations and rows * cols dividing the tensor");
      any = true;
    }
  }
  if (any)
    hipLaunchKernelGGL(clip_adam_kernel<true>, dim3((unsigned)chunks), dim3(256), 0, s, t, im, partial, (int)chunks, max_norm,
                       lr, lr_dev, beta1, beta2, eps, step, step_dev, norm_out);
  else
    hipLaunchKernelGGL(clip_adam_kernel<false>, dim3((unsigned)chunks), dim3(256), 0, s, t, im, partial, (int)chunks, max_norm,
                       lr, lr_dev, beta1, beta2, eps, step, step_dev, norm_out);
  if (step_dev) hipLaunchKernelGGL(bump_step_kernel, dim3(1), dim3(1), 0, s, step_dev);
  return check_launch("bkg_clip_adam_f32");
}

extern "C" int bkg_clip_adam_f32(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                                 float* const* exp_avg_sq, const int64_t* numel, float max_norm, float lr,
                                 const float* lr_dev, float beta1, float beta2, float eps, float step, float* step_dev,
                                 float* partial, int64_t n_partial, float* norm_out, void* stream) {
  return clip_adam_impl(count, params, grads, exp_avg, exp_avg_sq, numel, nullptr, nullptr, nullptr, nullptr, max_norm, lr,
                        lr_dev, beta1, beta2, eps, step, step_dev, partial, n_partial, norm_out, stream);
}

// bkg_clip_adam_f32 that also refreshes the bf16 images of the weights it moves (bkg_weight_to_bf16's two orientations):
// w16[i] / w16t[i] (nullable per tensor) = the [batch][rows][cols] and [batch][cols][rows] bf16 images of tensor i.
extern "C" int bkg_clip_adam_images_f32(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                                        float* const* exp_avg_sq, const int64_t* numel, void* const* w16, void* const* w16t,
                                        const int32_t* rows, const int32_t* cols, float max_norm, float lr,
                                        const float* lr_dev, float beta1, float beta2, float eps, float step,
                                        float* step_dev, float* partial, int64_t n_partial, float* norm_out, void* stream) {
  return clip_adam_impl(count, params, grads, exp_avg, exp_avg_sq, numel, w16, w16t, rows, cols, max_norm, lr, lr_dev, beta1,
                        beta2, eps, step, step_dev, partial, n_partial, norm_out, stream);
}
