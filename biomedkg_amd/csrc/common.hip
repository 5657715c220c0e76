// Error reporting for the C-ABI (see include/biomedkg_hip.h).
#include "common.h"

#include <algorithm>
#include <cstdio>
#include <cstring>

namespace bkg {
static thread_local char g_err[512] = "";

void set_error(const char* what, hipError_t e) {
  snprintf(g_err, sizeof(g_err), "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
}
void set_error_msg(const char* what) { snprintf(g_err, sizeof(g_err), "%s", what); }

static DetWorkspace g_det = {nullptr, 0};
DetWorkspace det_workspace() { return g_det; }
static const uint32_t* g_rng_epoch = nullptr;
const uint32_t* rng_epoch_ptr() { return g_rng_epoch; }

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, int64_t ldc,
                                                            int64_t strideC, int M, int N, const int32_t* __restrict__ group_ptr,
                                                            int G, int kchunk, int n_chunks) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per = (int64_t)M * N;
  // the slabs of group g: chunks [first, first + cnt).  When a block lies inside one group (per % 256 == 0) one lane
  // walks group_ptr for all of them.
  __shared__ int s_first, s_cnt;
  int first = 0, cnt = n_chunks;
  const bool uniform = per % 256 == 0;
  if (group_ptr && uniform) {
    if (threadIdx.x == 0) {
      const int g0 = (int)(((int64_t)blockIdx.x * 256) / per);
      int f = 0;
      for (int q = 0; q < g0 && q < G; ++q) f += (group_ptr[q + 1] - group_ptr[q] + kchunk - 1) / kchunk;
      s_first = f;
      s_cnt = g0 < G ? (group_ptr[g0 + 1] - group_ptr[g0] + kchunk - 1) / kchunk : 0;
    }
    __syncthreads();
    first = s_first; cnt = s_cnt;
  }
  if (i >= per * G) return;
  const int g = (int)(i / per);
  const int64_t e = i % per;
  if (group_ptr && !uniform) {
    first = 0;
    for (int q = 0; q < g; ++q) first += (group_ptr[q + 1] - group_ptr[q] + kchunk - 1) / kchunk;
    cnt = (group_ptr[g + 1] - group_ptr[g] + kchunk - 1) / kchunk;
  }
  float acc = 0.f;
  const float* p = ws + (int64_t)first * per + e;
  int c = 0;
  for (; c + 4 <= cnt; c += 4) {      // four slabs in flight; still added in chunk order: bitwise repeatable
    const float v0 = p[(int64_t)c * per], v1 = p[(int64_t)(c + 1) * per], v2 = p[(int64_t)(c + 2) * per],
                v3 = p[(int64_t)(c + 3) * per];
    acc += v0; acc += v1; acc += v2; acc += v3;
  }
  for (; c < cnt; ++c) acc += p[(int64_t)c * per];
  C[(int64_t)g * strideC + (e / N) * ldc + e % N] = acc;
}

__global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t* __restrict__ p, uint32_t v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = v;
}

__global__ __launch_bounds__(256) void fill_zero_rows_kernel(float* __restrict__ C, int64_t ldc, int64_t M, int64_t N) {
  const int64_t total = M * N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    C[(i / N) * ldc + i % N] = 0.f;
}

void fill_u32(void* p, uint32_t value, int64_t n_words, hipStream_t s) {
  if (n_words <= 0) return;
  const unsigned blocks = (unsigned)std::min<int64_t>((n_words + 255) / 256, 4096);
  hipLaunchKernelGGL(fill_u32_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<uint32_t*>(p), value, n_words);
}

void fill_zero_rows_f32(float* C, int64_t ldc, int64_t M, int64_t N, hipStream_t s) {
  if (M <= 0 || N <= 0) return;
  if (ldc == N) { fill_u32(C, 0u, M * N, s); return; }
  const unsigned blocks = (unsigned)std::min<int64_t>((M * N + 255) / 256, 4096);
  hipLaunchKernelGGL(fill_zero_rows_kernel, dim3(blocks), dim3(256), 0, s, C, ldc, M, N);
}

void splitk_reduce(const float* ws, float* C, int64_t ldc, int64_t strideC, int M, int N, const int32_t* group_ptr, int G,
                   int kchunk, int n_chunks, hipStream_t s) {
  const int64_t total = (int64_t)M * N * G;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ws, C, ldc, strideC, M, N,
                     group_ptr, G, kchunk, n_chunks);
}
}  // namespace bkg

extern "C" int bkg_set_rng_epoch(const uint32_t* epoch) {
  bkg::g_rng_epoch = epoch;
  return 0;
}

extern "C" int bkg_set_deterministic_workspace(float* workspace, int64_t floats) {
  bkg::g_det = {workspace, workspace ? floats : 0};
  return 0;
}

extern "C" const char* bkg_last_error() { return bkg::g_err; }
extern "C" int bkg_abi_version() { return 3; }
