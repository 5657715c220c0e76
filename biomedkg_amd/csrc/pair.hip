// Side-by-side views of a bf16 table: pair[i] = [x[i] | x[perm[i]]] and its gradient x_grad[i] = g[i, :h] + g[inv[i], h:]
// (inv = the inverse permutation).  GGD's corrupted input is a row permutation of the positive one and
// x[perm] W^T = (x W^T)[perm] (reference biomedkg/model/gcl.py:69-71,85), so both views' first-layer gather rows come
// from one product; this is the glue that lays them side by side (one pass each way, 16 B per lane, no atomics).
#include "common.h"

namespace bkg {

__global__ __launch_bounds__(256) void pair_rows_kernel(const uint4* __restrict__ x, const int64_t* __restrict__ perm,
                                                        uint4* __restrict__ out, int64_t n, int pieces) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;     // one 16-byte piece of one output half-row
  if (t >= n * pieces * 2) return;
  const int64_t i = t / (2 * pieces);
  const int p = (int)(t % (2 * pieces));
  const int64_t src = p < pieces ? i : perm[i];
  out[t] = x[src * pieces + (p < pieces ? p : p - pieces)];
}

__device__ __forceinline__ float2 pr_widen(uint32_t w) {
  return make_float2(__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u));
}
__device__ __forceinline__ uint32_t pr_sum2(uint32_t a, uint32_t b) {
  const float2 fa = pr_widen(a), fb = pr_widen(b);
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 h;
  h[0] = (__bf16)(fa.x + fb.x); h[1] = (__bf16)(fa.y + fb.y);
  return __builtin_bit_cast(uint32_t, h);
}

__global__ __launch_bounds__(256) void unpair_rows_kernel(const uint4* __restrict__ g, const int64_t* __restrict__ inv,
                                                          uint4* __restrict__ out, int64_t n, int pieces) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;     // one 16-byte piece of one input-gradient row
  if (t >= n * pieces) return;
  const int64_t i = t / pieces;
  const int p = (int)(t % pieces);
  const uint4 a = g[i * 2 * pieces + p], b = g[inv[i] * 2 * pieces + pieces + p];
  out[t] = make_uint4(pr_sum2(a.x, b.x), pr_sum2(a.y, b.y), pr_sum2(a.z, b.z), pr_sum2(a.w, b.w));
}

}  // namespace bkg

using namespace bkg;

extern "C" int bkg_pair_rows_bf16(const void* x, const int64_t* perm, void* out, int64_t n, int32_t h, void* stream) {
  BKG_ARG(n >= 0 && h > 0 && h % 8 == 0, "bkg_pair_rows_bf16: h must be a multiple of 8");
  BKG_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)out % 16 == 0, "bkg_pair_rows_bf16: 16-byte alignment");
  if (n == 0) return 0;
  hipLaunchKernelGGL(pair_rows_kernel, dim3(ceil_div(n * (h / 8) * 2, 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const uint4*>(x), perm, reinterpret_cast<uint4*>(out), n, h / 8);
  return check_launch("bkg_pair_rows_bf16");
}

extern "C" int bkg_unpair_rows_bf16(const void* g, const int64_t* inv, void* out, int64_t n, int32_t h, void* stream) {
  BKG_ARG(n >= 0 && h > 0 && h % 8 == 0, "bkg_unpair_rows_bf16: h must be a multiple of 8");
  BKG_ARG((uintptr_t)g % 16 == 0 && (uintptr_t)out % 16 == 0, "bkg_unpair_rows_bf16: 16-byte alignment");
  if (n == 0) return 0;
  hipLaunchKernelGGL(unpair_rows_kernel, dim3(ceil_div(n * (h / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const uint4*>(g), inv, reinterpret_cast<uint4*>(out), n, h / 8);
  return check_launch("bkg_unpair_rows_bf16");
}
