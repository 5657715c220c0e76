#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>
template <class Cfg>
float run(const char* name, uint32_t* k_in, uint32_t* k_out, uint32_t* v_in, uint32_t* v_out, size_t n, int bits, std::vector<uint32_t>& ref) {
  size_t bytes = 0;
  rocprim::radix_sort_pairs<Cfg>(nullptr, bytes, k_in, k_out, v_in, v_out, n, 0, bits, 0);
  void* ws; hipMalloc(&ws, bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) rocprim::radix_sort_pairs<Cfg>(ws, bytes, k_in, k_out, v_in, v_out, n, 0, bits, 0);
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) rocprim::radix_sort_pairs<Cfg>(ws, bytes, k_in, k_out, v_in, v_out, n, 0, bits, 0);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<uint32_t> out(n); hipMemcpy(out.data(), k_out, n * 4, hipMemcpyDeviceToHost);
  bool ok = out == ref;
  std::vector<uint32_t> vo(n); hipMemcpy(vo.data(), v_out, n * 4, hipMemcpyDeviceToHost);
  bool stable = true; for (size_t i = 1; i < n; ++i) if (out[i] == out[i-1] && vo[i] < vo[i-1]) { stable = false; break; }
  printf("%-28s n=%zu bits=%d  %.1f us  keys %s stable %s ws=%zu\n", name, n, bits, ms / 20 * 1e3, ok ? "ok" : "WRONG", stable ? "yes" : "NO", bytes);
  hipFree(ws); return ms;
}
int main() {
  using namespace rocprim;
  for (size_t n : {(size_t)8000000, (size_t)4000000, (size_t)1000000, (size_t)300000, (size_t)100000, (size_t)40000}) for (int bits : {17, 22}) {
    std::vector<uint32_t> h(n), v(n); std::mt19937 g(1);
    for (size_t i = 0; i < n; ++i) { h[i] = g() & ((1u << bits) - 1); if (bits == 17) h[i] %= 130000; v[i] = (uint32_t)i; }
    std::vector<uint32_t> ref = h; std::sort(ref.begin(), ref.end());
    uint32_t *ki, *ko, *vi, *vo; hipMalloc(&ki, n*4); hipMalloc(&ko, n*4); hipMalloc(&vi, n*4); hipMalloc(&vo, n*4);
    hipMemcpy(ki, h.data(), n*4, hipMemcpyHostToDevice); hipMemcpy(vi, v.data(), n*4, hipMemcpyHostToDevice);
    run<default_config>("default", ki, ko, vi, vo, n, bits, ref);
    run<radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024,8>, kernel_config<1024,8>, 9, block_radix_rank_algorithm::match>, 32768>>("9b 1024x8 ml32k", ki, ko, vi, vo, n, bits, ref);
    run<radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024,12>, kernel_config<1024,12>, 9, block_radix_rank_algorithm::match>, 32768>>("9b 1024x12 ml32k", ki, ko, vi, vo, n, bits, ref);
    run<radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024,6>, kernel_config<1024,6>, 9, block_radix_rank_algorithm::match>, 32768>>("9b 1024x6 ml32k", ki, ko, vi, vo, n, bits, ref);
    run<radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024,8>, kernel_config<1024,8>, 10, block_radix_rank_algorithm::match>, 32768>>("10b 1024x8 ml32k", ki, ko, vi, vo, n, bits, ref);
    run<radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024,8>, kernel_config<1024,8>, 11, block_radix_rank_algorithm::match>, 32768>>("11b 1024x8 ml32k", ki, ko, vi, vo, n, bits, ref);
    run<radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024,8>, kernel_config<1024,8>, 8, block_radix_rank_algorithm::match>, 32768>>("8b 1024x8 ml32k", ki, ko, vi, vo, n, bits, ref);
    run<radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<1024,8>, kernel_config<1024,8>, 6, block_radix_rank_algorithm::match>, 32768>>("6b 1024x8 ml32k", ki, ko, vi, vo, n, bits, ref);
    hipFree(ki); hipFree(ko); hipFree(vi); hipFree(vo);
  }
}
