// dev probe: v_permlane16_swap through inline asm (the builtin returns the new vdst twice on ROCm 7.2)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void swap16(float& x, float& y) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
}
__global__ void probe(float* out) {
  const int lane = threadIdx.x;
  float x = lane, y = 100 + lane;
  x = x * 1.0f + 0.0f; y = y + 0.0f;
  swap16(x, y);
  out[lane] = x;
  out[64 + lane] = y;
  out[128 + lane] = x + y;
}
int main() {
  float* d; (void)hipMalloc(&d, 384 * 4);
  probe<<<1, 64>>>(d);
  float h[384]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[3] = {"x'", "y'", "x'+y'"};
  for (int k = 0; k < 3; ++k) { printf("%s:", names[k]); for (int i = 0; i < 64; ++i) printf(" %g", h[64 * k + i]); printf("\n"); }
  return 0;
}
