// what __builtin_amdgcn_permlane32_swap(a, b, false, false) returns in each lane (development probe)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  const unsigned lane = threadIdx.x;
  const unsigned a = 1000 + lane, b = 2000 + lane;
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[lane] = r[0];
  out[64 + lane] = r[1];
  const auto s = __builtin_amdgcn_permlane32_swap(a, a, false, false);
  out[128 + lane] = s[0];
  out[192 + lane] = s[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int part = 0; part < 4; ++part) { printf("part %d: lane0 %u lane1 %u lane31 %u lane32 %u lane33 %u lane63 %u\n", part, h[part*64], h[part*64+1], h[part*64+31], h[part*64+32], h[part*64+33], h[part*64+63]); }
  return 0;
}
