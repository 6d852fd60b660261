// Lane maps of v_permlane32_swap / v_permlane16_swap on gfx950: inputs a[l] = l, b[l] = 100 + l.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
  unsigned c = threadIdx.x, d = 100 + threadIdx.x;
  auto q = __builtin_amdgcn_permlane16_swap(c, d, false, false);
  o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
}
int main() {
  unsigned* o; hipMalloc(&o, 1024); unsigned h[256];
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
  hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
  const char* names[4] = {"swap32 r0", "swap32 r1", "swap16 r0", "swap16 r1"};
  for (int t = 0; t < 4; ++t) { printf("%s:", names[t]); for (int i = 0; i < 64; i += 4) printf(" %u", h[64 * t + i]); printf("\n"); }
  return 0;
}
