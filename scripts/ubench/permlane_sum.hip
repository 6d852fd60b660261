// Cross-quad sums (lanes l, l^16, l^32, l^48) two ways: __shfl_xor (ds_bpermute_b32, through the LDS crossbar) and the gfx950
// lane-swap instructions v_permlane32_swap / v_permlane16_swap (pure VALU).  Checks that both give the same sums and times a
// dependent chain of each.   hipcc --offload-arch=gfx950 -O3 -o /tmp/permlane_sum scripts/ubench/permlane_sum.hip && /tmp/permlane_sum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ float xq_shfl(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ float xq_swap(float v) {
  // (inline asm: with the builtin, hipcc 7.2 adds the first result to itself -- it loses the second one)
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));      // a = [a.lo | b.lo], b = [a.hi | b.hi]
  float s = a + b, c = s, d = s;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(c), "+v"(d));      // odd rows of c <-> even rows of d
  return c + d;
}
__global__ void check(const float* in, float* o1, float* o2) {
  const float v = in[threadIdx.x];
  o1[threadIdx.x] = xq_shfl(v);
  o2[threadIdx.x] = xq_swap(v);
}
template <int MODE> __global__ void chain(float* o, int n, unsigned long long* cyc) {
  float v = o[threadIdx.x];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) v = (MODE ? xq_swap(v) : xq_shfl(v)) * 0.25f + 1.0f;
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  o[threadIdx.x] = v;
  if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float *in, *o1, *o2; unsigned long long* cyc;
  hipMalloc(&in, 256); hipMalloc(&o1, 256); hipMalloc(&o2, 256); hipMalloc(&cyc, 8);
  std::vector<float> h(64);
  for (int i = 0; i < 64; ++i) h[i] = 0.37f * i - 3.f + (i % 7) * 0.11f;
  hipMemcpy(in, h.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, in, o1, o2);
  std::vector<float> a(64), b(64);
  hipMemcpy(a.data(), o1, 256, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, 256, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) {
    const float ref = h[i & 15] + h[(i & 15) + 16] + h[(i & 15) + 32] + h[(i & 15) + 48];
    if (fabsf(a[i] - ref) > 1e-4f || fabsf(b[i] - ref) > 1e-4f) { ++bad; printf("lane %d ref %f shfl %f swap %f\n", i, ref, a[i], b[i]); }
  }
  printf("cross-quad sums: %d lanes wrong\n", bad);
  for (int mode = 0; mode < 2; ++mode) {
    unsigned long long c = 0;
    if (mode) hipLaunchKernelGGL(chain<1>, dim3(1), dim3(64), 0, 0, o1, 1000, cyc);
    else hipLaunchKernelGGL(chain<0>, dim3(1), dim3(64), 0, 0, o1, 1000, cyc);
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%s: %.1f ticks per cross-quad sum (dependent chain, one wave)\n", mode ? "permlane swap" : "ds_bpermute  ", c / 1000.0);
  }
  return bad != 0;
}
