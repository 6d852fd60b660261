// Micro-benchmark: issue cost (cycles per wave-instruction) of VALU instruction kinds on gfx950, with one
// wave per SIMD and with two waves per SIMD running the same stream.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int iters) {
  float a[16];
  f32x2 p[16];
  for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f32x2{a[i], a[i] + 1.f}; }
  const float c = 1.0001f, d = 0.5f;
  const f32x2 c2 = {1.0001f, 0.9999f}, d2 = {0.5f, 0.25f};
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
      if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(c2), "v"(d2));
      if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 4) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
      if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
      if (KIND == 6) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (KIND == 7) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
      if (KIND == 8) { asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p[i][0]) : "v"(c), "v"(d)); }
      if (KIND == 9) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (KIND == 10) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
      if (KIND == 11) asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(a[i]));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += a[i] + p[i][0] + p[i][1];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND> void run(const char* name, int per_iter) {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 64 * 8 * sizeof(unsigned long long)); hipMalloc(&sink, 64 * 512 * sizeof(float));
  const int iters = 2000;
  for (int waves : {4, 8}) {
    hipLaunchKernelGGL(k<KIND>, dim3(8), dim3(64 * waves), 0, 0, out, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(64);
    hipMemcpy(h.data(), out, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost);
    double avg = 0; for (int w = 0; w < waves; ++w) avg += (double)h[w]; avg /= waves;
    printf("%-28s waves/SIMD=%d  ticks per wave-instruction %.2f  (per SIMD: %.2f)\n", name, waves / 4, avg / (iters * 16.0 * per_iter),
           avg / (iters * 16.0 * per_iter) / (waves / 4));
  }
  hipFree(out); hipFree(sink);
}
int main() {
  run<0>("v_fma_f32", 1); run<9>("v_add_f32", 1); run<1>("v_pk_fma_f32", 1); run<5>("v_pk_mul_f32", 1); run<10>("v_pk_add_f32", 1);
  run<2>("v_exp_f32", 1); run<3>("v_rcp_f32", 1); run<4>("v_sin_f32", 1); run<6>("v_cvt_pk_bf16_f32", 1); run<7>("v_med3_f32", 1);
  run<11>("v_pk_max_i16", 1); run<8>("v_exp_f32+v_fma_f32 pair", 2);
  return 0;
}
