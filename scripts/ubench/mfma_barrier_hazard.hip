// Micro-benchmark (gfx950): is an MFMA that was issued just before s_barrier finished when the wave leaves the barrier?
//   v_mfma X = A*B + c ; s_barrier ; s_nop (N-1) ; v_mov r = X[0]
// hipcc counts the s_barrier as ONE wait state of the MFMA -> VALU hazard (7 needed, mfma_hazard.hip).  Half of the waves of
// each workgroup are delayed before the barrier (s_sleep), so the others wait there for a long time.
// Variant DEP: a second MFMA (srcC = X, other destination) is issued right AFTER the barrier and its result is read instead.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int DEP, int N>
__global__ void probe(unsigned* stale, unsigned* rows, int iters) {
  unsigned bad = 0;
  const int wave = threadIdx.x >> 6;
  for (int it = 0; it < iters; ++it) {
    const float c = (float)((it * 7 + threadIdx.x) & 1023);
    float r;
    if ((wave + it) & 1) { __builtin_amdgcn_s_sleep(20); }
    asm volatile(
        "v_mov_b32 v32, %1\n\tv_mov_b32 v33, %1\n\tv_mov_b32 v34, %1\n\tv_mov_b32 v35, %1\n\t"
        "v_mov_b32 v36, 0x3f803f80\n\tv_mov_b32 v37, 0x3f803f80\n\tv_mov_b32 v38, 0x3f803f80\n\tv_mov_b32 v39, 0x3f803f80\n\t"
        "v_mov_b32 v40, 0x3f803f80\n\tv_mov_b32 v41, 0x3f803f80\n\tv_mov_b32 v42, 0x3f803f80\n\tv_mov_b32 v43, 0x3f803f80\n\t"
        "v_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\t"
        "s_nop 15\n\t"
        "v_mfma_f32_16x16x32_bf16 v[32:35], v[36:39], v[40:43], v[32:35]\n\t"
        "s_barrier\n\t"
        ".if %2 == 1\n\tv_mfma_f32_16x16x32_bf16 v[44:47], v[36:39], v[40:43], v[32:35]\n\t.endif\n\t"
        ".if %3 > 0\n\ts_nop %3 - 1\n\t.endif\n\t"
        ".if %2 == 1\n\tv_mov_b32 %0, v44\n\t.else\n\tv_mov_b32 %0, v32\n\t.endif\n\t"
        "s_nop 15\n\t"
        : "=v"(r)
        : "v"(c), "i"(DEP), "i"(N)
        : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    const float expect = DEP ? c + 64.0f : c + 32.0f;
    if (r != expect) { ++bad; atomicAdd(&rows[(threadIdx.x & 63) >> 4], 1u); }
  }
  if (bad) atomicAdd(stale, bad);
}

template <int DEP, int N>
unsigned run(int threads, unsigned* d) {
  hipMemset(d, 0, 4);
  hipLaunchKernelGGL((probe<DEP, N>), dim3(512), dim3(threads), 0, 0, d, d + 1, 1000);
  unsigned h = 0;
  hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
  return h;
}
template <int DEP>
void row(unsigned* d) {
  for (int threads : {256, 512}) {
    hipMemset(d, 0, 20);
    printf("%s, block %4d: stale reads for N = 0..10 wait states after the barrier:", DEP ? "MFMA2(srcC = X) issued after the barrier, read Y" : "read X after the barrier", threads);
    unsigned r[11] = {run<DEP, 0>(threads, d), run<DEP, 1>(threads, d), run<DEP, 2>(threads, d), run<DEP, 3>(threads, d), run<DEP, 4>(threads, d),
                      run<DEP, 5>(threads, d), run<DEP, 6>(threads, d), run<DEP, 7>(threads, d), run<DEP, 8>(threads, d), run<DEP, 9>(threads, d), run<DEP, 10>(threads, d)};
    for (int i = 0; i < 11; ++i) printf(" %u", r[i]);
    unsigned q[4];
    hipMemcpy(q, d + 1, 16, hipMemcpyDeviceToHost);
    printf("   | by 16-lane row: %u %u %u %u\n", q[0], q[1], q[2], q[3]);
  }
}
int main() {
  unsigned* d;
  hipMalloc(&d, 64);
  row<0>(d);
  row<1>(d);
  return 0;
}
