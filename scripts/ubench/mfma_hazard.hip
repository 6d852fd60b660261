// Micro-benchmark (gfx950): how many wait states does a VALU read of a v_mfma_f32_16x16x32_bf16 result need?
// hipcc pads 8 (s_nop 7: its model has the instruction at 4 passes); the hand-written stages of enf_gemm_asm.h use 12.
// Each wave runs: [PRE back-to-back MFMAs on other accumulators], the tested MFMA (acc = c + 32 with all-ones operands),
// s_nop (N-1), v_mov reading acc[0].  A read that comes too early returns the old accumulator value c.
// Usage: ./mfma_hazard            -> table of stale reads per (waves per SIMD, PRE, N)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int N, int PRE>
__global__ void probe(unsigned* stale, int iters) {
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const float c = (float)((it * 7 + threadIdx.x) & 1023);
    float r;
    // v[32:35] tested accumulator, v[36:39] A, v[40:43] B (bf16 1.0 = 0x3F80 pairs), v[44:55] three other accumulators
    asm volatile(
        "v_mov_b32 v32, %1\n\tv_mov_b32 v33, %1\n\tv_mov_b32 v34, %1\n\tv_mov_b32 v35, %1\n\t"
        "v_mov_b32 v36, 0x3f803f80\n\tv_mov_b32 v37, 0x3f803f80\n\tv_mov_b32 v38, 0x3f803f80\n\tv_mov_b32 v39, 0x3f803f80\n\t"
        "v_mov_b32 v40, 0x3f803f80\n\tv_mov_b32 v41, 0x3f803f80\n\tv_mov_b32 v42, 0x3f803f80\n\tv_mov_b32 v43, 0x3f803f80\n\t"
        "v_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\t"
        "v_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\tv_mov_b32 v50, 0\n\tv_mov_b32 v51, 0\n\t"
        "v_mov_b32 v52, 0\n\tv_mov_b32 v53, 0\n\tv_mov_b32 v54, 0\n\tv_mov_b32 v55, 0\n\t"
        "s_nop 15\n\t"
        ".if %2 > 0\n\tv_mfma_f32_16x16x32_bf16 v[44:47], v[36:39], v[40:43], v[44:47]\n\t.endif\n\t"
        ".if %2 > 1\n\tv_mfma_f32_16x16x32_bf16 v[48:51], v[36:39], v[40:43], v[48:51]\n\t.endif\n\t"
        ".if %2 > 2\n\tv_mfma_f32_16x16x32_bf16 v[52:55], v[36:39], v[40:43], v[52:55]\n\t.endif\n\t"
        "v_mfma_f32_16x16x32_bf16 v[32:35], v[36:39], v[40:43], v[32:35]\n\t"
        ".if %3 > 0\n\ts_nop %3 - 1\n\t.endif\n\t"
        "v_mov_b32 %0, v32\n\t"
        "s_nop 15\n\ts_nop 15\n\t"
        : "=v"(r)
        : "v"(c), "i"(PRE), "i"(N)
        : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48",
          "v49", "v50", "v51", "v52", "v53", "v54", "v55");
    bad += r != c + 32.0f;
  }
  if (bad) atomicAdd(stale, bad);
}

template <int N, int PRE>
unsigned run(int threads, unsigned* d) {
  hipMemset(d, 0, 4);
  hipLaunchKernelGGL((probe<N, PRE>), dim3(512), dim3(threads), 0, 0, d, 2000);
  unsigned h = 0;
  hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
  return h;
}

template <int PRE>
void table(unsigned* d) {
  for (int threads : {256, 512, 1024}) {
    unsigned r[17] = {run<0, PRE>(threads, d),  run<1, PRE>(threads, d),  run<2, PRE>(threads, d),  run<3, PRE>(threads, d),
                      run<4, PRE>(threads, d),  run<5, PRE>(threads, d),  run<6, PRE>(threads, d),  run<7, PRE>(threads, d),
                      run<8, PRE>(threads, d),  run<9, PRE>(threads, d),  run<10, PRE>(threads, d), run<11, PRE>(threads, d),
                      run<12, PRE>(threads, d), run<13, PRE>(threads, d), run<14, PRE>(threads, d), run<15, PRE>(threads, d),
                      run<16, PRE>(threads, d)};
    printf("PRE %d MFMAs ahead, %d waves/SIMD: stale reads for N = 0..16 wait states:", PRE, threads / 256);
    for (int i = 0; i < 17; ++i) printf(" %u", r[i]);
    printf("\n");
  }
}

int main() {
  unsigned* d;
  hipMalloc(&d, 4);
  table<0>(d);
  table<1>(d);
  table<3>(d);
  return 0;
}
