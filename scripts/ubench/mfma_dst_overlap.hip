// Micro-benchmark (gfx950): v_mfma_f32_16x16x32_bf16 whose destination is also its B (or A) operand, alone and in the
// dependent pair hipcc emits at the end of an accumulation chain:
//     X = A1 * X + C        (vDst == srcB)
//     Y = A2 * Y + X        (vDst == srcB, srcC = previous result)
// Expected with A = all ones (bf16), X0 = Y0 = ones in every k slot: X = 32 + c, Y = 32 + X = 64 + c.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int KIND, int GAP>
__global__ void probe(unsigned* stale, unsigned* rows, int iters) {
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const float c = (float)((it * 7 + threadIdx.x) & 1023);
    float r0, r1;
    asm volatile(
        "v_mov_b32 v32, 0x3f803f80\n\tv_mov_b32 v33, 0x3f803f80\n\tv_mov_b32 v34, 0x3f803f80\n\tv_mov_b32 v35, 0x3f803f80\n\t"   // X (as B: ones)
        "v_mov_b32 v36, 0x3f803f80\n\tv_mov_b32 v37, 0x3f803f80\n\tv_mov_b32 v38, 0x3f803f80\n\tv_mov_b32 v39, 0x3f803f80\n\t"   // A
        "v_mov_b32 v40, 0x3f803f80\n\tv_mov_b32 v41, 0x3f803f80\n\tv_mov_b32 v42, 0x3f803f80\n\tv_mov_b32 v43, 0x3f803f80\n\t"   // Y (as B: ones)
        "v_mov_b32 v44, %2\n\tv_mov_b32 v45, %2\n\tv_mov_b32 v46, %2\n\tv_mov_b32 v47, %2\n\t"                                 // C
        "s_nop 15\n\t"
        ".if %3 == 0\n\t"                                                        // dst == B
        "v_mfma_f32_16x16x32_bf16 v[32:35], v[36:39], v[32:35], v[44:47]\n\t"
        ".if %4 == 1\n\ts_barrier\n\t.endif\n\t"
        "v_mfma_f32_16x16x32_bf16 v[40:43], v[36:39], v[40:43], v[32:35]\n\t"
        ".else\n\t"                                                              // dst == A
        "v_mfma_f32_16x16x32_bf16 v[32:35], v[32:35], v[36:39], v[44:47]\n\t"
        ".if %4 == 1\n\ts_barrier\n\t.endif\n\t"
        "v_mfma_f32_16x16x32_bf16 v[40:43], v[40:43], v[36:39], v[32:35]\n\t"
        ".endif\n\t"
        "v_mov_b32 v36, 0\n\tv_mov_b32 v37, 0\n\tv_mov_b32 v38, 0\n\tv_mov_b32 v39, 0\n\t"                   // hipcc reuses the shared operand right away
        "s_nop 15\n\ts_nop 15\n\t"
        "v_mov_b32 %0, v32\n\tv_mov_b32 %1, v40\n\t"
        : "=v"(r0), "=v"(r1)
        : "v"(c), "i"(KIND), "i"(GAP)
        : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    if (r0 != c + 32.0f || r1 != c + 64.0f) { ++bad; atomicAdd(&rows[(threadIdx.x & 63) >> 4], 1u); }
  }
  if (bad) atomicAdd(stale, bad);
}

template <int KIND, int GAP>
void run(const char* name, unsigned* d) {
  for (int threads : {64, 512}) {
    hipMemset(d, 0, 20);
    hipLaunchKernelGGL((probe<KIND, GAP>), dim3(512), dim3(threads), 0, 0, d, d + 1, 2000);
    unsigned h[5];
    hipMemcpy(h, d, 20, hipMemcpyDeviceToHost);
    printf("%-44s block %4d: wrong results %u   | by 16-lane row: %u %u %u %u\n", name, threads, h[0], h[1], h[2], h[3], h[4]);
  }
}
int main() {
  unsigned* d;
  hipMalloc(&d, 64);
  run<0, 0>("vDst == srcB, dependent pair back to back", d);
  run<0, 1>("vDst == srcB, s_barrier between the pair", d);
  run<1, 0>("vDst == srcA, dependent pair back to back", d);
  run<1, 1>("vDst == srcA, s_barrier between the pair", d);
  return 0;
}
