// Micro-benchmark (gfx950): wait states needed between a VALU write of a VGPR and a v_mfma_f32_16x16x32_bf16 that reads it as
// A / B / C.  hipcc pads to 2.  Producer variants: v_cvt_pk_bf16_f32 (new on gfx950), v_mov_b32, v_perm_b32, v_pk_mul_f32.
// The operand register holds 0 long before; the producer writes bf16 ones (or c for srcC); a stale read changes the result.
#include <hip/hip_runtime.h>
#include <cstdio>

// KIND: 0 cvt_pk -> B, 1 mov -> B, 2 cvt_pk -> A, 3 perm -> B, 4 pk_mul -> B (two registers), 5 mov -> C
template <int KIND, int N>
__global__ void probe(unsigned* stale, int iters) {
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const float c = (float)((it * 7 + threadIdx.x) & 1023);
    float r;
    asm volatile(
        "v_mov_b32 v32, %1\n\tv_mov_b32 v33, %1\n\tv_mov_b32 v34, %1\n\tv_mov_b32 v35, %1\n\t"
        "v_mov_b32 v36, 0x3f803f80\n\tv_mov_b32 v37, 0x3f803f80\n\tv_mov_b32 v38, 0x3f803f80\n\tv_mov_b32 v39, 0x3f803f80\n\t"
        "v_mov_b32 v40, 0x3f803f80\n\tv_mov_b32 v41, 0x3f803f80\n\tv_mov_b32 v42, 0x3f803f80\n\tv_mov_b32 v43, 0x3f803f80\n\t"
        "v_mov_b32 v44, 1.0\n\tv_mov_b32 v45, 0x3f803f80\n\tv_mov_b32 v46, 0x05040100\n\tv_mov_b32 v47, 1.0\n\t"
        ".if %2 == 0 || %2 == 1 || %2 == 3\n\tv_mov_b32 v41, 0\n\t.endif\n\t"
        ".if %2 == 2\n\tv_mov_b32 v37, 0\n\t.endif\n\t"
        ".if %2 == 4\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\t.endif\n\t"
        ".if %2 == 5\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v32, 0\n\t.endif\n\t"
        "s_nop 15\n\ts_nop 15\n\t"
        ".if %2 == 0\n\tv_cvt_pk_bf16_f32 v41, v44, v44\n\t.endif\n\t"
        ".if %2 == 1\n\tv_mov_b32 v41, v45\n\t.endif\n\t"
        ".if %2 == 2\n\tv_cvt_pk_bf16_f32 v37, v44, v44\n\t.endif\n\t"
        ".if %2 == 3\n\tv_perm_b32 v41, v45, v45, v46\n\t.endif\n\t"
        ".if %2 == 4\n\tv_pk_mul_f32 v[42:43], v[44:45], v[44:45] op_sel_hi:[0,0]\n\t.endif\n\t"
        ".if %2 == 5\n\tv_mov_b32 v32, %1\n\t.endif\n\t"
        ".if %3 > 0\n\ts_nop %3 - 1\n\t.endif\n\t"
        "v_mfma_f32_16x16x32_bf16 v[32:35], v[36:39], v[40:43], v[32:35]\n\t"
        "s_nop 15\n\t"
        "v_mov_b32 %0, v32\n\t"
        : "=v"(r)
        : "v"(c), "i"(KIND), "i"(N)
        : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    // KIND 4: v[42:43] = (1.0f * 1.0f) twice as f32 = 0x3f800000: bf16 pairs (0, 1.0): half of the k values of two registers
    const float expect = KIND == 4 ? c + 32.0f - 8.0f : c + 32.0f;
    bad += r != expect;
  }
  if (bad) atomicAdd(stale, bad);
}

template <int KIND, int N>
unsigned run(unsigned* d) {
  hipMemset(d, 0, 4);
  hipLaunchKernelGGL((probe<KIND, N>), dim3(512), dim3(512), 0, 0, d, 2000);
  unsigned h = 0;
  hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
  return h;
}
template <int KIND>
void row(const char* name, unsigned* d) {
  printf("%-28s stale results for N = 0..5 wait states: %u %u %u %u %u %u\n", name, run<KIND, 0>(d), run<KIND, 1>(d), run<KIND, 2>(d),
         run<KIND, 3>(d), run<KIND, 4>(d), run<KIND, 5>(d));
}
int main() {
  unsigned* d;
  hipMalloc(&d, 4);
  row<0>("v_cvt_pk_bf16_f32 -> B", d);
  row<1>("v_mov_b32 -> B", d);
  row<2>("v_cvt_pk_bf16_f32 -> A", d);
  row<3>("v_perm_b32 -> B", d);
  row<4>("v_pk_mul_f32 -> B", d);
  row<5>("v_mov_b32 -> C", d);
  return 0;
}
