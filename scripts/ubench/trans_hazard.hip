// Micro-benchmark (gfx950): wait states needed between a transcendental VALU op (v_exp_f32 / v_rcp_f32 / v_sin_f32) and a
// consumer of its result.  hipcc pads 1 (trans -> non-trans VALU).  A stale read sees the register's old value (0).
#include <hip/hip_runtime.h>
#include <cstdio>

// KIND: 0 exp -> v_add_f32, 1 exp -> v_rcp_f32 (trans -> trans), 2 exp -> v_pk_mul_f32, 3 exp -> v_cvt_pk_bf16_f32,
//       4 rcp -> v_mul_f32, 5 sin -> v_cvt_pk_bf16_f32, 6 exp -> v_fma_f32 (as addend), 7 v_add (plain VALU) -> v_exp (control)
template <int KIND, int N>
__global__ void probe(unsigned* stale, int iters) {
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const float x = (float)(((it * 7 + threadIdx.x) & 7) + 1);        // 1..8
    float r, ref;
    asm volatile(
        "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, %2\n\tv_mov_b32 v43, %2\n\tv_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\t"
        "s_nop 15\n\t"
        // reference: the same producer and consumer with a long gap
        ".if %3 == 0\n\tv_exp_f32 v44, v42\n\ts_nop 7\n\tv_add_f32 v45, 1.0, v44\n\t.endif\n\t"
        ".if %3 == 1\n\tv_exp_f32 v44, v42\n\ts_nop 7\n\tv_rcp_f32 v45, v44\n\t.endif\n\t"
        ".if %3 == 2\n\tv_exp_f32 v44, v42\n\tv_mov_b32 v45, v42\n\ts_nop 7\n\tv_pk_mul_f32 v[44:45], v[44:45], v[42:43]\n\ts_nop 1\n\tv_mov_b32 v45, v44\n\t.endif\n\t"
        ".if %3 == 3\n\tv_exp_f32 v44, v42\n\ts_nop 7\n\tv_cvt_pk_bf16_f32 v45, v44, v44\n\t.endif\n\t"
        ".if %3 == 4\n\tv_rcp_f32 v44, v42\n\ts_nop 7\n\tv_mul_f32 v45, v42, v44\n\tv_add_f32 v45, v45, v44\n\t.endif\n\t"
        ".if %3 == 5\n\tv_sin_f32 v44, v42\n\ts_nop 7\n\tv_cvt_pk_bf16_f32 v45, v44, v42\n\t.endif\n\t"
        ".if %3 == 6\n\tv_exp_f32 v44, v42\n\ts_nop 7\n\tv_fma_f32 v45, v42, v42, v44\n\t.endif\n\t"
        ".if %3 == 7\n\tv_add_f32 v44, 1.0, v42\n\ts_nop 7\n\tv_exp_f32 v45, v44\n\t.endif\n\t"
        "s_nop 7\n\tv_mov_b32 %1, v45\n\t"
        // test: old value of v40 is 0
        ".if %3 == 0\n\tv_exp_f32 v40, v42\n\t.if %4 > 0\n\ts_nop %4 - 1\n\t.endif\n\tv_add_f32 v41, 1.0, v40\n\t.endif\n\t"
        ".if %3 == 1\n\tv_exp_f32 v40, v42\n\t.if %4 > 0\n\ts_nop %4 - 1\n\t.endif\n\tv_rcp_f32 v41, v40\n\t.endif\n\t"
        ".if %3 == 2\n\tv_mov_b32 v41, v42\n\ts_nop 7\n\tv_exp_f32 v40, v42\n\t.if %4 > 0\n\ts_nop %4 - 1\n\t.endif\n\tv_pk_mul_f32 v[40:41], v[40:41], v[42:43]\n\ts_nop 1\n\tv_mov_b32 v41, v40\n\t.endif\n\t"
        ".if %3 == 3\n\tv_exp_f32 v40, v42\n\t.if %4 > 0\n\ts_nop %4 - 1\n\t.endif\n\tv_cvt_pk_bf16_f32 v41, v40, v40\n\t.endif\n\t"
        ".if %3 == 4\n\tv_rcp_f32 v40, v42\n\t.if %4 > 0\n\ts_nop %4 - 1\n\t.endif\n\tv_mul_f32 v41, v42, v40\n\ts_nop 3\n\tv_add_f32 v41, v41, v40\n\t.endif\n\t"
        ".if %3 == 5\n\tv_sin_f32 v40, v42\n\t.if %4 > 0\n\ts_nop %4 - 1\n\t.endif\n\tv_cvt_pk_bf16_f32 v41, v40, v42\n\t.endif\n\t"
        ".if %3 == 6\n\tv_exp_f32 v40, v42\n\t.if %4 > 0\n\ts_nop %4 - 1\n\t.endif\n\tv_fma_f32 v41, v42, v42, v40\n\t.endif\n\t"
        ".if %3 == 7\n\tv_add_f32 v40, 1.0, v42\n\t.if %4 > 0\n\ts_nop %4 - 1\n\t.endif\n\tv_exp_f32 v41, v40\n\t.endif\n\t"
        "s_nop 7\n\tv_mov_b32 %0, v41\n\t"
        : "=v"(r), "=v"(ref)
        : "v"(x * 0.25f), "i"(KIND), "i"(N)
        : "v40", "v41", "v42", "v43", "v44", "v45");
    bad += __float_as_uint(r) != __float_as_uint(ref);
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
  printf("%-32s stale results for N = 0..4 wait states: %u %u %u %u %u\n", name, run<KIND, 0>(d), run<KIND, 1>(d), run<KIND, 2>(d),
         run<KIND, 3>(d), run<KIND, 4>(d));
}
int main() {
  unsigned* d;
  hipMalloc(&d, 4);
  row<0>("v_exp_f32 -> v_add_f32", d);
  row<1>("v_exp_f32 -> v_rcp_f32", d);
  row<2>("v_exp_f32 -> v_pk_mul_f32", d);
  row<3>("v_exp_f32 -> v_cvt_pk_bf16_f32", d);
  row<4>("v_rcp_f32 -> v_mul_f32", d);
  row<5>("v_sin_f32 -> v_cvt_pk_bf16_f32", d);
  row<6>("v_exp_f32 -> v_fma_f32 (src2)", d);
  row<7>("v_add_f32 -> v_exp_f32", d);
  return 0;
}
