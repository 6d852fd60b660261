// Micro-benchmark (gfx950): WAR hazard on the C operand of a v_mfma_f32_16x16x32_bf16.
// hipcc lets a VALU (or an LDS load) overwrite an MFMA's srcA / srcB register in the very next instruction: its hazard table
// only protects srcC.  Question: when the MFMA cannot start at once -- its srcC is the result of the MFMA issued just before it
// (DEP), and/or other waves' MFMAs occupy the SIMD's matrix pipe -- does it still read A / B before the overwrite lands?
// Sequence per iteration:  [PRE independent MFMAs] ; MFMA1: X = ones*ones + c ; MFMA2: Y = A*B + (DEP ? X : c') ;
//                          s_nop (N-1) ; v_mov A[0..3] = 0 (clobber) ; ... read Y.   Expected Y = 32 + srcC.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int DEP, int PRE, int N>
__global__ void probe(unsigned* stale, int iters) {
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const float c = (float)((it * 7 + threadIdx.x) & 1023);
    float r;
    asm volatile(
        "v_mov_b32 v32, %1\n\tv_mov_b32 v33, %1\n\tv_mov_b32 v34, %1\n\tv_mov_b32 v35, %1\n\t"             // X (srcC of MFMA1)
        "v_mov_b32 v36, 0x3f803f80\n\tv_mov_b32 v37, 0x3f803f80\n\tv_mov_b32 v38, 0x3f803f80\n\tv_mov_b32 v39, 0x3f803f80\n\t"   // A
        "v_mov_b32 v40, 0x3f803f80\n\tv_mov_b32 v41, 0x3f803f80\n\tv_mov_b32 v42, 0x3f803f80\n\tv_mov_b32 v43, 0x3f803f80\n\t"   // B
        "v_mov_b32 v44, %1\n\tv_mov_b32 v45, %1\n\tv_mov_b32 v46, %1\n\tv_mov_b32 v47, %1\n\t"             // Y / independent srcC
        "v_mov_b32 v48, 0x3f803f80\n\tv_mov_b32 v49, 0x3f803f80\n\tv_mov_b32 v50, 0x3f803f80\n\tv_mov_b32 v51, 0x3f803f80\n\t"   // A of MFMA1 / PRE
        "v_mov_b32 v52, 0\n\tv_mov_b32 v53, 0\n\tv_mov_b32 v54, 0\n\tv_mov_b32 v55, 0\n\t"
        "v_mov_b32 v56, 0\n\tv_mov_b32 v57, 0\n\tv_mov_b32 v58, 0\n\tv_mov_b32 v59, 0\n\t"
        "s_nop 15\n\t"
        ".if %3 > 0\n\tv_mfma_f32_16x16x32_bf16 v[52:55], v[48:51], v[40:43], v[52:55]\n\t.endif\n\t"
        ".if %3 > 1\n\tv_mfma_f32_16x16x32_bf16 v[56:59], v[48:51], v[40:43], v[56:59]\n\t.endif\n\t"
        ".if %3 > 2\n\tv_mfma_f32_16x16x32_bf16 v[52:55], v[48:51], v[40:43], v[52:55]\n\t.endif\n\t"
        ".if %3 > 3\n\tv_mfma_f32_16x16x32_bf16 v[56:59], v[48:51], v[40:43], v[56:59]\n\t.endif\n\t"
        "v_mfma_f32_16x16x32_bf16 v[32:35], v[48:51], v[40:43], v[32:35]\n\t"                               // MFMA1: X = 32 + c
        ".if %2 == 1\n\tv_mfma_f32_16x16x32_bf16 v[44:47], v[36:39], v[40:43], v[32:35]\n\t.endif\n\t"      // MFMA2 dependent on X
        ".if %2 == 0\n\tv_mfma_f32_16x16x32_bf16 v[44:47], v[36:39], v[40:43], v[44:47]\n\t.endif\n\t"      // MFMA2 independent
        ".if %4 > 0\n\ts_nop %4 - 1\n\t.endif\n\t"
        ".if %2 == 1\n\tv_mov_b32 v32, 0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, 0\n\tv_mov_b32 v35, 0\n\t.endif\n\t"  // clobber srcC (= X) of MFMA2
        "s_nop 15\n\ts_nop 15\n\t"
        "v_mov_b32 %0, v44\n\t"
        : "=v"(r)
        : "v"(c), "i"(DEP), "i"(PRE), "i"(N)
        : "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49",
          "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59");
    const float expect = DEP ? c + 64.0f : c + 32.0f;
    bad += r != expect;
  }
  if (bad) atomicAdd(stale, bad);
}

template <int DEP, int PRE, int N>
unsigned run(int threads, unsigned* d) {
  hipMemset(d, 0, 4);
  hipLaunchKernelGGL((probe<DEP, PRE, N>), dim3(512), dim3(threads), 0, 0, d, 2000);
  unsigned h = 0;
  hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
  return h;
}
template <int DEP, int PRE>
void row(unsigned* d) {
  for (int threads : {64, 512}) {
    printf("MFMA2 %s, %d MFMAs ahead, block %4d: wrong results for N = 0..10 wait states before the A overwrite:", DEP ? "srcC = result of MFMA1" : "independent", PRE, threads);
    unsigned r[11] = {run<DEP, PRE, 0>(threads, d), run<DEP, PRE, 1>(threads, d), run<DEP, PRE, 2>(threads, d), run<DEP, PRE, 3>(threads, d),
                      run<DEP, PRE, 4>(threads, d), run<DEP, PRE, 5>(threads, d), run<DEP, PRE, 6>(threads, d), run<DEP, PRE, 7>(threads, d),
                      run<DEP, PRE, 8>(threads, d), run<DEP, PRE, 9>(threads, d), run<DEP, PRE, 10>(threads, d)};
    for (int i = 0; i < 11; ++i) printf(" %u", r[i]);
    printf("\n");
  }
}
int main() {
  unsigned* d;
  hipMalloc(&d, 4);
  row<0, 0>(d);
  row<1, 0>(d);
  row<0, 4>(d);
  row<1, 4>(d);
  return 0;
}
