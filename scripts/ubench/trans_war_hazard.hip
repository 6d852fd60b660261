// Micro-benchmark (gfx950): WAR hazard on the SOURCE of a transcendental op.
//   v_rcp_f32 vD, vS ; s_nop (N-1) ; v_mov_b32 vS, <other value>
// hipcc emits the overwrite in the very next instruction (it knows no such hazard).  If the quarter-rate trans unit reads vS
// for its later 16-lane passes after the next full-rate VALU has already written them -- more likely when K earlier trans ops
// are still queued in the unit -- vD comes out as rcp(<other value>) in the upper lanes.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int K, int N>
__global__ void probe(unsigned* stale, unsigned* rows, int iters) {
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const float x = (float)(((it * 7 + threadIdx.x) & 7) + 1) * 0.25f;     // 0.25 .. 2
    float r, ref;
    asm volatile(
        "v_mov_b32 v48, %2\n\tv_mov_b32 v50, 0x42c80000\n\t"              // v48 = x, v50 = 100.0 (the overwriting value)
        "v_rcp_f32 %1, v48\n\ts_nop 15\n\t"                                 // reference rcp(x)
        "v_mov_b32 v49, v48\n\ts_nop 7\n\t"                                 // vS = x
        ".if %3 > 7\n\tv_exp_f32 v40, v48\n\t.endif\n\t"
        ".if %3 > 6\n\tv_exp_f32 v41, v48\n\t.endif\n\t"
        ".if %3 > 5\n\tv_exp_f32 v42, v48\n\t.endif\n\t"
        ".if %3 > 4\n\tv_exp_f32 v43, v48\n\t.endif\n\t"
        ".if %3 > 3\n\tv_exp_f32 v44, v48\n\t.endif\n\t"
        ".if %3 > 2\n\tv_exp_f32 v45, v48\n\t.endif\n\t"
        ".if %3 > 1\n\tv_exp_f32 v46, v48\n\t.endif\n\t"
        ".if %3 > 0\n\tv_exp_f32 v47, v48\n\t.endif\n\t"
        "v_rcp_f32 v51, v49\n\t"                                            // the tested trans op: reads v49
        ".if %4 > 0\n\ts_nop %4 - 1\n\t.endif\n\t"
        "v_mov_b32 v49, v50\n\t"                                            // overwrite its source
        "s_nop 15\n\t"
        "v_mov_b32 %0, v51\n\t"
        : "=v"(r), "=v"(ref)
        : "v"(x), "i"(K), "i"(N)
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
    if (__float_as_uint(r) != __float_as_uint(ref)) { ++bad; atomicAdd(&rows[(threadIdx.x & 63) >> 4], 1u); }
  }
  if (bad) atomicAdd(stale, bad);
}

template <int K, int N>
unsigned run(int threads, unsigned* d) {
  hipMemset(d, 0, 4);
  hipLaunchKernelGGL((probe<K, N>), dim3(512), dim3(threads), 0, 0, d, d + 1, 2000);
  unsigned h = 0;
  hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
  return h;
}
template <int K>
void row(unsigned* d) {
  for (int threads : {64, 512, 1024}) {
    hipMemset(d, 0, 20);
    printf("%d trans ops queued ahead, block %4d: wrong rcp results for N = 0..6 wait states before the source overwrite:", K, threads);
    unsigned r[7] = {run<K, 0>(threads, d), run<K, 1>(threads, d), run<K, 2>(threads, d), run<K, 3>(threads, d), run<K, 4>(threads, d),
                     run<K, 5>(threads, d), run<K, 6>(threads, d)};
    for (int i = 0; i < 7; ++i) printf(" %u", r[i]);
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
  row<2>(d);
  row<4>(d);
  row<8>(d);
  return 0;
}
