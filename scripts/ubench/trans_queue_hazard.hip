// Micro-benchmark (gfx950): trans -> VALU hazard under load.  hipcc pads ONE wait state between a transcendental op and a
// non-trans VALU that reads its result (measured sufficient for a lone trans op, trans_hazard.hip).  Question: is one still
// enough when K independent trans ops were issued back-to-back just before (the quarter-rate trans unit backed up), as in a
// gelu epilogue (exp, exp, rcp, rcp ...), with several waves per SIMD doing the same?
// Per iteration: K x v_exp_f32 (distinct destinations, same source) ; s_nop (N-1) ; v_add_f32 r = 1.0 + <last destination>.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int K, int N>
__global__ void probe(unsigned* stale, unsigned* lanes, int iters) {
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    const float x = (float)(((it * 7 + threadIdx.x) & 7) + 1) * 0.25f;
    float r, ref;
    asm volatile(
        "v_mov_b32 v40, 0\n\tv_mov_b32 v41, 0\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0\n\t"
        "v_mov_b32 v44, 0\n\tv_mov_b32 v45, 0\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\t"
        "v_mov_b32 v48, %2\n\t"
        "v_exp_f32 v49, v48\n\ts_nop 15\n\tv_add_f32 %1, 1.0, v49\n\ts_nop 7\n\t"          // reference
        ".if %3 > 7\n\tv_exp_f32 v40, v48\n\t.endif\n\t"
        ".if %3 > 6\n\tv_exp_f32 v41, v48\n\t.endif\n\t"
        ".if %3 > 5\n\tv_exp_f32 v42, v48\n\t.endif\n\t"
        ".if %3 > 4\n\tv_exp_f32 v43, v48\n\t.endif\n\t"
        ".if %3 > 3\n\tv_exp_f32 v44, v48\n\t.endif\n\t"
        ".if %3 > 2\n\tv_exp_f32 v45, v48\n\t.endif\n\t"
        ".if %3 > 1\n\tv_exp_f32 v46, v48\n\t.endif\n\t"
        "v_exp_f32 v47, v48\n\t"
        ".if %4 > 0\n\ts_nop %4 - 1\n\t.endif\n\t"
        "v_add_f32 %0, 1.0, v47\n\t"
        "s_nop 15\n\t"
        : "=v"(r), "=v"(ref)
        : "v"(x), "i"(K), "i"(N)
        : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49");
    if (__float_as_uint(r) != __float_as_uint(ref)) { ++bad; atomicAdd(&lanes[(threadIdx.x & 63) >> 4], 1u); }
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
    printf("%d back-to-back v_exp_f32, block %4d: stale reads of the last result for N = 0..8 wait states:", K, threads);
    unsigned r[9] = {run<K, 0>(threads, d), run<K, 1>(threads, d), run<K, 2>(threads, d), run<K, 3>(threads, d), run<K, 4>(threads, d),
                     run<K, 5>(threads, d), run<K, 6>(threads, d), run<K, 7>(threads, d), run<K, 8>(threads, d)};
    for (int i = 0; i < 9; ++i) printf(" %u", r[i]);
    unsigned q[4];
    hipMemcpy(q, d + 1, 16, hipMemcpyDeviceToHost);
    printf("   | stale reads by 16-lane row: %u %u %u %u\n", q[0], q[1], q[2], q[3]);
  }
}
int main() {
  unsigned* d;
  hipMalloc(&d, 64);
  row<1>(d);
  row<2>(d);
  row<4>(d);
  row<8>(d);
  return 0;
}
