// Micro-benchmark (gfx950): does an LDS-DMA load (buffer_load_dwordx4 ... lds) touch the wave's LGKM counter?
// One wave samples HW_REG_IB_STS (vm_cnt, lgkm_cnt) every few cycles into VGPR lanes while
//   test 0: only an LDS-DMA is in flight
//   test 1: slow scalar loads (cold lines) are in flight and an LDS-DMA of L2-warm data lands in between
//   test 2: only the scalar loads (control)
// and prints the sequence of distinct (vm_cnt, lgkm_cnt) pairs seen.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__global__ __launch_bounds__(64) void probe(const char* warm, const unsigned* cold, unsigned* out, int test) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  // warm the DMA source in L2 / L1
  unsigned w = reinterpret_cast<const unsigned*>(warm)[lane * 4];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(warm), 0, 4096, 0x00020000);
  unsigned s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  unsigned a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  if (test >= 1) {   // four scalar loads from lines 1 MiB apart: HBM misses
    asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40000\n\ts_load_dword %2, %4, 0x80000\n\ts_load_dword %3, %4, 0xc0000"
                 : "=s"(a0), "=s"(a1), "=s"(a2), "=s"(a3) : "s"(cold) : "memory");
  }
  if (test <= 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)smem, 16, lane * 16, 0, 0, 0);
  for (int i = 0; i < 64; ++i) {
    unsigned r;
    asm volatile("s_getreg_b32 %1, hwreg(HW_REG_IB_STS)\n\ts_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0\n\ts_nop 3" : "+v"(s0), "=&s"(r) : "s"(i));
  }
  for (int i = 0; i < 64; ++i) {
    unsigned r;
    asm volatile("s_getreg_b32 %1, hwreg(HW_REG_IB_STS)\n\ts_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0\n\ts_nop 7\n\ts_nop 7" : "+v"(s1), "=&s"(r) : "s"(i));
  }
  for (int i = 0; i < 64; ++i) {
    unsigned r;
    asm volatile("s_getreg_b32 %1, hwreg(HW_REG_IB_STS)\n\ts_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(s2), "=&s"(r) : "s"(i));
  }
  for (int i = 0; i < 64; ++i) {
    unsigned r;
    asm volatile("s_getreg_b32 %1, hwreg(HW_REG_IB_STS)\n\ts_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(s3), "=&s"(r) : "s"(i));
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  out[lane] = s0; out[64 + lane] = s1; out[128 + lane] = s2; out[192 + lane] = s3;
  if (lane == 0) out[256] = a0 + a1 + a2 + a3 + w + reinterpret_cast<unsigned*>(smem)[5];
}

int main() {
  char* warm; unsigned* cold; unsigned* out;
  hipMalloc(&warm, 4096); hipMalloc(&cold, 8 << 20); hipMalloc(&out, 4 * 260);
  hipMemset(warm, 1, 4096); hipMemset(cold, 0, 8 << 20);
  // flush caches for the cold lines: touch a large buffer
  char* big; hipMalloc(&big, 1 << 30);
  for (int test = 0; test < 3; ++test) {
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(big, rep, 1 << 30);
      hipDeviceSynchronize();
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 65536, 0, warm, cold, out, test);
      std::vector<unsigned> h(257);
      hipMemcpy(h.data(), out, 4 * 257, hipMemcpyDeviceToHost);
      printf("test %d rep %d (sample index: vm_cnt, lgkm_cnt):", test, rep);
      int pv = -1, pl = -1;
      for (int i = 0; i < 256; ++i) {
        const unsigned r = h[i];
        const int vm = (r & 0xF) | ((r >> 22) & 0x3) << 4, lg = (r >> 8) & 0xF;
        if (vm != pv || lg != pl) { printf(" [%d: %d,%d]", i, vm, lg); pv = vm; pl = lg; }
      }
      printf("\n");
    }
  }
  return 0;
}
