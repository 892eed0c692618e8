// How a grid of one-wave workgroups is dealt to the XCDs and CUs (the XCD-local bulge chase, sb2st_chase<true>, launches 8 x the waves
// it needs and keeps those that land on one XCD): per launch, the number of workgroups each XCC_ID received and the number of distinct
// CUs among them, for the grids the chase uses (128 / 256 / 512 workgroups), while every workgroup stays resident for ~20 us.
//   hipcc --offload-arch=gfx950 -O2 -o xcc_deal_probe xcc_deal_probe.hip && ./xcc_deal_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

__global__ __launch_bounds__(64) void deal(unsigned* cnt, unsigned long long* cus, int spin) {
  unsigned xcc = 0, hwid = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  xcc &= 0xf;
  // HW_ID: bits 11:8 CU_ID, 12 SH_ID, 15:13 SE_ID  -> a 7-bit CU index inside the XCD
  const unsigned cu = ((hwid >> 8) & 0xf) | (((hwid >> 12) & 0x1) << 4) | (((hwid >> 13) & 0x7) << 5);
  if (threadIdx.x == 0) {
    atomicAdd(cnt + xcc, 1u);
    atomicOr(cus + xcc * 2 + (cu >> 6), 1ull << (cu & 63));
  }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
}

int main() {
  unsigned* cnt; unsigned long long* cus;
  hipMalloc(&cnt, 16 * sizeof(unsigned));
  hipMalloc(&cus, 32 * sizeof(unsigned long long));
  for (int grid : {128, 256, 512, 1024}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(cnt, 0, 16 * sizeof(unsigned));
      hipMemset(cus, 0, 32 * sizeof(unsigned long long));
      hipLaunchKernelGGL(deal, dim3(grid), dim3(64), 0, 0, cnt, cus, 2000);   // 100 MHz wall clock: 20 us
      hipDeviceSynchronize();
      unsigned h[16]; unsigned long long c[32];
      hipMemcpy(h, cnt, sizeof(h), hipMemcpyDeviceToHost);
      hipMemcpy(c, cus, sizeof(c), hipMemcpyDeviceToHost);
      printf("grid %4d rep %d  workgroups per XCC:", grid, rep);
      for (int x = 0; x < 8; ++x) printf(" %3u", h[x]);
      printf("   distinct CUs:");
      for (int x = 0; x < 8; ++x) printf(" %2d", __builtin_popcountll(c[2 * x]) + __builtin_popcountll(c[2 * x + 1]));
      printf("\n");
    }
  }
  return 0;
}
