// The half-wave sum of syevd_internal.h in two forms -- row totals through scalar registers (8 readlanes) and through
// v_permlane16_swap (gfx950) -- on random data: every lane of every wave must get the same bits from both.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/native/wsum_check.hip -o wsum_check && ./wsum_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

template <int CTRL>
__device__ __forceinline__ double dpp_add(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return x + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rdlane(double x, int l) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  return __hiloint2double(__builtin_amdgcn_readlane(hi, l), __builtin_amdgcn_readlane(lo, l));
}
__device__ __forceinline__ double rows16(double x) {
  x = dpp_add<0xB1>(x);
  x = dpp_add<0x4E>(x);
  x = dpp_add<0x141>(x);
  return dpp_add<0x140>(x);
}
__device__ __forceinline__ double wsum32_readlane(double x, int lane) {
  x = rows16(x);
  const double r0 = rdlane(x, 0), r1 = rdlane(x, 16), r2 = rdlane(x, 32), r3 = rdlane(x, 48);
  return (lane < 32) ? r0 + r1 : r2 + r3;
}
__device__ __forceinline__ double wsum32_swap(double x) {
  x = rows16(x);
  const int lo = __double2loint(x), hi = __double2hiint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
// x + (the value 32 lanes away): through the LDS crossbar (__shfl_xor) and through v_permlane32_swap
__device__ __forceinline__ double halves_shfl(double x) { return x + __shfl_xor(x, 32); }
__device__ __forceinline__ double halves_swap(double x) {
  const int lo = __double2loint(x), hi = __double2hiint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__global__ void both(const double* in, double* o1, double* o2, double* o3, double* o4) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  o1[t] = wsum32_readlane(in[t], threadIdx.x);
  o2[t] = wsum32_swap(in[t]);
  o3[t] = halves_shfl(in[t]);
  o4[t] = halves_swap(in[t]);
}
int main() {
  const int waves = 4096, n = waves * 64;
  std::vector<double> h(n), a(n), b(n), c(n), d(n);
  srand(7);
  for (auto& v : h) v = ((double)rand() / RAND_MAX - 0.5) * pow(10.0, rand() % 12 - 6);
  double *din, *d1, *d2, *d3, *d4;
  if (hipMalloc(&din, n * 8) != hipSuccess || hipMalloc(&d1, n * 8) != hipSuccess || hipMalloc(&d2, n * 8) != hipSuccess ||
      hipMalloc(&d3, n * 8) != hipSuccess || hipMalloc(&d4, n * 8) != hipSuccess) return 1;
  if (hipMemcpy(din, h.data(), n * 8, hipMemcpyHostToDevice) != hipSuccess) return 1;
  hipLaunchKernelGGL(both, dim3(waves), dim3(64), 0, 0, din, d1, d2, d3, d4);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  if (hipMemcpy(a.data(), d1, n * 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(b.data(), d2, n * 8, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(c.data(), d3, n * 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(d.data(), d4, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  long bad = 0, bad2 = 0;
  for (int i = 0; i < n; ++i) {
    bad += memcmp(&a[i], &b[i], 8) != 0;
    bad2 += memcmp(&c[i], &d[i], 8) != 0;
  }
  printf("sum across the halves: %ld of %d lanes differ between __shfl_xor(x, 32) and v_permlane32_swap\n", bad2, n);
  double ref = 0.0;
  for (int i = 0; i < 32; ++i) ref += h[i];
  printf("%ld of %d lanes differ between the two forms; wave 0 lanes 0-31: readlane form %.17g, swap form %.17g, host sum in order %.17g\n", bad, n, a[0], b[0], ref);
  return bad != 0 || bad2 != 0;
}
