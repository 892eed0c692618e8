// Write bandwidth of the assembly's CURRENT store pattern: every wave instruction writes 4 rows x 16 lanes x 8 bytes (128-byte
// row segments), 64 x 64 tiles per 256-thread workgroup, 16 stores per thread -- against the 16-byte-per-lane fill of
// store_pattern.hip.   hipcc -O3 --offload-arch=gfx950 tools/native/store_pattern8.hip -o sp8 && ./sp8
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void fill8(double* K, int n, double v) {
  const int tiles_n = n / 64;
  const int ti = blockIdx.x / tiles_n, tj = blockIdx.x % tiles_n;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) K[(size_t)(ti * 64 + ty + 16 * i) * n + tj * 64 + tx + 16 * j] = v + i + j;
}
// MFMA-output layout: lane (g = lane >> 4, c = lane & 15) of wave w writes rows 16 w + g + 4 r, column c + 16 b
__global__ __launch_bounds__(256) void fill8m(double* K, int n, double v) {
  const int tiles_n = n / 64;
  const int ti = blockIdx.x / tiles_n, tj = blockIdx.x % tiles_n;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) K[(size_t)(ti * 64 + 16 * w + g + 4 * r) * n + tj * 64 + 16 * b + c] = v + b + r;
}

int main() {
  const int n = 16384;
  double* K;
  if (hipMalloc(&K, (size_t)n * n * 8) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int which = 0; which < 2; ++which) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      (void)hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(fill8, dim3((n / 64) * (n / 64)), dim3(256), 0, 0, K, n, 1.0 + rep);
      else hipLaunchKernelGGL(fill8m, dim3((n / 64) * (n / 64)), dim3(256), 0, 0, K, n, 1.0 + rep);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%s: %.3f ms  %.2f TB/s\n", which ? "MFMA-output layout, 8 B per lane" : "assembly layout, 8 B per lane   ", best, 8.0 * n * n / best / 1e9);
  }
  return 0;
}
