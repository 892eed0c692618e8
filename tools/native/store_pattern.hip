// Write bandwidth by store pattern (development probe for the covariance assembly): every workgroup (256 threads) writes
// one R x C tile of a 16384 x 16384 fp64 matrix, 16 bytes per lane, tiles handed out row-major.
//   hipcc -O3 --offload-arch=gfx950 tools/native/store_pattern.hip -o store_pattern && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2_t __attribute__((ext_vector_type(2)));

template <int R, int C>
__global__ __launch_bounds__(256) void fill_tiles(double* K, int n, double v) {
  const int tiles_n = n / C;
  const int ti = blockIdx.x / tiles_n, tj = blockIdx.x % tiles_n;
  constexpr int PAIRS = R * C / 2;                 // 16-byte pairs per tile
  for (int idx = threadIdx.x; idx < PAIRS; idx += 256) {
    const int r = idx / (C / 2), c2 = idx % (C / 2);
    d2_t x = {v + r, v + c2};
    *reinterpret_cast<d2_t*>(K + (size_t)(ti * R + r) * n + tj * C + 2 * c2) = x;
  }
}

template <int R, int C>
void run(double* K, int n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill_tiles<R, C>), dim3((n / R) * (n / C)), dim3(256), 0, 0, K, n, 1.0 + rep);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("tile %4d x %4d : %.3f ms  %.2f TB/s\n", R, C, best, 8.0 * n * n / best / 1e9);
}

int main() {
  const int n = 16384;
  double* K;
  hipMalloc(&K, (size_t)n * n * 8);
  run<128, 128>(K, n);
  run<64, 256>(K, n);
  run<32, 512>(K, n);
  run<16, 1024>(K, n);
  run<8, 2048>(K, n);
  run<4, 4096>(K, n);
  run<256, 64>(K, n);
  run<64, 64>(K, n);
  return 0;
}
