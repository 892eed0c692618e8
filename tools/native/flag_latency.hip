// Cross-workgroup hand-off latency on MI355X (development probe for the persistent tail factorisation):
// G persistent workgroups form a chain: workgroup w waits for flag[w-1] (acquire, agent scope), optionally reads the
// `bytes` its predecessor wrote, writes `bytes` of its own, releases flag[w].  Consecutive workgroup ids sit on
// different XCDs (round-robin dispatch), so every hop crosses an L2.  Prints ns per hop and verifies the data.
//   hipcc -O3 --offload-arch=gfx950 tools/native/flag_latency.hip -o flag_latency && ./flag_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void chain(int* flag, double* data, int doubles_per_wg, int* abort_flag) {
  const int w = blockIdx.x, tid = threadIdx.x;
  if (w > 0) {
    if (tid == 0) {
      long spins = 0;
      while (__hip_atomic_load(flag + w - 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < 1) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1L << 24)) { atomicExch(abort_flag, 1); break; }
      }
    }
    __syncthreads();
  }
  const double* src = data + (size_t)(w > 0 ? w - 1 : 0) * doubles_per_wg;
  double* dst = data + (size_t)w * doubles_per_wg;
  for (int i = tid; i < doubles_per_wg; i += 256) dst[i] = (w > 0 ? src[i] : 0.0) + 1.0;
  __threadfence();
  __syncthreads();
  if (tid == 0) __hip_atomic_store(flag + w, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

int main() {
  const int G = 256;
  for (int kb : {0, 1, 32, 128}) {
    const int dpw = kb * 1024 / 8;
    int *flag, *ab;
    double* data;
    hipMalloc(&flag, G * sizeof(int));
    hipMalloc(&ab, sizeof(int));
    hipMalloc(&data, (size_t)G * (dpw + 1) * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemset(flag, 0, G * sizeof(int));
      hipMemset(ab, 0, sizeof(int));
      hipMemset(data, 0, (size_t)G * (dpw + 1) * sizeof(double));
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(chain, dim3(G), dim3(256), 0, 0, flag, data, dpw, ab);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    int h_ab = 0;
    hipMemcpy(&h_ab, ab, sizeof(int), hipMemcpyDeviceToHost);
    double last = 0.0;
    if (dpw) hipMemcpy(&last, data + (size_t)(G - 1) * dpw + dpw - 1, sizeof(double), hipMemcpyDeviceToHost);
    printf("chain of %d workgroups, %3d KiB written per hop: %.2f us per hop (total %.1f us), abort=%d, last=%g (expect %d)\n", G, kb,
           best * 1e3f / G, best * 1e3f, h_ab, last, dpw ? G : 0);
    hipFree(flag); hipFree(ab); hipFree(data);
  }
  return 0;
}
