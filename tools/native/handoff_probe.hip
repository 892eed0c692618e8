// Cross-STREAM hand-off latency on MI355X (development probe, round 5): what one dependency between two hardware queues costs
// under each mechanism the look-ahead of the factorisation could use.  HOPS short kernels alternate between two streams; kernel k
// reads what kernel k - 1 wrote (data[k] = data[k - 1] + 1, verified at the end: the mechanism must also make the data visible).
//   same      all kernels on ONE stream (the in-stream boundary: the floor)
//   event     hipEventRecord on the producer's stream + hipStreamWaitEvent on the consumer's   (what potrf.hip does today)
//   value     hipStreamWriteValue32 behind the producer + hipStreamWaitValue32 in front of the consumer (command-processor packets)
//   kflag     the producer kernel itself publishes the value (release, agent scope) + hipStreamWaitValue32
//   kgate     the producer kernel publishes + a one-wave gate kernel polls it in front of the consumer
//   vgate     hipStreamWriteValue32 + the gate kernel
//   valdev    as `value`, the word in plain hipMalloc memory instead of hipMallocSignalMemory
//   hipcc -O3 --offload-arch=gfx950 tools/native/handoff_probe.hip -o handoff_probe && ./handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void work(double* data, int k, int spin, unsigned* flag, unsigned value, unsigned* done_ctr) {
  const int tid = threadIdx.x + blockIdx.x * 256;
  // (workgroup b reads what workgroup b + 1 of the previous kernel wrote: another XCD's L2 held those lines)
  const double prev = (k > 0) ? data[(size_t)(k - 1) * 1024 + ((tid + 256) & 1023)] : 0.0;
  for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);
  data[(size_t)k * 1024 + tid] = prev + 1.0;
  if (flag) {      // last workgroup out publishes: every workgroup releases its stores, the last one writes the value
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned n = __hip_atomic_fetch_add(done_ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (n == gridDim.x - 1) {
        __hip_atomic_store(done_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

__global__ void gate(unsigned* flag, unsigned value, int* timeout) {
  long spins = 0;
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < value) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1L << 20)) { *timeout = 1; break; }
  }
}

int main(int argc, char** argv) {
  const int HOPS = 200;
  const int spin = argc > 1 ? atoi(argv[1]) : 100;
  double* data;
  unsigned *sig = nullptr, *ctr;
  int* tmo;
  CK(hipMalloc(&data, (size_t)HOPS * 1024 * sizeof(double)));
  CK(hipMalloc(&ctr, 4));
  CK(hipMalloc(&tmo, 4));
  hipError_t es = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory);
  printf("signal memory: %s\n", hipGetErrorString(es));
  if (es != hipSuccess) CK(hipMalloc(&sig, 8));
  hipStream_t s[2];
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&s[0], hipStreamNonBlocking, lo));
  CK(hipStreamCreateWithPriority(&s[1], hipStreamNonBlocking, hi));
  hipEvent_t ev[4];
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const char* names[] = {"same", "event", "value", "kflag", "kgate", "vgate", "valdev"};
  unsigned* sig_dev;
  CK(hipMalloc(&sig_dev, 8));
  unsigned* const sig_signal = sig;
  double base = 0.0;
  for (int mode = 0; mode < 7; ++mode) {
    sig = (mode == 6) ? sig_dev : sig_signal;
    double best = 1e30;
    int bad = 0, h_tmo = 0;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipMemset(data, 0, (size_t)HOPS * 1024 * sizeof(double)));
      CK(hipMemset(sig, 0, 8));
      CK(hipMemset(ctr, 0, 4));
      CK(hipMemset(tmo, 0, 4));
      CK(hipDeviceSynchronize());
      const auto t0 = std::chrono::steady_clock::now();
      for (int k = 0; k < HOPS; ++k) {
        hipStream_t me = (mode == 0) ? s[0] : s[k & 1], other = (mode == 0) ? s[0] : s[(k + 1) & 1];
        const unsigned v = (unsigned)(k + 1);      // published when kernel k is done; kernel k waits for value k
        if (k > 0) {
          if (mode == 2 || mode == 3 || mode == 6) CK(hipStreamWaitValue32(me, sig, (unsigned)k, hipStreamWaitValueGte, 0xffffffffu));
          if (mode == 4 || mode == 5) hipLaunchKernelGGL(gate, dim3(1), dim3(1), 0, me, sig, (unsigned)k, tmo);
        }
        const bool kpub = (mode == 3 || mode == 4);
        hipLaunchKernelGGL(work, dim3(4), dim3(256), 0, me, data, k, spin, kpub ? sig : nullptr, v, ctr);
        if (mode == 1) {
          CK(hipEventRecord(ev[k & 3], me));
          CK(hipStreamWaitEvent(other, ev[k & 3], 0));
        }
        if (mode == 2 || mode == 5 || mode == 6) CK(hipStreamWriteValue32(me, sig, v, 0));
      }
      CK(hipDeviceSynchronize());
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (us < best) best = us;
      static double last[1024];
      CK(hipMemcpy(last, data + (size_t)(HOPS - 1) * 1024, sizeof(last), hipMemcpyDeviceToHost));
      for (int i = 0; i < 1024; ++i)
        if (last[i] != (double)HOPS) bad = 1;
      CK(hipMemcpy(&h_tmo, tmo, 4, hipMemcpyDeviceToHost));
    }
    if (mode == 0) base = best;
    printf("%-6s %8.2f us per hop  (+%.2f over the in-stream boundary)%s%s\n", names[mode], best / HOPS, (best - base) / HOPS,
           bad ? "  DATA WRONG" : "", h_tmo ? "  GATE TIMED OUT" : "");
  }
  return 0;
}
