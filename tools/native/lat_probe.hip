// Dependent-latency probe for the instructions on the diagonal-block kernel's pivot chain (gfx950): one wave, N dependent
// repetitions of each pattern, cycles per repetition from s_memtime (shader clock) and wall_clock64 (100 MHz).
//   hipcc -O3 --offload-arch=gfx950 tools/native/lat_probe.hip -o lat_probe && ./lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double dpp_share5(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x155, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, 0x155, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bperm(double x, int src) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_ds_bpermute(src << 2, lo);
  hi = __builtin_amdgcn_ds_bpermute(src << 2, hi);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane5(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_readlane(lo, 5);
  hi = __builtin_amdgcn_readlane(hi, 5);
  return __hiloint2double(hi, lo);
}

#define N 512
template <int MODE>
__global__ void probe(double* out, unsigned long long* cyc, double seed, __attribute__((address_space(3))) double* unused = nullptr) {
  __shared__ double sm[64 * 17];
  const int lane = threadIdx.x;
  double x = seed + lane * 1e-3, y = 1.0 - lane * 1e-4;
  d4_t acc = {x, y, x, y};
  sm[lane] = x;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
#pragma unroll 8
  for (int i = 0; i < N; ++i) {
    if (MODE == 0) x = __builtin_fma(x, y, 0.5);                                   // v_fma_f64
    if (MODE == 1) x = __builtin_amdgcn_rcp(x) + 0.0 * y;                           // v_rcp_f64 (+ one add to keep it alive)
    if (MODE == 2) x = dpp_share5(x);                                               // 2 x v_mov_dpp
    if (MODE == 3) x = bperm(x, (lane + 1) & 63);                                   // 2 x ds_bpermute
    if (MODE == 4) x = readlane5(x) * y;                                            // 2 x v_readlane + v_mul_f64
    if (MODE == 5) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);  // dependent MFMA chain (through acc)
    if (MODE == 6) { sm[lane * 17 % 1024] = x; x = sm[(lane * 17 + 17) % 1024] + 1.0; }   // ds_write + ds_read round trip
    if (MODE == 7) x = __builtin_fma(dpp_share5(x), y, 0.5);                        // dpp + fma
    if (MODE == 8) x = __builtin_amdgcn_rcp(x);                                     // v_rcp_f64 only
    if (MODE == 9) { double e = __builtin_fma(-x, y, 1.0); x = __builtin_fma(e, e, e); }   // two dependent fma
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  out[lane] = x + acc[0] + acc[1] + acc[2] + acc[3];
  if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

template <int MODE>
void run(const char* name, double* out, unsigned long long* cyc) {
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(64), 0, 0, out, cyc, 0.731);
    hipDeviceSynchronize();
  }
  unsigned long long h[2];
  hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  printf("%-34s %7.1f memtime ticks / rep   %7.2f ns / rep\n", name, (double)h[0] / N, (double)h[1] * 10.0 / N);
}

int main() {
  double* out;
  unsigned long long* cyc;
  hipMalloc(&out, 64 * sizeof(double));
  hipMalloc(&cyc, 2 * sizeof(unsigned long long));
  run<0>("v_fma_f64 dependent", out, cyc);
  run<9>("2 x v_fma_f64 dependent", out, cyc);
  run<8>("v_rcp_f64 dependent", out, cyc);
  run<1>("v_rcp_f64 + v_fma dependent", out, cyc);
  run<2>("2 x v_mov_dpp (row_share) dep.", out, cyc);
  run<7>("dpp row_share + v_fma_f64", out, cyc);
  run<3>("2 x ds_bpermute dependent", out, cyc);
  run<4>("2 x v_readlane + v_mul_f64", out, cyc);
  run<5>("v_mfma_f64_16x16x4 dependent", out, cyc);
  run<6>("ds_write + ds_read round trip", out, cyc);
  return 0;
}
