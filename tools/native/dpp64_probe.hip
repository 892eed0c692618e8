// Issue cost (8 independent chains) and dependent latency (1 chain) of the instructions of the diagonal-block kernel's pivot step,
// one wave alone on its SIMD (gfx950): the 32-bit DPP pair against the DP-ALU DPP forms (v_mov_b64_dpp, v_fmac_f64_dpp).
//   hipcc -O3 --offload-arch=gfx950 tools/native/dpp64_probe.hip -o dpp64_probe && ./dpp64_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define N 256
template <int MODE, int CH>
__global__ void probe(double* out, unsigned long long* cyc, double seed) {
  const int lane = threadIdx.x;
  double x[8], y = 1.0 - lane * 1e-4;
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = seed + lane * 1e-3 + k;
  const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      if (MODE == 0) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[k]) : "v"(y));
      if (MODE == 1) {
        int lo = __double2loint(x[k]), hi = __double2hiint(x[k]);
        asm volatile("v_mov_b32_dpp %0, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf"
                     : "+v"(lo), "+v"(hi));
        x[k] = __hiloint2double(hi, lo);
      }
      if (MODE == 2) asm volatile("v_mov_b64_dpp %0, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(x[k]));
      if (MODE == 3) asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(x[k]) : "v"(y));
      if (MODE == 4) asm volatile("v_fmac_f64_e32 %0, %0, %1" : "+v"(x[k]) : "v"(y));
      if (MODE == 5) asm volatile("v_rcp_f64_e32 %0, %0" : "+v"(x[k]));
      if (MODE == 6) {
        int lo = __double2loint(x[k]), hi = __double2hiint(x[k]);
        asm volatile("v_cndmask_b32_e32 %0, 0, %0, vcc\n\tv_cndmask_b32_e32 %1, 0, %1, vcc" : "+v"(lo), "+v"(hi));
        x[k] = __hiloint2double(hi, lo);
      }
      if (MODE == 7) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[k]) : "v"(y));
      if (MODE == 8) asm volatile("v_mov_b64 %0, %0" : "+v"(x[k]));   // plain 64-bit move
      if (MODE == 9) {   // the old rank-1 term: two dpp moves + fma (independent of the accumulator chain except through x)
        int lo = __double2loint(x[k]), hi = __double2hiint(x[k]);
        asm volatile("v_mov_b32_dpp %0, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf"
                     : "+v"(lo), "+v"(hi));
        double c = __hiloint2double(hi, lo);
        asm volatile("v_fma_f64 %0, -%1, %2, %0" : "+v"(x[k]) : "v"(c), "v"(y));
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += x[k];
  out[lane] = s;
  if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

template <int MODE>
void run(const char* name, double* out, unsigned long long* cyc) {
  unsigned long long h1[2], h8[2];
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((probe<MODE, 1>), dim3(1), dim3(64), 0, 0, out, cyc, 0.731); hipDeviceSynchronize(); }
  hipMemcpy(h1, cyc, sizeof h1, hipMemcpyDeviceToHost);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((probe<MODE, 8>), dim3(1), dim3(64), 0, 0, out, cyc, 0.731); hipDeviceSynchronize(); }
  hipMemcpy(h8, cyc, sizeof h8, hipMemcpyDeviceToHost);
  printf("%-44s dependent %6.1f ticks %6.2f ns | independent x8: %6.1f ticks %6.2f ns per op\n", name, (double)h1[0] / N, (double)h1[1] * 10.0 / N,
         (double)h8[0] / N / 8, (double)h8[1] * 10.0 / N / 8);
}

int main() {
  double* out;
  unsigned long long* cyc;
  hipMalloc(&out, 64 * sizeof(double));
  hipMalloc(&cyc, 2 * sizeof(unsigned long long));
  run<0>("v_fma_f64", out, cyc);
  run<4>("v_fmac_f64_e32", out, cyc);
  run<7>("v_mul_f64", out, cyc);
  run<8>("v_mov_b64", out, cyc);
  run<1>("2 x v_mov_b32_dpp row_newbcast", out, cyc);
  run<2>("v_mov_b64_dpp row_newbcast", out, cyc);
  run<3>("v_fmac_f64_dpp row_newbcast", out, cyc);
  run<9>("2 x v_mov_b32_dpp + v_fma_f64", out, cyc);
  run<5>("v_rcp_f64", out, cyc);
  run<6>("2 x v_cndmask_b32", out, cyc);
  return 0;
}
