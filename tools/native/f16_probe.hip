// The bare 16-pivot loop of the diagonal-block kernel (fidelityfusion_amd/csrc/f16_steps.h), timed with s_memtime:
//   MODE 0 = round-3 step (32-bit DPP moves, row J+1 fetched in the step that needs it), 1 = DP-ALU DPP step, row J+2 two steps ahead
//   NEIGH 0 = the wave alone on the CU; 1 = 7 more waves polling an LDS flag with s_sleep (the helpers while they wait);
//         2 = 7 more waves streaming ds_read_b64 + fp64 MFMA (the helpers at work)
//   hipcc -O3 --offload-arch=gfx950 -Ifidelityfusion_amd/csrc tools/native/f16_probe.hip -o tools/native/f16_probe
#include "f16_steps.h"
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));

template <int MODE, int NEIGH, int SKIPW = 4, int PRIO = 0>
__global__ __launch_bounds__(512) void probe(double* out, unsigned long long* cyc, const double* in, int reps) {
  __shared__ double sm[4096];
  __shared__ int flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 4096; i += blockDim.x) sm[i] = 1.0 + 1e-3 * i;
  if (tid == 0) flag = 0;
  __syncthreads();
  if (wave > 0) {
    if (NEIGH == 0) return;
    if (NEIGH == 3 && wave == SKIPW) return;      // every wave streams except one (SIMD-partner search)
    if (NEIGH == 4 && wave != SKIPW) return;      // only that one streams
    if (NEIGH == 1) {
      volatile int* f = &flag;
      while (*f == 0) __builtin_amdgcn_s_sleep(1);
      return;
    }
    d4_t acc = {0, 0, 0, 0};
    volatile int* f = &flag;
    int it = 0;
    if (NEIGH == 5) {      // MFMA only, operands in registers: no LDS traffic from the neighbours
      const double a = sm[lane], b = sm[lane + 64];
      while (*f == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      }
      out[tid] = acc[0] + acc[1] + acc[2] + acc[3];
      return;
    }
    if (NEIGH == 6) {      // LDS reads only
      double s_ = 0.0;
      while (*f == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s_ += sm[(lane & 15) * 17 + k * 4 + (lane >> 4) + 272 * ((wave + it) & 7)];
        ++it;
      }
      out[tid] = s_;
      return;
    }
    while (*f == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double a = sm[(lane & 15) * 17 + k * 4 + (lane >> 4) + 272 * ((wave + it) & 7)];
        const double b = sm[(lane & 15) * 17 + k * 4 + (lane >> 4) + 272 * ((wave + it + 3) & 7)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      }
      ++it;
    }
    out[tid] = acc[0] + acc[1] + acc[2] + acc[3];
    return;
  }
  if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
  const int g = lane >> 4, c = lane & 15;
  double v[4], w[4];
  unsigned long long total = 0;
  double sink = 0.0;
  for (int rep = 0; rep < reps; ++rep) {
    int cc = c, gg = g;
    asm volatile("" : "+v"(cc), "+v"(gg));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = gg + 4 * r;
      v[r] = in[i * 16 + cc];
      w[r] = (i == cc) ? 1.0 : 0.0;
    }
    double rowA = bperm_d(v[0], cc), rowW = (cc == 0) ? 1.0 : 0.0;
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t0 = __builtin_readcyclecounter();
    if constexpr (MODE == 0) {
#define F16_S(JJ) f16_step<JJ>(v, w, rowA, rowW, cc, gg);
      F16_S(0) F16_S(1) F16_S(2) F16_S(3) F16_S(4) F16_S(5) F16_S(6) F16_S(7) F16_S(8) F16_S(9) F16_S(10) F16_S(11) F16_S(12) F16_S(13)
      F16_S(14) F16_S(15)
#undef F16_S
    } else {
      double hA = bperm_d(v[0], 16 + cc), hW = (cc == 1) ? 1.0 : 0.0;
      double pRow = 0.0, pt = 0.0, ptw = 0.0;
      double dcur = row_bcast64<0>(rowA), ycur = __builtin_amdgcn_rcp(dcur);
#define F16_S(JJ) f16_step_dpp<JJ>(v, w, rowA, rowW, hA, hW, pRow, pt, ptw, dcur, ycur, cc, gg);
      F16_S(0) F16_S(1) F16_S(2) F16_S(3) F16_S(4) F16_S(5) F16_S(6) F16_S(7) F16_S(8) F16_S(9) F16_S(10) F16_S(11) F16_S(12) F16_S(13)
      F16_S(14) F16_S(15)
#undef F16_S
    }
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
    const unsigned long long t1 = __builtin_readcyclecounter();
    total += t1 - t0;
    sink += v[0] + v[1] + v[2] + v[3] + w[0] + w[1] + w[2] + w[3];
  }
  out[lane] = sink;
  for (int r = 0; r < 4; ++r) { out[64 + (g + 4 * r) * 16 + c] = v[r]; out[64 + 256 + (g + 4 * r) * 16 + c] = w[r]; }
  if (lane == 0) { cyc[0] = total; flag = 1; }
  __threadfence_block();
}

template <int MODE, int NEIGH, int SKIPW = 4, int PRIO = 0>
void run(const char* name, double* out, unsigned long long* cyc, const double* in, double* host) {
  const int reps = 64;
  hipLaunchKernelGGL((probe<MODE, NEIGH, SKIPW, PRIO>), dim3(1), dim3(512), 0, 0, out, cyc, in, reps);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((probe<MODE, NEIGH, SKIPW, PRIO>), dim3(1), dim3(512), 0, 0, out, cyc, in, reps);
  hipDeviceSynchronize();
  unsigned long long h;
  hipMemcpy(&h, cyc, sizeof h, hipMemcpyDeviceToHost);
  hipMemcpy(host, out, (64 + 512) * sizeof(double), hipMemcpyDeviceToHost);
  printf("%-64s %8.1f cycles per 16 pivots = %6.1f per pivot\n", name, (double)h / reps, (double)h / reps / 16);
}

int main() {
  double *out, *in;
  unsigned long long* cyc;
  hipMalloc(&out, (64 + 512) * sizeof(double));
  hipMalloc(&in, 256 * sizeof(double));
  hipMalloc(&cyc, sizeof(unsigned long long));
  double a[256];
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) a[i * 16 + j] = (i == j ? 2.0 : 0.0) + 1.0 / (1.0 + (i > j ? i - j : j - i));
  hipMemcpy(in, a, sizeof a, hipMemcpyHostToDevice);
  static double h0[576], h1[576];
  run<0, 0>("round-3 step, alone", out, cyc, in, h0);
  run<1, 0>("DP-ALU DPP step (2 ahead), alone", out, cyc, in, h1);
  double dv = 0, dw = 0;
  for (int i = 0; i < 256; ++i) { dv = fmax(dv, fabs(h0[64 + i] - h1[64 + i])); dw = fmax(dw, fabs(h0[320 + i] - h1[320 + i])); }
  printf("   max |v3 - v4| = %.3e   max |w3 - w4| = %.3e\n", dv, dw);
  run<0, 1>("round-3 step, 7 polling waves", out, cyc, in, h0);
  run<1, 1>("DP-ALU DPP step (2 ahead), 7 polling waves", out, cyc, in, h1);
  run<0, 2>("round-3 step, 7 waves streaming ds_read + MFMA", out, cyc, in, h0);
  run<1, 2>("DP-ALU DPP step (2 ahead), 7 waves streaming ds_read + MFMA", out, cyc, in, h1);
  run<1, 5>("DPP step, 7 waves MFMA only (register operands)", out, cyc, in, h1);
  run<1, 6>("DPP step, 7 waves LDS reads only", out, cyc, in, h1);
  run<1, 3, 1>("DPP step, streaming waves except wave 1", out, cyc, in, h1);
  run<1, 3, 2>("DPP step, streaming waves except wave 2", out, cyc, in, h1);
  run<1, 3, 3>("DPP step, streaming waves except wave 3", out, cyc, in, h1);
  run<1, 3, 4>("DPP step, streaming waves except wave 4", out, cyc, in, h1);
  run<1, 3, 5>("DPP step, streaming waves except wave 5", out, cyc, in, h1);
  run<1, 4, 4>("DPP step, only wave 4 streams", out, cyc, in, h1);
  run<1, 4, 4, 3>("DPP step at s_setprio 3, only wave 4 streams", out, cyc, in, h1);
  run<1, 2, 4, 3>("DPP step at s_setprio 3, 7 waves streaming ds_read + MFMA", out, cyc, in, h1);
  run<1, 5, 4, 3>("DPP step at s_setprio 3, 7 waves MFMA only", out, cyc, in, h1);
  run<1, 4, 1>("DPP step, only wave 1 streams", out, cyc, in, h1);
  run<1, 4, 2>("DPP step, only wave 2 streams", out, cyc, in, h1);
  return 0;
}
