// Where one time step of q2_apply_wave4 (Z <- Q2 Z, four sweep groups per pass) spends its time (development probe): the library's own
// kernel, compiled here with shader-clock stamps of lane 0 of one wave in two consecutive steps of one pass (FFGP_Q2_STAMPS in
// csrc/sb2st.hip).  Operand values do not matter for the timing: zeros.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ifidelityfusion_amd/csrc tools/native/q2_phases.hip \
//         -Lfidelityfusion_amd -lffgp -Wl,-rpath,$PWD/fidelityfusion_amd -o q2_phases && ./q2_phases [wave]
#define FFGP_Q2_STAMPS 1
#define FFGP_Q2_STAMP_B 100
#define FFGP_Q2_STAMP_PASS 20
#define FFGP_Q2_STAMP_T 60
#ifndef FFGP_Q2_STAMP_W
#define FFGP_Q2_STAMP_W 0
#endif
#define sb2st_chase probe_sb2st_chase
#define sb2st_tail probe_sb2st_tail
#define ffgp_sb2st_impl probe_sb2st_impl
#define ffgp_sb2st_init probe_sb2st_init
#define ffgp_sb2st_chunk probe_sb2st_chunk
#define ffgp_sb2st_finish probe_sb2st_finish
#define ffgp_q2_prep_impl probe_q2_prep_impl
#define ffgp_q2_apply_impl probe_q2_apply_impl
#define ffgp_q2_block_doubles probe_q2_block_doubles
#define q2_prep probe_q2_prep
#define q2_apply probe_q2_apply
#define q2_apply_wave4 probe_q2_apply_wave4
#include "../../fidelityfusion_amd/csrc/sb2st.hip"
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                                \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

int main() {
  const int n = 8192;
  ffgp_handle* h = nullptr;
  if (ffgp_create(0, &h) != FFGP_OK) { printf("ffgp_create failed\n"); return 1; }
  double *blocks, *Z;
  const size_t nb = probe_q2_block_doubles(n);
  CK(hipMalloc(&blocks, nb * sizeof(double)));
  CK(hipMalloc(&Z, (size_t)n * n * sizeof(double)));
  CK(hipMemset(blocks, 0, nb * sizeof(double)));
  CK(hipMemset(Z, 0, (size_t)n * n * sizeof(double)));
  h->q2_blocks_lanes = 1;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, h->stream));
    if (probe_q2_apply_impl(h, blocks, n, Z, n, n, 0, n / 32, 0, 0) != FFGP_OK) { printf("apply failed\n"); return 1; }
    CK(hipEventRecord(e1, h->stream));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  printf("q2_apply_wave4 n = %d: %.2f ms (best of 3)\n", n, best);
  unsigned long long st[32];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(ffgp_q2_stamp), sizeof(st)));
  const double mhz = 2400.0;   // shader clock (tools/diag_trace.py measures 2401 MHz under load)
  auto us = [&](int a, int b) { return (double)(st[b] - st[a]) / mhz; };
  for (int q = 0; q < 2; ++q) {
    const int o = 16 * q;
    printf("wave %d, step t = %d: band load issued + barrier 1 %.2f | X = V^T Z %.2f | X through LDS %.2f | Z -= W X %.2f | barrier 2 %.2f | band out / in %.2f | step %.2f us\n",
           FFGP_Q2_STAMP_W, FFGP_Q2_STAMP_T + q, us(o + 0, o + 1), us(o + 1, o + 2), us(o + 2, o + 3), us(o + 3, o + 4), us(o + 4, o + 5), us(o + 5, o + 6),
           us(o + 0, o + 6));
  }
  printf("step to step: %.2f us\n", us(0, 16));
  return 0;
}
