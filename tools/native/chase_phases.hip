// Where one step of the bulge chase spends its time (development probe): the library's own kernel, compiled here with 100 MHz clock
// stamps of lane 0 in two consecutive steps of one steady-state sweep (FFGP_CH_STAMPS in csrc/sb2st.hip), on a random band matrix.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ifidelityfusion_amd/csrc tools/native/chase_phases.hip \
//         -Lfidelityfusion_amd -lffgp -Wl,-rpath,$PWD/fidelityfusion_amd -o chase_phases && ./chase_phases
#define FFGP_CH_STAMPS 1
#define FFGP_CH_STAMP_S 2000
#define FFGP_CH_STAMP_K 20
// (own names for what is launched here: the library linked below exports the unstamped kernel under the original name)
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
#include <vector>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("%s: %s\n", #x, hipGetErrorString(e_));                                \
      return 1;                                                                     \
    }                                                                               \
  } while (0)

int main() {
  const int n = 4096, K = n / 32 + 1;
  std::vector<double> hAB((size_t)n * SB_LDB, 0.0);
  srand(2);
  for (int c = 0; c < n; ++c)
    for (int k = 0; k <= 32 && c + k < n; ++k) hAB[(size_t)c * SB_LDB + k] = (double)rand() / RAND_MAX - 0.5 + (k == 0 ? 4.0 : 0.0);
  ffgp_handle* h = nullptr;
  if (ffgp_create(0, &h) != FFGP_OK) { printf("ffgp_create failed\n"); return 1; }
  double *AB, *d, *e, *V2, *tau2;
  int* prog;
  CK(hipMalloc(&AB, hAB.size() * sizeof(double)));
  CK(hipMalloc(&d, n * sizeof(double)));
  CK(hipMalloc(&e, n * sizeof(double)));
  CK(hipMalloc(&V2, (size_t)n * K * 32 * sizeof(double)));
  CK(hipMalloc(&tau2, (size_t)n * K * sizeof(double)));
  CK(hipMalloc(&prog, (size_t)(n + 64) * sizeof(int)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemcpy(AB, hAB.data(), hAB.size() * sizeof(double), hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, h->stream));
    if (probe_sb2st_impl(h, AB, n, d, e, V2, tau2, prog) != FFGP_OK) { printf("chase failed\n"); return 1; }
    CK(hipEventRecord(e1, h->stream));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  int err = 0;
  CK(hipMemcpy(&err, prog + n, sizeof(int), hipMemcpyDeviceToHost));
  printf("chase n = %d: %.2f ms (best of 3; incl. the stores' memsets), watchdog word %d -> %.2f us per sweep\n", n, best, err, best * 1e3 / n);
  unsigned long long st[16];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(ffgp_ch_stamp), sizeof(st)));
  auto us = [&](int a, int b) { return (double)(st[b] - st[a]) * 0.01; };
  for (int q = 0; q < 2; ++q) {
    const int o = 8 * q;
    printf("sweep %d step %d:  blocks arrive %.2f  diagonal block %.2f  block below %.2f  stores complete %.2f  publish %.2f  | step %.2f us", FFGP_CH_STAMP_S,
           FFGP_CH_STAMP_K + q, us(o + 0, o + 1), us(o + 1, o + 2), us(o + 2, o + 3), us(o + 3, o + 4), us(o + 4, o + 5), us(o + 0, o + 5));
    if (q == 0) printf("  then waits %.2f us for the predecessor", us(5, 8));
    printf("\n");
  }
  return 0;
}
