// Where the time of the band reduction's leaf QR goes (development probe): the library's own kernel, compiled here with 100 MHz clock
// stamps of workgroup 0 / thread 0 (FFGP_QR_STAMPS in csrc/sy2sb.hip), on a random 8192 x 32 panel with the matrix's row stride.
// Prints the phases of the kernel, the duration of every block of four columns and the steps inside column 16.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ifidelityfusion_amd/csrc tools/native/qr_phases.hip \
//         -Lfidelityfusion_amd -lffgp -Wl,-rpath,$PWD/fidelityfusion_amd -o qr_phases && ./qr_phases
#define FFGP_QR_STAMPS 1
// (own names for the two kernels launched here: the library linked below exports the unstamped ones under the original names, and
//  with equal names the loader binds both registrations to one handle -- the library's code object then runs)
#define sy2sb_leaf_qr4 probe_leaf_qr4
#define sy2sb_leaf_qr probe_leaf_qr
#define sy2sb_top probe_top
#include "../../fidelityfusion_amd/csrc/sy2sb.hip"
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
  const int n = 8192, m = 8160, L = (m + QR_ROWS - 1) / QR_ROWS;
  std::vector<double> hA((size_t)m * 32);
  srand(1);
  for (auto& v : hA) v = (double)rand() / RAND_MAX - 0.5;
  double *A, *Rst, *Tst;
  CK(hipMalloc(&A, (size_t)m * n * sizeof(double)));
  CK(hipMalloc(&Rst, (size_t)L * 1024 * sizeof(double)));
  CK(hipMalloc(&Tst, (size_t)L * 1024 * sizeof(double)));
  LeafArgs la;
  la.A = A; la.lda = n; la.m = m; la.Rst = Rst; la.Tst = Tst;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int which = 0; which < 2; ++which) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemcpy2D(A, (size_t)n * sizeof(double), hA.data(), 32 * sizeof(double), 32 * sizeof(double), m, hipMemcpyHostToDevice));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      if (which) hipLaunchKernelGGL(sy2sb_leaf_qr4, dim3(L), dim3(QR4_THREADS), 0, 0, la);
      else hipLaunchKernelGGL(sy2sb_leaf_qr, dim3(L), dim3(QR_THREADS), 0, 0, la);
      CK(hipGetLastError());
      CK(hipEventRecord(e1, 0));
      CK(hipDeviceSynchronize());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("%s: %.1f us (events, best of 5, %d leaves)\n", which ? "sy2sb_leaf_qr4 (256 threads)" : "sy2sb_leaf_qr (1024 threads)", best * 1e3, L);
  }
  unsigned long long st[64];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(ffgp_qr_stamp), sizeof(st)));
  printf("raw stamps 0..3: %llu %llu %llu %llu\n", st[0], st[1], st[2], st[3]);
  auto us = [&](int a, int b) { return (double)(st[b] - st[a]) * 0.01; };
  printf("leaf_qr4 workgroup 0:  load %.2f  QR %.2f  store %.2f us\n", us(0, 1), us(1, 2), us(2, 3));
  printf("  columns in blocks of four:");
  for (int jb = 0; jb < 8; ++jb) printf(" %.2f", us(8 + jb, jb < 7 ? 9 + jb : 16));
  printf(" us;  T build %.2f us\n", us(16, 2));
  printf("  column 16:  LDS reads %.2f  dots %.2f  sums %.2f  scalars %.2f  updates %.2f  publish %.2f  barrier %.2f us\n", us(20, 21), us(21, 22),
         us(22, 23), us(23, 24), us(24, 25), us(25, 26), us(26, 27));
  // the top kernel of the same panel: QR of the 16 stacked R factors, then the reconstruction algebra (one workgroup)
  double *Vtst, *small, *Tpan, *Yp, *ABp;
  CK(hipMalloc(&Vtst, 512 * 32 * sizeof(double)));
  CK(hipMalloc(&small, 4096 * sizeof(double)));
  CK(hipMalloc(&Tpan, 1024 * sizeof(double)));
  CK(hipMalloc(&Yp, (size_t)32 * 32 * sizeof(double)));
  CK(hipMalloc(&ABp, (size_t)32 * SB_LDB * sizeof(double)));
  CK(hipMemcpy2D(A, (size_t)n * sizeof(double), hA.data(), 32 * sizeof(double), 32 * sizeof(double), m, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(probe_leaf_qr, dim3(L), dim3(QR_THREADS), 0, 0, la);
  TopArgs ta;
  ta.A = A; ta.lda = n; ta.m = m; ta.L = L; ta.Rst = Rst; ta.Tst = Tst; ta.Vtst = Vtst; ta.small = small;
  ta.Vmid0 = Rst; ta.Tmid0 = Tst; ta.three = 0; ta.Tpan = Tpan; ta.Y = Yp; ta.ldy = 32; ta.AB = ABp; ta.use_tree = 1;
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(probe_top, dim3(1), dim3(QR_THREADS), 0, 0, ta);
  CK(hipGetLastError());
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float tms;
  CK(hipEventElapsedTime(&tms, e0, e1));
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(ffgp_qr_stamp), sizeof(st)));
  printf("sy2sb_top: %.1f us (events).  load %.2f  QR of the R stack %.2f  V / Xt / Q0 %.2f  X0, W, top block of Q1 %.2f  modified LU %.2f  "
         "U, T, U^-1 %.2f  stores %.2f us\n", tms * 1e3, us(30, 31), us(31, 32), us(32, 33), us(33, 34), us(34, 35), us(35, 36), us(36, 37));
  return 0;
}
