// Stand-alone consumer of the C ABI (no Python, no torch): builds a synthetic GP block on the device, runs the fused
// NLML + gradients through include/ffgp.h and prints the value, the stage timings and a gradient check against a
// central finite difference of the value w.r.t. one inverse length scale; then the symmetric eigensolver (ffgp_syevd) on a kernel
// matrix of the same points, checked on the host.
//
//   hipcc -O2 --offload-arch=gfx950 -I include examples/nlml_c_abi.cpp -L fidelityfusion_amd -lffgp \
//         -Wl,-rpath,$PWD/fidelityfusion_amd -o /tmp/nlml_c_abi && /tmp/nlml_c_abi 4096 8 2
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ffgp.h"

#define HIPCHK(x)                                                                  \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 2;                                                                    \
    }                                                                              \
  } while (0)

static double lcg(unsigned long long& s) {
  s = s * 6364136223846793005ull + 1442695040888963407ull;
  return (double)(s >> 11) / 9007199254740992.0;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 2048, D = argc > 2 ? atoi(argv[2]) : 8, d = argc > 3 ? atoi(argv[3]) : 1;
  unsigned long long seed = 12345;
  std::vector<double> X((size_t)n * D), Y((size_t)n * d), w(D, 1.0), Wm((size_t)D * d);
  for (auto& v : X) v = lcg(seed);
  for (auto& v : Wm) v = lcg(seed);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < d; ++j) {
      double t = 0.0;
      for (int k = 0; k < D; ++k) t += X[(size_t)i * D + k] * Wm[(size_t)k * d + j];
      Y[(size_t)i * d + j] = sin(6.283185307179586 * t) + 0.1 * (lcg(seed) - 0.5);
    }
  for (int k = 0; k < D; ++k) w[k] = 0.8 + 0.05 * k;
  const double amp = 1.3, dadd = exp(-1.0) + 1e-6;

  double *dX, *dY, *dw, *damp, *ddadd, *dout, *dgw, *dgamp, *dgdadd, *dgY;
  HIPCHK(hipMalloc(&dX, X.size() * 8));
  HIPCHK(hipMalloc(&dY, Y.size() * 8));
  HIPCHK(hipMalloc(&dw, D * 8));
  HIPCHK(hipMalloc(&damp, 8));
  HIPCHK(hipMalloc(&ddadd, 8));
  HIPCHK(hipMalloc(&dout, 8));
  HIPCHK(hipMalloc(&dgw, D * 8));
  HIPCHK(hipMalloc(&dgamp, 8));
  HIPCHK(hipMalloc(&dgdadd, 8));
  HIPCHK(hipMalloc(&dgY, Y.size() * 8));
  HIPCHK(hipMemcpy(dX, X.data(), X.size() * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dY, Y.data(), Y.size() * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(damp, &amp, 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ddadd, &dadd, 8, hipMemcpyHostToDevice));

  ffgp_handle* h = nullptr;
  if (ffgp_create(0, &h) != 0) {
    fprintf(stderr, "ffgp_create failed\n");
    return 3;
  }
  printf("%s\n", ffgp_version());
  ffgp_set_option(h, "timing", 1.0);

  ffgp_problem p = {};
  p.n = n; p.D = D; p.d = d;
  p.X_dev = dX; p.Y_dev = dY; p.w_dev = dw; p.amp_dev = damp;
  p.clamp_min = 1e-30;
  p.diag_add_dev = ddadd;
  p.ll_variant = FFGP_LL_V1;
  p.pi_const = 3.1415;     /* the reference's constant */
  p.kfun = FFGP_KFUN_SE;
  p.kparam = 1.0;
  ffgp_grads g = {};
  g.g_w_dev = dgw; g.g_amp_dev = dgamp; g.g_diag_add_dev = dgdadd; g.g_Y_dev = dgY;

  auto value = [&](const std::vector<double>& ww, ffgp_grads* gp, double* out) -> int {
    if (hipMemcpy(dw, ww.data(), D * 8, hipMemcpyHostToDevice) != hipSuccess) return -100;
    const int rc = ffgp_nlml_fused(h, &p, dout, gp);
    if (rc != 0) return rc;
    return hipMemcpy(out, dout, 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -100;
  };
  double nll = 0.0;
  int rc = value(w, &g, &nll);
  if (rc != 0) {
    fprintf(stderr, "ffgp_nlml_fused returned %d\n", rc);
    return 4;
  }
  std::vector<double> gw(D);
  HIPCHK(hipMemcpy(gw.data(), dgw, D * 8, hipMemcpyDeviceToHost));
  const char* names[16];
  float ms[16];
  int ns = 0;
  ffgp_last_timings(h, ms, names, 16, &ns);
  printf("n=%d D=%d d=%d  nll = %.10f\n", n, D, d, nll);
  for (int i = 0; i < ns; ++i) printf("  stage %-10s %8.3f ms\n", names[i], ms[i]);
  // finite-difference check of d nll / d w[1]
  const int kk = D > 1 ? 1 : 0;
  const double eps = 1e-6;
  std::vector<double> wp = w, wm = w;
  wp[kk] += eps;
  wm[kk] -= eps;
  double vp = 0.0, vm = 0.0;
  if (value(wp, nullptr, &vp) != 0 || value(wm, nullptr, &vm) != 0) return 5;
  const double fd = (vp - vm) / (2 * eps);
  const double relerr = fabs(fd - gw[kk]) / fmax(fabs(fd), 1e-300);
  printf("d nll / d w[%d]: closed form %.8e, finite difference %.8e, rel. diff %.2e\n", kk, gw[kk], fd, relerr);
  if (!(relerr < 1e-5)) {
    fprintf(stderr, "gradient check failed\n");
    ffgp_destroy(h);
    return 6;
  }
  // The eigensolver of the HOGP block through the same ABI: K = exp(-1/2 |x - x'|^2) on the first m points (m not a multiple of 64:
  // the library pads), A = Z diag(W) Z^T checked through trace(K) = sum(W), ||K z_last - w_last z_last|| and ||z_last|| = 1.
  {
    const int m = n < 1000 ? n : 1000;
    std::vector<double> K((size_t)m * m), W(m), Z((size_t)m * m);
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) {
        double s2 = 0.0;
        for (int k = 0; k < D; ++k) {
          const double df = X[(size_t)i * D + k] - X[(size_t)j * D + k];
          s2 += df * df;
        }
        K[(size_t)i * m + j] = exp(-0.5 * s2);
      }
    double *dK, *dWv, *dZ;
    HIPCHK(hipMalloc(&dK, K.size() * 8));
    HIPCHK(hipMalloc(&dWv, m * 8));
    HIPCHK(hipMalloc(&dZ, Z.size() * 8));
    HIPCHK(hipMemcpy(dK, K.data(), K.size() * 8, hipMemcpyHostToDevice));
    rc = ffgp_syevd(h, dK, m, m, dWv, dZ, m);
    if (rc != 0) {
      fprintf(stderr, "ffgp_syevd returned %d\n", rc);
      return 7;
    }
    HIPCHK(hipMemcpy(W.data(), dWv, m * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(Z.data(), dZ, Z.size() * 8, hipMemcpyDeviceToHost));
    double tr = 0.0, sw = 0.0, res = 0.0, nz = 0.0;
    for (int i = 0; i < m; ++i) tr += K[(size_t)i * m + i], sw += W[i];
    for (int i = 0; i < m; ++i) {          // the eigenvector of the largest eigenvalue is the last column
      double s = 0.0;
      for (int j = 0; j < m; ++j) s += K[(size_t)i * m + j] * Z[(size_t)j * m + m - 1];
      const double r = s - W[m - 1] * Z[(size_t)i * m + m - 1];
      res += r * r;
      nz += Z[(size_t)i * m + m - 1] * Z[(size_t)i * m + m - 1];
    }
    printf("ffgp_syevd m=%d: lambda_max %.8f, |trace - sum(W)| %.1e, |K z - w z| %.1e, |z|^2 - 1 %.1e\n", m, W[m - 1], fabs(tr - sw),
           sqrt(res), fabs(nz - 1.0));
    (void)hipFree(dK); (void)hipFree(dWv); (void)hipFree(dZ);
    if (!(fabs(tr - sw) < 1e-9 * tr && sqrt(res) < 1e-10 * W[m - 1] && fabs(nz - 1.0) < 1e-12)) {
      fprintf(stderr, "eigensolver check failed\n");
      return 8;
    }
  }
  ffgp_destroy(h);
  printf("c-abi example ok\n");
  return 0;
}
