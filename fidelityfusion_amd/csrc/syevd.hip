// Symmetric eigensolver: the driver (pad -> dense-to-band -> bulge chasing -> tridiagonal divide & conquer -> two
// back-transformations) and the back-transformation with the stage-1 reflectors.  C ABI: ffgp_syevd and the stage entry points
// (include/ffgp.h).  Replaces `torch.linalg.eigh(K)` = LAPACK / rocSOLVER syevd in the HOGP block of GAR
// (FidelityFusion_Models/two_fidelity_models/hogp_simple.py:15-19,97-100; MFGP_ver2023May/base_gp/hogp.py:20-24).
#include "ffgp_internal.h"
#include "syevd_internal.h"

int ffgp_syevj_small_impl(ffgp_handle* h, const double* M, int n, int ldm, int batch, long strideM, double* Q, int ldq, long strideQ,
                          double* evals, long strideE, int descending);

#define FFGP_SYEVD_MAX_N 32768

int ffgp_ensure_ews(ffgp_handle* h, size_t bytes) {
  if (bytes <= h->ews_bytes) return FFGP_OK;
  if (h->ews) {
    hipStreamSynchronize(h->stream);
    hipFree(h->ews);
    h->ews = nullptr;
    h->ews_bytes = 0;
  }
  const size_t gran = (size_t)64 << 20;
  const size_t want = (bytes + gran - 1) / gran * gran;
  if (hipMalloc(&h->ews, want) != hipSuccess) {
    fprintf(stderr, "[ffgp] eigensolver workspace allocation of %zu bytes failed\n", want);
    return FFGP_ERR_ALLOC;
  }
  h->ews_bytes = want;
  return FFGP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Z <- Q1 Z,  Q1 = prod_p (I - Y_p T_p Y_p^T).  Eight panels at a time form one 256-wide block reflector I - V T V^T whose T follows
// from orthogonality alone:  T^-1 + T^-T = V^T V  =>  T^-1 = striu(V^T V) + 1/2 diag(V^T V); three GEMMs per group.
// ---------------------------------------------------------------------------------------------------------------------
// T (256 x 256) of the wide block reflector from G = V^T V alone: U = striu(G) + diag(G) / 2 is T^-1.  Diagonal 32 x 32 blocks
// by back substitution (one workgroup each), then three doubling levels [A B; 0 C]^-1 = [A^-1, -A^-1 B C^-1; 0, C^-1] as batched
// GEMMs.  Groups narrower than 256 are padded with identity blocks (G arrives zeroed outside its w x w corner).
__global__ __launch_bounds__(256) void q1_diag_inv(const double* __restrict__ G, int w, double* __restrict__ T) {
  __shared__ double Gi[32][33], Tj[32][33];
  const int tid = threadIdx.x, J0 = 32 * blockIdx.x;
  const bool live = J0 < w;
  for (int idx = tid; idx < 1024; idx += 256) Gi[idx >> 5][idx & 31] = live ? G[(size_t)(J0 + (idx >> 5)) * 256 + J0 + (idx & 31)] : 0.0;
  __syncthreads();
  if (tid < 32) {   // column tid of T_jj by back substitution
    const int c = tid;
    for (int i = 31; i >= 0; --i) {
      if (i > c || !live) {
        Tj[i][c] = (i == c) ? 1.0 : 0.0;
        continue;
      }
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = i + 1; k <= c; ++k) s = __builtin_fma(-Gi[i][k], Tj[k][c], s);
      Tj[i][c] = s / (0.5 * Gi[i][i]);
    }
  }
  __syncthreads();
  for (int idx = tid; idx < 1024; idx += 256) T[(size_t)(J0 + (idx >> 5)) * 256 + J0 + (idx & 31)] = Tj[idx >> 5][idx & 31];
}

static int q1_tbuild(ffgp_handle* h, const double* G, int w, double* T, double* tmp) {
  hipStream_t st = h->stream;
  FFGP_HIP(hipMemsetAsync(T, 0, (size_t)256 * 256 * sizeof(double), st));
  hipLaunchKernelGGL(q1_diag_inv, dim3(8), dim3(256), 0, st, G, w, T);
  if (hipGetLastError() != hipSuccess) return FFGP_ERR_HIP;
  for (int s = 32; s < 256; s *= 2) {
    if (s >= w) break;                          // everything beyond is identity padding
    const int np = 256 / (2 * s);
    const long sp = (long)2 * s * 256 + 2 * s;  // from one pair's diagonal position to the next
    // tmp_p = A_p^-1 B_p ;  T[A rows, C cols] = -tmp_p C_p^-1
    FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, T, 256, G + s, 256, tmp, s, s, s, s, 1.0, 0.0, 0, ALIAS_NONE, np, sp, sp,
                                (long)s * s));
    FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, tmp, s, T + (size_t)s * 256 + s, 256, T + s, 256, s, s, s, -1.0, 0.0, 0,
                                ALIAS_NONE, np, (long)s * s, sp, sp));
  }
  return FFGP_OK;
}

#define Q1_AGG 8
size_t ffgp_q1_ws_doubles(int n, int ncols) { return (size_t)3 * 256 * 256 + (size_t)2 * 256 * ncols + 64; }

// trans = 0: Z <- Q1 Z (groups last to first);  trans = 1: Z <- Q1^T Z (groups first to last, every block reflector transposed:
// (I - V T V^T)^T = I - V T^T V^T)
int ffgp_q1_apply_impl(ffgp_handle* h, const double* Y, int ldy, int n, double* Z, int ldz, int ncols, double* ws, int trans) {
  hipStream_t st = h->stream;
  double* Gm = ws;
  double* Tw = Gm + 256 * 256;
  double* tmp = Tw + 256 * 256;
  double* P1 = tmp + 256 * 256;
  double* P2 = P1 + (size_t)256 * ncols;
  const int npan = n / 32 - 1;
  const int ngrp = (npan + Q1_AGG - 1) / Q1_AGG;
  for (int gi = 0; gi < ngrp; ++gi) {
    const int g = trans ? gi : ngrp - 1 - gi;
    const int p0 = g * Q1_AGG, p1 = min(npan, p0 + Q1_AGG);
    const int w = 32 * (p1 - p0);
    const int r0 = 32 * p0 + 32, mg = n - r0;
    const double* V = Y + (size_t)r0 * ldy + 32 * p0;
    if (w < 256) FFGP_HIP(hipMemsetAsync(Gm, 0, (size_t)256 * 256 * sizeof(double), st));
    FFGP_CHECK(ffgp_gemm_launch(h, OP_MNMAJOR, OP_MNMAJOR, TILES_FULL, 0, V, ldy, V, ldy, Gm, 256, w, w, mg, 1.0, 0.0));
    FFGP_CHECK(q1_tbuild(h, Gm, w, Tw, tmp));
    double* Zr = Z + (size_t)r0 * ldz;
    FFGP_CHECK(ffgp_gemm_launch(h, OP_MNMAJOR, OP_MNMAJOR, TILES_FULL, 0, V, ldy, Zr, ldz, P1, ncols, w, ncols, mg, 1.0, 0.0));
    FFGP_CHECK(ffgp_gemm_launch(h, trans ? OP_MNMAJOR : OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, Tw, 256, P1, ncols, P2, ncols, w, ncols, w, 1.0, 0.0));
    FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, V, ldy, P2, ncols, Zr, ldz, mg, ncols, w, -1.0, 1.0));
  }
  return FFGP_OK;
}

__global__ __launch_bounds__(256) void syevd_identity(double* __restrict__ M, int n) {
  const int r = blockIdx.x;
  for (int j = threadIdx.x; j < n; j += 256) M[(size_t)r * n + j] = (j == r) ? 1.0 : 0.0;
}

// ---------------------------------------------------------------------------------------------------------------------
// padding to a multiple of 64: decoupled diagonal entries above the spectrum (Gershgorin) -- every reflector component on a
// padded row is exactly zero, so they never mix with the problem and their eigenpairs come out last
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void syevd_rowsum(const double* __restrict__ A, int n, int lda, double* __restrict__ rs) {
  __shared__ double red[256];
  const int r = blockIdx.x;
  double s = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) s += fabs(A[(size_t)r * lda + j]);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) rs[r] = red[0];
}
__global__ __launch_bounds__(256) void syevd_max(const double* __restrict__ rs, int n, double* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) s = fmax(s, rs[j]);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = 2.0 * red[0] + 1.0;
}
__global__ __launch_bounds__(256) void syevd_pad(const double* __restrict__ A, int n, int lda, double* __restrict__ Ap, int np,
                                                 const double* __restrict__ big) {
  const int r = blockIdx.x;
  for (int j = threadIdx.x; j < np; j += 256) {
    double v = 0.0;
    if (r < n && j < n) v = (j <= r) ? A[(size_t)r * lda + j] : A[(size_t)j * lda + r];   // the lower triangle, mirrored
    else if (r == j) v = big[0] * (1.0 + (double)(r - n) / 64.0);
    Ap[(size_t)r * np + j] = v;
  }
}

struct SyevdPlan {
  int np, K;
  double *Ap, *AB, *Y, *Tpan, *ws1, *d, *e, *V2, *tau2, *blocks, *lam, *Zt, *ws3, *ws4, *scal;
  int* prog;
  size_t total;
};

static SyevdPlan syevd_plan(double* base, int n) {
  SyevdPlan s;
  s.np = (n + 63) / 64 * 64;
  const size_t np = s.np;
  s.K = s.np / 32 + 1;
  size_t o = 0;
  auto take = [&](size_t cnt) {
    double* p = base ? base + o : nullptr;
    o += (cnt + 63) / 64 * 64;
    return p;
  };
  s.Ap = take(np * np);
  s.AB = take(np * SB_LDB);
  s.Y = take(np * np);
  s.Tpan = take((np / 32) * 1024);
  s.ws1 = take(ffgp_sy2sb_ws_doubles(s.np));
  s.d = take(np);
  s.e = take(np);
  s.V2 = take(np * s.K * 32);
  s.tau2 = take(np * s.K);
  s.prog = reinterpret_cast<int*>(take(np / 2 + 64));
  s.blocks = take(ffgp_q2_block_doubles(s.np));
  s.lam = take(np);
  s.Zt = take(np * np);
  s.ws3 = take(ffgp_stedc_ws_doubles(s.np));
  s.ws4 = take(ffgp_q1_ws_doubles(s.np, s.np));
  s.scal = take(np + 64);
  s.total = o;
  return s;
}

static int syevd_check_watchdog(ffgp_handle* h, const int* prog, int np) {
  FFGP_HIP(hipMemcpyAsync(h->h_info, prog + np, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  FFGP_HIP(hipStreamSynchronize(h->stream));
  if (h->h_info[0] != 0) {
    fprintf(stderr, "[ffgp] sb2st: a sweep waited too long for its predecessor (watchdog)\n");
    return FFGP_ERR_HIP;
  }
  return FFGP_OK;
}

extern "C" int ffgp_syevd(ffgp_handle* h, const double* A_dev, int n, int lda, double* W_dev, double* Z_dev, int ldz) {
  if (!h || !A_dev || !W_dev || !Z_dev || n < 1 || lda < n || ldz < n) return FFGP_ERR_ARG;
  if (n > FFGP_SYEVD_MAX_N) return FFGP_ERR_ARG;   // (workspace ~ 10 n^2 doubles: 86 GB at the limit)
  FFGP_HIP(hipSetDevice(h->device));
  hipStream_t st = h->stream;
  if (n <= 64) {   // one LDS Jacobi problem
    return ffgp_syevj_small_impl(h, A_dev, n, lda, 1, 0, Z_dev, ldz, 0, W_dev, 0, 0);
  }
  SyevdPlan s = syevd_plan(nullptr, n);
  FFGP_CHECK(ffgp_ensure_ews(h, s.total * sizeof(double)));
  s = syevd_plan(h->ews, n);
  const int np = s.np;
  hipLaunchKernelGGL(syevd_rowsum, dim3(n), dim3(256), 0, st, A_dev, n, lda, s.scal + 64);
  hipLaunchKernelGGL(syevd_max, dim3(1), dim3(256), 0, st, s.scal + 64, n, s.scal);
  hipLaunchKernelGGL(syevd_pad, dim3(np), dim3(256), 0, st, A_dev, n, lda, s.Ap, np, s.scal);
  if (hipGetLastError() != hipSuccess) return FFGP_ERR_HIP;
  FFGP_CHECK(ffgp_sy2sb_impl(h, s.Ap, np, np, s.AB, s.Y, np, s.Tpan, s.ws1));
  const int ncols = (n + 31) / 32 * 32;   // the padding's eigenpairs are the last columns: never transformed
  const int ngroups = np / 32;
  if (!h->eig_overlap) {
    // one stage after the other on the handle's stream
    FFGP_CHECK(ffgp_sb2st_impl(h, s.AB, np, s.d, s.e, s.V2, s.tau2, s.prog));
    FFGP_CHECK(ffgp_stedc_impl(h, s.d, s.e, np, s.lam, s.Zt, np, s.ws3));
    FFGP_CHECK(ffgp_q2_prep_impl(h, s.V2, s.tau2, np, s.blocks, 0, ngroups, 0));
    FFGP_CHECK(ffgp_q2_apply_impl(h, s.blocks, np, s.Zt, np, ncols, 0, ngroups, 0, 0));
    FFGP_CHECK(ffgp_q1_apply_impl(h, s.Y, np, np, s.Zt, np, ncols, s.ws4, 0));
    FFGP_HIP(hipMemcpy2DAsync(Z_dev, (size_t)ldz * sizeof(double), s.Zt, (size_t)np * sizeof(double), (size_t)n * sizeof(double), n,
                              hipMemcpyDeviceToDevice, st));
  } else {
    // Two queues side by side.  The bulge chase (latency-bound, all of it on one XCD) runs on the handle's side stream, cut into a
    // few launches; behind every launch the main stream prepares and applies that launch's reflectors -- in the order they were
    // produced, which is the order of Q2^T -- to M, which starts as the identity; those kernels keep off the chase's XCD.  When
    // the chase ends M = Q2^T is complete up to the last launch's blocks; then the tridiagonal problem, ONE GEMM
    // Z_B = Q2 Z_T = M^T Z_T, and the stage-1 back-transformation.
    hipStream_t side = h->aux;
    double* M = s.Ap;                       // the dense matrix is no longer needed
    double* ZB = s.blocks;                  // (free again once the last blocks have been applied)
    if (!h->eig_ev[0])
      for (int i = 0; i < 12; ++i) FFGP_HIP(hipEventCreateWithFlags(&h->eig_ev[i], hipEventDisableTiming));
    FFGP_HIP(hipEventRecord(h->eig_ev[0], st));
    FFGP_HIP(hipStreamWaitEvent(side, h->eig_ev[0], 0));
    FFGP_CHECK(ffgp_sb2st_init(h, side, np, s.V2, s.tau2, s.prog));
    const int gchunk = max(4, (ngroups + 7) / 8);
    const int nchunk = (ngroups + gchunk - 1) / gchunk;     // <= 8
    for (int c = 0; c < nchunk; ++c) {
      FFGP_CHECK(ffgp_sb2st_chunk(h, side, s.AB, np, s.d, s.e, s.V2, s.tau2, s.prog, 32 * c * gchunk, 32 * (c + 1) * gchunk));
      FFGP_HIP(hipEventRecord(h->eig_ev[1 + c], side));
    }
    FFGP_CHECK(ffgp_sb2st_finish(h, side, s.AB, np, s.d, s.e));
    FFGP_HIP(hipEventRecord(h->eig_ev[10], side));
    hipLaunchKernelGGL(syevd_identity, dim3(np), dim3(256), 0, st, M, np);
    for (int c = 0; c < nchunk; ++c) {
      FFGP_HIP(hipStreamWaitEvent(st, h->eig_ev[1 + c], 0));
      const int G0 = c * gchunk, G1 = min(ngroups, (c + 1) * gchunk);
      FFGP_CHECK(ffgp_q2_prep_impl(h, s.V2, s.tau2, np, s.blocks, G0, G1, 1));
      FFGP_CHECK(ffgp_q2_apply_impl(h, s.blocks, np, M, np, np, G0, G1, 1, c + 1 < nchunk ? 1 : 0));   // M <- (Q2 part)^T M
    }
    FFGP_HIP(hipStreamWaitEvent(st, h->eig_ev[10], 0));
    FFGP_CHECK(ffgp_stedc_impl(h, s.d, s.e, np, s.lam, s.Zt, np, s.ws3));
    FFGP_CHECK(ffgp_gemm_launch(h, OP_MNMAJOR, OP_MNMAJOR, TILES_FULL, 0, M, np, s.Zt, np, ZB, np, np, ncols, np, 1.0, 0.0));
    FFGP_CHECK(ffgp_q1_apply_impl(h, s.Y, np, np, ZB, np, ncols, s.ws4, 0));
    FFGP_HIP(hipMemcpy2DAsync(Z_dev, (size_t)ldz * sizeof(double), ZB, (size_t)np * sizeof(double), (size_t)n * sizeof(double), n,
                              hipMemcpyDeviceToDevice, st));
  }
  FFGP_HIP(hipMemcpyAsync(W_dev, s.lam, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
  return syevd_check_watchdog(h, s.prog, np);
}

// ---- stage entry points (n a multiple of 64, 64 <= n <= 32768; all buffers caller-owned device memory) ----------------------
extern "C" int ffgp_sy2sb(ffgp_handle* h, double* A_dev, int n, int lda, double* AB_dev, double* Y_dev, int ldy) {
  if (!h || !A_dev || !AB_dev || !Y_dev || n % 64 || n < 64 || n > FFGP_SYEVD_MAX_N || lda < n || ldy < n) return FFGP_ERR_ARG;
  FFGP_HIP(hipSetDevice(h->device));
  const size_t need = ffgp_sy2sb_ws_doubles(n) + (size_t)(n / 32) * 1024;
  FFGP_CHECK(ffgp_ensure_ews(h, need * sizeof(double)));
  return ffgp_sy2sb_impl(h, A_dev, n, lda, AB_dev, Y_dev, ldy, h->ews + ffgp_sy2sb_ws_doubles(n), h->ews);
}

extern "C" long ffgp_sb2st_reflector_doubles(int n) { return (long)n * (n / 32 + 1) * 33; }

extern "C" int ffgp_sb2st(ffgp_handle* h, double* AB_dev, int n, double* d_dev, double* e_dev, double* refl_dev) {
  if (!h || !AB_dev || !d_dev || !e_dev || !refl_dev || n % 64 || n < 64) return FFGP_ERR_ARG;
  FFGP_HIP(hipSetDevice(h->device));
  const int K = n / 32 + 1;
  FFGP_CHECK(ffgp_ensure_ews(h, ((size_t)n + 64) * sizeof(int)));
  int* prog = reinterpret_cast<int*>(h->ews);
  FFGP_CHECK(ffgp_sb2st_impl(h, AB_dev, n, d_dev, e_dev, refl_dev, refl_dev + (size_t)n * K * 32, prog));
  return syevd_check_watchdog(h, prog, n);
}

extern "C" int ffgp_stedc(ffgp_handle* h, const double* d_dev, const double* e_dev, int n, double* W_dev, double* Z_dev, int ldz) {
  if (!h || !d_dev || !e_dev || !W_dev || !Z_dev || n % 64 || n < 64 || ldz < n) return FFGP_ERR_ARG;
  FFGP_HIP(hipSetDevice(h->device));
  FFGP_CHECK(ffgp_ensure_ews(h, ffgp_stedc_ws_doubles(n) * sizeof(double)));
  return ffgp_stedc_impl(h, d_dev, e_dev, n, W_dev, Z_dev, ldz, h->ews);
}

extern "C" int ffgp_ormq2(ffgp_handle* h, const double* refl_dev, int n, double* Z_dev, int ldz, int ncols) {
  if (!h || !refl_dev || !Z_dev || n % 64 || n < 64 || ldz < ncols || ncols < 1) return FFGP_ERR_ARG;
  FFGP_HIP(hipSetDevice(h->device));
  const int K = n / 32 + 1;
  FFGP_CHECK(ffgp_ensure_ews(h, ffgp_q2_block_doubles(n) * sizeof(double)));
  FFGP_CHECK(ffgp_q2_prep_impl(h, refl_dev, refl_dev + (size_t)n * K * 32, n, h->ews, 0, n / 32, 0));
  return ffgp_q2_apply_impl(h, h->ews, n, Z_dev, ldz, ncols, 0, n / 32, 0, 0);
}

extern "C" int ffgp_ormq1(ffgp_handle* h, const double* Y_dev, int ldy, int n, double* Z_dev, int ldz, int ncols) {
  if (!h || !Y_dev || !Z_dev || n % 64 || n < 64 || ldz < ncols || ncols < 1 || ldy < n) return FFGP_ERR_ARG;
  FFGP_HIP(hipSetDevice(h->device));
  FFGP_CHECK(ffgp_ensure_ews(h, ffgp_q1_ws_doubles(n, ncols) * sizeof(double)));
  return ffgp_q1_apply_impl(h, Y_dev, ldy, n, Z_dev, ldz, ncols, h->ews, 0);
}
