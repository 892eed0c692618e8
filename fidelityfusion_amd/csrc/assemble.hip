// Covariance assembly for gfx950:  K_ij = amp * exp(-1/2 * max(sum_k ((x_ik - x_jk) w_k)^2, clamp))  (+ Sigma extras).
//
// One pass, write-only over K: the reference's torch path makes >= 6 N x N passes (scaled copies, norms, N x N
// matmul, sqrt, square, exp, mul, plus 2-3 torch.eye temporaries for the diagonal adds -- GaussianProcess/kernel.py:
// 100-105, cigp_v10.py:57-60).  64x64 output tile per 256-thread workgroup, 4x4 outputs per thread; the two
// scaled input panels are staged through LDS in 16-dimension chunks (x1 as [row][17], x2 transposed [dim][64], both
// conflict-free), every output row is written as 128-byte segments.  The diagonal / full-matrix / all-entries
// adds of the four Sigma conventions (S1-S4) are fused into the epilogue; the mean(K) jitter of
// gp_computation_pack.negative_log_likelihood is a tile-sum atomics + one tiny follow-up kernel.
#include "ffgp_internal.h"

#define AT 64   // tile edge
#define DC 16   // dimension chunk

struct AsmArgs {
  const double* X1; int n1;
  const double* X2; int n2;
  int D;
  const double* w; const double* amp; double clamp;
  const double* diag_add; const double* diag_vec; long diag_stride;
  const double* add_mat; int ld_add; double add_all;
  double* K; int ldk; int lower_only; int symmetric;
  double* ksum;   // nullable: accumulates sum(K) over the full n1 x n2 matrix
  int tiles_n;
  int kfun; double rinv;   // radial profile and 1/rho
};

__global__ __launch_bounds__(256) void ffgp_assemble_kernel(AsmArgs a) {
  __shared__ double x1s[AT][DC + 1];
  __shared__ double x2t[DC][AT + 1];
  __shared__ double red[4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  int ti, tj;
  if (a.lower_only) {
    const int t = blockIdx.x;
    int r = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= t) ++r;
    while (r * (r + 1) / 2 > t) --r;
    ti = r;
    tj = t - r * (r + 1) / 2;
  } else {
    ti = blockIdx.x / a.tiles_n;
    tj = blockIdx.x % a.tiles_n;
  }
  const int r0 = ti * AT, c0 = tj * AT;

  double sq[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) sq[i][j] = 0.0;

  for (int d0 = 0; d0 < a.D; d0 += DC) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 256 * q;
      const int row = idx >> 4, dd = idx & 15;
      const int gd = d0 + dd;
      // (unconditional loads from clamped addresses, masked afterwards: guarded loads serialise on memory latency)
      const int gdc = min(gd, a.D - 1);
      const double wk = a.w[gdc];
      const int g1 = r0 + row, g2 = c0 + row;
      const double l1 = a.X1[(size_t)min(g1, a.n1 - 1) * a.D + gdc], l2 = a.X2[(size_t)min(g2, a.n2 - 1) * a.D + gdc];
      x1s[row][dd] = (gd < a.D && g1 < a.n1) ? l1 * wk : 0.0;
      x2t[dd][row] = (gd < a.D && g2 < a.n2) ? l2 * wk : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int dd = 0; dd < DC; ++dd) {
      double p[4], q2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) p[i] = x1s[ty + 16 * i][dd];
#pragma unroll
      for (int j = 0; j < 4; ++j) q2[j] = x2t[dd][tx + 16 * j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const double df = p[i] - q2[j];
          sq[i][j] = __builtin_fma(df, df, sq[i][j]);
        }
    }
    __syncthreads();
  }

  const double amp = a.amp[0];
  const double dadd = a.diag_add ? a.diag_add[0] : 0.0;
  ExpCoef ec;
  ffgp_exp_load(ec);
  const bool se = (a.kfun == FFGP_KFUN_SE);
  double tsum = 0.0;
  // interior tiles of the squared-exponential profile (all 64 x 64 entries exist, none on the diagonal, no matrix add): no
  // per-entry bounds / diagonal / triangle selects -- the kernel is bound by vector-instruction issue, every one counts
  if (se && r0 + AT <= a.n1 && c0 + AT <= a.n2 && !(a.symmetric && ti == tj) && !a.add_mat) {
    const double addall = a.symmetric ? a.add_all : 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double* __restrict__ dst = a.K + (size_t)(r0 + ty + 16 * i) * a.ldk + c0 + tx;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double k = amp * ffgp_exp_fast(-0.5 * fmax(sq[i][j], a.clamp), ec);
        tsum += k;
        dst[16 * j] = k + addall;
      }
    }
  } else
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = r0 + ty + 16 * i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = c0 + tx + 16 * j;
      if (row < a.n1 && col < a.n2) {
        const double s = fmax(sq[i][j], a.clamp);
        double k = amp * (se ? ffgp_exp_fast(-0.5 * s, ec) : ffgp_kfun_val(a.kfun, a.rinv, s));
        tsum += k;
        if (a.symmetric) {
          if (row == col) {
            k += dadd;
            if (a.diag_vec) k += a.diag_vec[(size_t)row * a.diag_stride];
          }
          if (a.add_mat) k += (row >= col) ? a.add_mat[(size_t)row * a.ld_add + col] : a.add_mat[(size_t)col * a.ld_add + row];
          k += a.add_all;
        }
        if (!a.lower_only || col <= row) a.K[(size_t)row * a.ldk + col] = k;
      }
    }
  }
  if (a.ksum) {
    // full-matrix sum: an off-diagonal tile of the lower-only sweep stands for its mirror image too
    if (a.lower_only && ti != tj) tsum *= 2.0;
    for (int o = 32; o > 0; o >>= 1) tsum += __shfl_down(tsum, o);
    if ((tid & 63) == 0) red[tid >> 6] = tsum;
    __syncthreads();
    if (tid == 0) atomicAdd(a.ksum, red[0] + red[1] + red[2] + red[3]);
  }
}

__global__ void ffgp_zero_scalar(double* p) { p[0] = 0.0; }

__global__ void ffgp_mean_jitter_kernel(double* K, int ldk, int n, const double* ksum, double coef) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) K[(size_t)i * ldk + i] += coef * ksum[0] / ((double)n * (double)n);
}

// dst[c, r] = src[r, c]  (src rows x cols, row-major) -- used to park Y^T / K_*^T as passenger rows and to
// transpose results back
__global__ __launch_bounds__(256) void ffgp_transpose_kernel(const double* __restrict__ src, int rows, int cols, int lds_,
                                                             double* __restrict__ dst, int ldd, double scale) {
  __shared__ double t[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;  // bx over cols, by over rows
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = by + i, c = bx + tx;
    t[i][tx] = (r < rows && c < cols) ? src[(size_t)r * lds_ + c] : 0.0;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = bx + i, r = by + tx;
    if (c < cols && r < rows) dst[(size_t)c * ldd + r] = scale * t[tx][i];
  }
}

int ffgp_transpose(ffgp_handle* h, const double* src, int rows, int cols, int ld_src, double* dst, int ld_dst, double scale) {
  if (rows <= 0 || cols <= 0) return FFGP_OK;
  dim3 grid((cols + 31) / 32, (rows + 31) / 32);
  hipLaunchKernelGGL(ffgp_transpose_kernel, grid, dim3(256), 0, h->stream, src, rows, cols, ld_src, dst, ld_dst, scale);
  return FFGP_OK;
}

int ffgp_assemble_impl(ffgp_handle* h, const double* X1, int n1, const double* X2, int n2, int D, const double* w,
                       const double* amp, double clamp_min, const double* diag_add, const double* diag_vec,
                       long diag_stride, const double* add_mat, int ld_add, double add_all, double mean_jitter, double* K,
                       int ldk, int lower_only, int kfun, double kparam) {
  if (n1 <= 0 || n2 <= 0) return FFGP_OK;
  if (!X1 || !X2 || !w || !amp || !K || D <= 0 || ldk < n2) return FFGP_ERR_ARG;
  const bool symmetric = (X1 == X2 && n1 == n2);
  const bool extras = diag_add || diag_vec || add_mat || add_all != 0.0 || mean_jitter != 0.0;
  if (extras && !symmetric) return FFGP_ERR_ARG;
  if (lower_only && !symmetric) return FFGP_ERR_ARG;
  AsmArgs a;
  a.X1 = X1; a.n1 = n1; a.X2 = X2; a.n2 = n2; a.D = D;
  a.w = w; a.amp = amp; a.clamp = clamp_min;
  a.diag_add = diag_add; a.diag_vec = diag_vec; a.diag_stride = diag_stride;
  a.add_mat = add_mat; a.ld_add = ld_add; a.add_all = add_all;
  a.K = K; a.ldk = ldk; a.lower_only = lower_only ? 1 : 0; a.symmetric = symmetric ? 1 : 0;
  a.ksum = nullptr;
  a.kfun = kfun;
  a.rinv = (kparam != 0.0) ? 1.0 / kparam : 1.0;
  if (mean_jitter != 0.0) {
    a.ksum = h->d_scal + 32;
    hipLaunchKernelGGL(ffgp_zero_scalar, dim3(1), dim3(1), 0, h->stream, a.ksum);
  }
  const int tm = (n1 + AT - 1) / AT;
  a.tiles_n = (n2 + AT - 1) / AT;
  const int tiles = lower_only ? tm * (tm + 1) / 2 : tm * a.tiles_n;
  hipLaunchKernelGGL(ffgp_assemble_kernel, dim3(tiles), dim3(256), 0, h->stream, a);
  if (mean_jitter != 0.0)
    hipLaunchKernelGGL(ffgp_mean_jitter_kernel, dim3((n1 + 255) / 256), dim3(256), 0, h->stream, K, ldk, n1, a.ksum,
                       mean_jitter);
  if (hipGetLastError() != hipSuccess) return FFGP_ERR_HIP;
  return FFGP_OK;
}
