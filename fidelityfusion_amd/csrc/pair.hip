// Two-descriptor covariance tiles for gfx950:  K = k_a (+ | x) k_b  in ONE pass, forward and backward.
//
// The reference composes kernels as modules -- SumKernel / ProductKernel (GaussianProcess/kernel.py:172-236), and its own
// demos and two-fidelity models run on SumKernel(LinearKernel, MaternKernel) (cigp_v10.py:81; two_fidelity_models/ResGP.py:25,
// AR_autoRegression.py:31, NAR_NonlinearAR.py:23) -- so torch evaluates each part as its own chain of N x N temporaries and
// adds / multiplies them (>= 12 N x N passes forward, as many again in autograd).  Here each part is a descriptor
//     stationary:  amp * phi(max(||(x - x') o w||^2, clamp))          (the FFGP_KFUN_* profiles)
//     linear:      amp * sum_k w_k^2 (x_k - c_k)(x'_k - c_k)          (LinearKernel, kernel.py:22-63)
// and one 64 x 64 tile pass stages the inputs once per descriptor scaling, accumulates both bilinear forms side by side,
// combines them in registers and applies the Sigma extras (S1-S4) -- write-only over K, like the single-kernel assembly.
// The gradient tile reads G = d(value)/d(Sigma) once, rebuilds both parts from X, routes G to each part (Sum: G;
// Product: G o K_other) and reduces g_w, g_amp, g_kparam and -- for the linear part -- g_center per descriptor.
#include "ffgp_internal.h"

#define AT 64
#define DC 16

struct PairDesc {
  const double* w; const double* amp; const double* center;
  double clamp; double rinv; int kfun;
};

struct PairArgs {
  const double* X1; int n1;
  const double* X2; int n2;
  int D;
  PairDesc k[2];
  int op;                 // FFGP_KOP_SUM | FFGP_KOP_PRODUCT
  // assembly
  const double* diag_add; const double* diag_vec; long diag_stride;
  const double* add_mat; int ld_add; double add_all;
  double* K; int ldk; int lower_only; int symmetric;
  double* ksum;
  int tiles_n;
  // gradient
  const double* G; int ldg; int rect;
  const double* trG; double mj_coef;
  double* partial;        // [blocks][2][2D + 2]: per descriptor  w-sums[D] | centre-sums[D] | amp | kparam
};

__device__ __forceinline__ void pair_tile_of(int t, int lower, int tiles_n, int& ti, int& tj) {
  if (lower) {
    int r = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= t) ++r;
    while (r * (r + 1) / 2 > t) --r;
    ti = r;
    tj = t - r * (r + 1) / 2;
  } else {
    ti = t / tiles_n;
    tj = t % tiles_n;
  }
}

// stage the 64 x 16 chunk of both point sets under one descriptor's scaling: (x - c) * w  (c = 0 for stationary parts).
// Unconditional loads from clamped addresses, masked afterwards (guarded loads serialise on memory latency).
__device__ __forceinline__ void pair_stage(const PairArgs& a, const PairDesc& kd, int d0, int r0, int c0, int tid,
                                           double (*x1s)[DC + 1], double (*x2t)[AT + 1]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = tid + 256 * q;
    const int row = idx >> 4, dd = idx & 15, gd = d0 + dd;
    const int gdc = min(gd, a.D - 1);
    const double wk = kd.w[gdc];
    const double cen = kd.center ? kd.center[gdc] : 0.0;
    const double l1 = a.X1[(size_t)min(r0 + row, a.n1 - 1) * a.D + gdc], l2 = a.X2[(size_t)min(c0 + row, a.n2 - 1) * a.D + gdc];
    x1s[row][dd] = (gd < a.D && r0 + row < a.n1) ? (l1 - cen) * wk : 0.0;
    x2t[dd][row] = (gd < a.D && c0 + row < a.n2) ? (l2 - cen) * wk : 0.0;
  }
}

template <bool LIN>
__device__ __forceinline__ void pair_accum(const double (*x1s)[DC + 1], const double (*x2t)[AT + 1], int tx, int ty,
                                           double (&acc)[4][4]) {
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
        if (LIN) {
          acc[i][j] = __builtin_fma(p[i], q2[j], acc[i][j]);
        } else {
          const double df = p[i] - q2[j];
          acc[i][j] = __builtin_fma(df, df, acc[i][j]);
        }
      }
  }
}

// both bilinear forms of the tile: squared scaled distance (stationary) or scaled dot product (linear), per descriptor
__device__ __forceinline__ void pair_forms(const PairArgs& a, int r0, int c0, int tid, int tx, int ty, double (*x1s)[AT][DC + 1],
                                           double (*x2t)[DC][AT + 1], double (&fa)[4][4], double (&fb)[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) fa[i][j] = fb[i][j] = 0.0;
  const bool lin_a = a.k[0].kfun == FFGP_KFUN_LINEAR, lin_b = a.k[1].kfun == FFGP_KFUN_LINEAR;
  for (int d0 = 0; d0 < a.D; d0 += DC) {
    pair_stage(a, a.k[0], d0, r0, c0, tid, x1s[0], x2t[0]);
    pair_stage(a, a.k[1], d0, r0, c0, tid, x1s[1], x2t[1]);
    __syncthreads();
    if (lin_a) pair_accum<true>(x1s[0], x2t[0], tx, ty, fa); else pair_accum<false>(x1s[0], x2t[0], tx, ty, fa);
    if (lin_b) pair_accum<true>(x1s[1], x2t[1], tx, ty, fb); else pair_accum<false>(x1s[1], x2t[1], tx, ty, fb);
    __syncthreads();
  }
}

// the part's value without its amplitude
__device__ __forceinline__ double pair_profile(const PairDesc& kd, double form) {
  return kd.kfun == FFGP_KFUN_LINEAR ? form : ffgp_kfun_val(kd.kfun, kd.rinv, fmax(form, kd.clamp));
}

__global__ __launch_bounds__(256) void ffgp_assemble_pair_kernel(PairArgs a) {
  __shared__ double x1s[2][AT][DC + 1];
  __shared__ double x2t[2][DC][AT + 1];
  __shared__ double red[4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  int ti, tj;
  pair_tile_of(blockIdx.x, a.lower_only, a.tiles_n, ti, tj);
  const int r0 = ti * AT, c0 = tj * AT;
  double fa[4][4], fb[4][4];
  pair_forms(a, r0, c0, tid, tx, ty, x1s, x2t, fa, fb);

  const double amp_a = a.k[0].amp[0], amp_b = a.k[1].amp[0];
  const double dadd = a.diag_add ? a.diag_add[0] : 0.0;
  double tsum = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = r0 + ty + 16 * i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = c0 + tx + 16 * j;
      if (row < a.n1 && col < a.n2) {
        const double ka = amp_a * pair_profile(a.k[0], fa[i][j]), kb = amp_b * pair_profile(a.k[1], fb[i][j]);
        double k = (a.op == FFGP_KOP_PRODUCT) ? ka * kb : ka + kb;
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
    if (a.lower_only && ti != tj) tsum *= 2.0;
    for (int o = 32; o > 0; o >>= 1) tsum += __shfl_down(tsum, o);
    if ((tid & 63) == 0) red[tid >> 6] = tsum;
    __syncthreads();
    if (tid == 0) atomicAdd(a.ksum, red[0] + red[1] + red[2] + red[3]);
  }
}

__global__ void ffgp_pair_zero_scalar(double* p) { p[0] = 0.0; }
__global__ void ffgp_pair_mean_jitter_kernel(double* K, int ldk, int n, const double* ksum, double coef) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) K[(size_t)i * ldk + i] += coef * ksum[0] / ((double)n * (double)n);
}

// block-wide sum of one value per thread -> thread 0 (red: 4 doubles of LDS)
__device__ __forceinline__ double pair_block_sum(double v, double* red, int tid) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void ffgp_grad_pair_kernel(PairArgs a) {
  __shared__ double x1s[2][AT][DC + 1];
  __shared__ double x2t[2][DC][AT + 1];
  __shared__ double red[4][2 * DC];
  __shared__ double red1[4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  int ti, tj;
  pair_tile_of(blockIdx.x, !a.rect, (a.n2 + AT - 1) / AT, ti, tj);
  const int r0 = ti * AT, c0 = tj * AT;
  double fa[4][4], fb[4][4];
  pair_forms(a, r0, c0, tid, tx, ty, x1s, x2t, fa, fb);

  const double amp_a = a.k[0].amp[0], amp_b = a.k[1].amp[0];
  const bool lin_a = a.k[0].kfun == FFGP_KFUN_LINEAR, lin_b = a.k[1].kfun == FFGP_KFUN_LINEAR;
  const double geff_add = (a.mj_coef != 0.0) ? a.mj_coef * a.trG[0] : 0.0;
  double gl[4][4];   // the tile of G, sixteen loads in flight (clamped addresses; entries outside the mask are not used)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rowc = min(r0 + ty + 16 * i, a.n1 - 1);
      const int colc = a.rect ? min(c0 + tx + 16 * j, a.n2 - 1) : min(c0 + tx + 16 * j, rowc);
      gl[i][j] = a.G[(size_t)rowc * a.ldg + colc];
    }
  // per-entry weights of the two parts (fa / fb are overwritten: Wa, Wb), amplitude and profile-parameter sums
  double s_amp[2] = {0.0, 0.0}, s_kp[2] = {0.0, 0.0};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = r0 + ty + 16 * i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = c0 + tx + 16 * j;
      double wa = 0.0, wb = 0.0;
      if (row < a.n1 && (a.rect ? col < a.n2 : col <= row)) {
        const double sym = (!a.rect && col < row) ? 2.0 : 1.0;
        const double g = sym * (gl[i][j] + geff_add);
        const double sa = lin_a ? fa[i][j] : fmax(fa[i][j], a.k[0].clamp), sb = lin_b ? fb[i][j] : fmax(fb[i][j], a.k[1].clamp);
        const double ea = lin_a ? sa : ffgp_kfun_val(a.k[0].kfun, a.k[0].rinv, sa);
        const double eb = lin_b ? sb : ffgp_kfun_val(a.k[1].kfun, a.k[1].rinv, sb);
        const double ga = (a.op == FFGP_KOP_PRODUCT) ? g * amp_b * eb : g;     // upstream of part a
        const double gb = (a.op == FFGP_KOP_PRODUCT) ? g * amp_a * ea : g;
        s_amp[0] += ga * ea;
        s_amp[1] += gb * eb;
        if (a.k[0].kfun == FFGP_KFUN_RQ) s_kp[0] += ga * amp_a * ffgp_kfun_dparam(FFGP_KFUN_RQ, a.k[0].rinv, sa, ea);
        if (a.k[1].kfun == FFGP_KFUN_RQ) s_kp[1] += gb * amp_b * ffgp_kfun_dparam(FFGP_KFUN_RQ, a.k[1].rinv, sb, eb);
        wa = lin_a ? ga * amp_a : ((fa[i][j] >= a.k[0].clamp) ? ga * amp_a * ffgp_kfun_m2d(a.k[0].kfun, a.k[0].rinv, sa) : 0.0);
        wb = lin_b ? gb * amp_b : ((fb[i][j] >= a.k[1].clamp) ? gb * amp_b * ffgp_kfun_m2d(a.k[1].kfun, a.k[1].rinv, sb) : 0.0);
      }
      fa[i][j] = wa;
      fb[i][j] = wb;
    }
  }

  // per-dimension sums: stationary  sum W df^2;  linear  sum W p q  and  sum W (p + q)
  const int stride = 2 * a.D + 2;
  double* out = a.partial + (size_t)blockIdx.x * 2 * stride;
  for (int e = 0; e < 2; ++e) {
    const bool lin = (e == 0) ? lin_a : lin_b;
    for (int d0 = 0; d0 < a.D; d0 += DC) {
      pair_stage(a, a.k[e], d0, r0, c0, tid, x1s[0], x2t[0]);
      __syncthreads();
      double accd[DC], accc[DC];
#pragma unroll
      for (int dd = 0; dd < DC; ++dd) {
        double p[4], q2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) p[i] = x1s[0][ty + 16 * i][dd];
#pragma unroll
        for (int j = 0; j < 4; ++j) q2[j] = x2t[0][dd][tx + 16 * j];
        double s = 0.0, sc = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const double wv = (e == 0) ? fa[i][j] : fb[i][j];
            if (lin) {
              s = __builtin_fma(wv * p[i], q2[j], s);
              sc = __builtin_fma(wv, p[i] + q2[j], sc);
            } else {
              const double df = p[i] - q2[j];
              s = __builtin_fma(wv * df, df, s);
            }
          }
        accd[dd] = s;
        accc[dd] = sc;
      }
#pragma unroll
      for (int dd = 0; dd < DC; ++dd) {
        double v = accd[dd], c = accc[dd];
        for (int o = 32; o > 0; o >>= 1) {
          v += __shfl_down(v, o);
          c += __shfl_down(c, o);
        }
        if ((tid & 63) == 0) {
          red[tid >> 6][dd] = v;
          red[tid >> 6][DC + dd] = c;
        }
      }
      __syncthreads();
      if (tid < 2 * DC) {
        const int dd = tid & (DC - 1), which = tid >> 4;   // 0: w-sums, 1: centre-sums
        if (d0 + dd < a.D) out[e * stride + which * a.D + d0 + dd] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    double v = pair_block_sum(s_amp[e], red1, tid);
    if (tid == 0) out[e * stride + 2 * a.D] = v;
    v = pair_block_sum(s_kp[e], red1, tid);
    if (tid == 0) out[e * stride + 2 * a.D + 1] = v;
  }
}

struct PairOut {
  double* g_w[2]; double* g_amp[2]; double* g_kparam[2]; double* g_center[2];
  const double* w[2]; int lin[2];
};

// deterministic second stage: one workgroup per (descriptor, slot)
__global__ __launch_bounds__(256) void ffgp_grad_pair_finish(const double* __restrict__ partial, int blocks, int D, PairOut o) {
  __shared__ double red[4];
  const int stride = 2 * D + 2;
  const int e = blockIdx.x / stride, k = blockIdx.x % stride;
  double s = 0.0;
  for (int b = threadIdx.x; b < blocks; b += 256) s += partial[((size_t)b * 2 + e) * stride + k];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x != 0) return;
  s = red[0] + red[1] + red[2] + red[3];
  if (k < D) {
    // stationary: dK/dw_k = -(1/w_k) W df_k^2 ;  linear: dK/dw_k = (2/w_k) amp p_k q_k
    if (o.g_w[e]) o.g_w[e][k] = (o.lin[e] ? 2.0 * s : -s) / o.w[e][k];
  } else if (k < 2 * D) {
    // linear: dK/dc_k = -w_k amp (p_k + q_k)
    if (o.g_center[e]) o.g_center[e][k - D] = o.lin[e] ? -o.w[e][k - D] * s : 0.0;
  } else if (k == 2 * D) {
    if (o.g_amp[e]) o.g_amp[e][0] = s;
  } else if (o.g_kparam[e]) {
    o.g_kparam[e][0] = s;
  }
}

static int pair_fill(PairArgs& a, const ffgp_kdesc* k, int op) {
  if (!k || (op != FFGP_KOP_SUM && op != FFGP_KOP_PRODUCT)) return FFGP_ERR_ARG;
  for (int e = 0; e < 2; ++e) {
    if (k[e].kfun < FFGP_KFUN_SE || k[e].kfun > FFGP_KFUN_LINEAR || !k[e].w_dev || !k[e].amp_dev) return FFGP_ERR_ARG;
    a.k[e].w = k[e].w_dev;
    a.k[e].amp = k[e].amp_dev;
    a.k[e].center = (k[e].kfun == FFGP_KFUN_LINEAR) ? k[e].center_dev : nullptr;
    a.k[e].clamp = k[e].clamp_min;
    a.k[e].rinv = (k[e].kparam != 0.0) ? 1.0 / k[e].kparam : 1.0;
    a.k[e].kfun = k[e].kfun;
  }
  a.op = op;
  return FFGP_OK;
}

int ffgp_assemble_pair_impl(ffgp_handle* h, const double* X1, int n1, const double* X2, int n2, int D, const ffgp_kdesc* k, int op,
                            const double* diag_add, const double* diag_vec, long diag_stride, const double* add_mat, int ld_add,
                            double add_all, double mean_jitter, double* K, int ldk, int lower_only) {
  if (n1 <= 0 || n2 <= 0) return FFGP_OK;
  if (!X1 || !X2 || !K || D <= 0 || ldk < n2) return FFGP_ERR_ARG;
  const bool symmetric = (X1 == X2 && n1 == n2);
  const bool extras = diag_add || diag_vec || add_mat || add_all != 0.0 || mean_jitter != 0.0;
  if ((extras || lower_only) && !symmetric) return FFGP_ERR_ARG;
  PairArgs a = {};
  FFGP_CHECK(pair_fill(a, k, op));
  a.X1 = X1; a.n1 = n1; a.X2 = X2; a.n2 = n2; a.D = D;
  a.diag_add = diag_add; a.diag_vec = diag_vec; a.diag_stride = diag_stride;
  a.add_mat = add_mat; a.ld_add = ld_add; a.add_all = add_all;
  a.K = K; a.ldk = ldk; a.lower_only = lower_only ? 1 : 0; a.symmetric = symmetric ? 1 : 0;
  if (mean_jitter != 0.0) {
    a.ksum = h->d_scal + 32;
    hipLaunchKernelGGL(ffgp_pair_zero_scalar, dim3(1), dim3(1), 0, h->stream, a.ksum);
  }
  const int tm = (n1 + AT - 1) / AT;
  a.tiles_n = (n2 + AT - 1) / AT;
  const int tiles = lower_only ? tm * (tm + 1) / 2 : tm * a.tiles_n;
  hipLaunchKernelGGL(ffgp_assemble_pair_kernel, dim3(tiles), dim3(256), 0, h->stream, a);
  if (mean_jitter != 0.0)
    hipLaunchKernelGGL(ffgp_pair_mean_jitter_kernel, dim3((n1 + 255) / 256), dim3(256), 0, h->stream, K, ldk, n1, a.ksum, mean_jitter);
  if (hipGetLastError() != hipSuccess) return FFGP_ERR_HIP;
  return FFGP_OK;
}

size_t ffgp_grad_pair_partial_doubles(int n1, int n2, int D, int rect) {
  const size_t tm = (n1 + AT - 1) / AT, tn = (n2 + AT - 1) / AT;
  return (rect ? tm * tn : tm * (tm + 1) / 2) * 2 * (size_t)(2 * D + 2);
}

// rect = 0: G is the lower triangle of a symmetric [n1, n1] weight matrix (trG_dev / mj_coef: the mean-jitter chain of S2);
// rect = 1: a dense [n1, n2] upstream dK of a standalone kernel call.
int ffgp_grad_pair_impl(ffgp_handle* h, const double* X1, int n1, const double* X2, int n2, int D, const ffgp_kdesc* k, int op,
                        const double* G, int ldg, int rect, const double* trG_dev, double mj_coef, double* partial_ws,
                        const ffgp_kdesc_grads* g) {
  if (n1 <= 0 || n2 <= 0 || !g) return FFGP_OK;
  if (!X1 || !X2 || !G || D <= 0 || !partial_ws) return FFGP_ERR_ARG;
  PairArgs a = {};
  FFGP_CHECK(pair_fill(a, k, op));
  a.X1 = X1; a.n1 = n1; a.X2 = X2; a.n2 = n2; a.D = D;
  a.G = G; a.ldg = ldg; a.rect = rect ? 1 : 0; a.trG = trG_dev; a.mj_coef = trG_dev ? mj_coef : 0.0;
  a.partial = partial_ws;
  const int tm = (n1 + AT - 1) / AT, tn = (n2 + AT - 1) / AT;
  const int blocks = rect ? tm * tn : tm * (tm + 1) / 2;
  PairOut o;
  for (int e = 0; e < 2; ++e) {
    o.g_w[e] = g[e].g_w_dev; o.g_amp[e] = g[e].g_amp_dev; o.g_kparam[e] = g[e].g_kparam_dev; o.g_center[e] = g[e].g_center_dev;
    o.w[e] = k[e].w_dev;
    o.lin[e] = (k[e].kfun == FFGP_KFUN_LINEAR);
  }
  hipLaunchKernelGGL(ffgp_grad_pair_kernel, dim3(blocks), dim3(256), 0, h->stream, a);
  hipLaunchKernelGGL(ffgp_grad_pair_finish, dim3(2 * (2 * D + 2)), dim3(256), 0, h->stream, partial_ws, blocks, D, o);
  if (hipGetLastError() != hipSuccess) return FFGP_ERR_HIP;
  return FFGP_OK;
}
