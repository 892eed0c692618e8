// Composed-kernel covariance tiles for gfx950:  K = k_a (+ | x) k_b -- and nested compositions of up to four leaves -- in ONE
// pass, forward and backward.
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
#define TREE_MAX 4

struct PairDesc {
  const double* w; const double* amp; const double* center;
  double clamp; double rinv; int kfun;
};

// Composed kernels with up to four leaves (nested SumKernel / ProductKernel objects, kernel.py:172-236).  Every binary tree with
// <= 4 leaves is, up to the order of the operands of a commutative node (bit-identical in IEEE arithmetic), one of
//     nl = 2:  l0 op0 l1          nl = 3:  (l0 op0 l1) op1 l2
//     nl = 4, chain:  ((l0 op0 l1) op1 l2) op2 l3          nl = 4, balanced:  (l0 op0 l1) op2 (l2 op1 l3)
// so the tile kernels are instantiated per leaf count and evaluate the combination as straight-line code on registers.
struct PairArgs {
  const double* X1; int n1;
  const double* X2; int n2;
  int D;
  PairDesc k[TREE_MAX];
  int nl, shape, op[3];   // FFGP_KOP_*; shape: FFGP_TREE_CHAIN | FFGP_TREE_BALANCED (nl = 4)
  // assembly
  const double* diag_add; const double* diag_vec; long diag_stride;
  const double* add_mat; int ld_add; double add_all;
  double* K; int ldk; int lower_only; int symmetric;
  double* ksum;
  int tiles_n;
  // gradient
  const double* G; int ldg; int rect;
  const double* trG; double mj_coef;
  double* partial;        // [blocks][nl][2D + 2]: per leaf  w-sums[D] | centre-sums[D] | amp | kparam
  double* Wt; int ldw; size_t wt_stride;   // input weights: one [n1, ldw] matrix per leaf
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

// the bilinear form of every leaf on this tile: squared scaled distance (stationary) or scaled dot product (linear);
// the leaves are staged two at a time (two descriptor scalings of the same 64 x 16 input chunk per barrier pair)
template <int NL, int E0>
__device__ __forceinline__ void pair_forms_step(const PairArgs& a, int r0, int c0, int tid, int tx, int ty, double (*x1s)[AT][DC + 1],
                                                double (*x2t)[DC][AT + 1], double (&f)[NL][4][4]) {
  constexpr bool two = (E0 + 1 < NL);
  constexpr int E1 = two ? E0 + 1 : E0;
  const bool lin_a = a.k[E0].kfun == FFGP_KFUN_LINEAR, lin_b = a.k[E1].kfun == FFGP_KFUN_LINEAR;
  for (int d0 = 0; d0 < a.D; d0 += DC) {
    pair_stage(a, a.k[E0], d0, r0, c0, tid, x1s[0], x2t[0]);
    if (two) pair_stage(a, a.k[E1], d0, r0, c0, tid, x1s[1], x2t[1]);
    __syncthreads();
    if (lin_a) pair_accum<true>(x1s[0], x2t[0], tx, ty, f[E0]); else pair_accum<false>(x1s[0], x2t[0], tx, ty, f[E0]);
    if (two) {
      if (lin_b) pair_accum<true>(x1s[1], x2t[1], tx, ty, f[E1]); else pair_accum<false>(x1s[1], x2t[1], tx, ty, f[E1]);
    }
    __syncthreads();
  }
}

template <int NL>
__device__ __forceinline__ void pair_forms(const PairArgs& a, int r0, int c0, int tid, int tx, int ty, double (*x1s)[AT][DC + 1],
                                           double (*x2t)[DC][AT + 1], double (&f)[NL][4][4]) {
#pragma unroll
  for (int e = 0; e < NL; ++e)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) f[e][i][j] = 0.0;
  pair_forms_step<NL, 0>(a, r0, c0, tid, tx, ty, x1s, x2t, f);
  if constexpr (NL > 2) pair_forms_step<NL, 2>(a, r0, c0, tid, tx, ty, x1s, x2t, f);
}

// the part's value without its amplitude
__device__ __forceinline__ double pair_profile(const PairDesc& kd, double form) {
  return kd.kfun == FFGP_KFUN_LINEAR ? form : ffgp_kfun_val(kd.kfun, kd.rinv, fmax(form, kd.clamp));
}

// one node: separately rounded product / sum, as torch's elementwise kernels round them (no contraction into an fma)
__device__ __forceinline__ double tree_op(int op, double x, double y) {
#pragma clang fp contract(off)
  const double pr = x * y, sm = x + y;
  return op == FFGP_KOP_PRODUCT ? pr : sm;
}

template <int NL>
__device__ __forceinline__ double tree_eval(const PairArgs& a, const double (&v)[NL]) {
  const double t0 = tree_op(a.op[0], v[0], v[1]);
  if (NL == 2) return t0;
  if (NL == 3) return tree_op(a.op[1], t0, v[NL > 2 ? 2 : 0]);
  if (a.shape == FFGP_TREE_BALANCED) return tree_op(a.op[2], t0, tree_op(a.op[1], v[NL > 2 ? 2 : 0], v[NL > 3 ? 3 : 0]));
  return tree_op(a.op[2], tree_op(a.op[1], t0, v[NL > 2 ? 2 : 0]), v[NL > 3 ? 3 : 0]);
}

// d root / d leaf values times the upstream g (reverse sweep over the same straight-line code)
template <int NL>
__device__ __forceinline__ void tree_back(const PairArgs& a, const double (&v)[NL], double g, double (&gv)[NL]) {
  const double t0 = tree_op(a.op[0], v[0], v[1]);
  double gt0 = g;
  if (NL == 3) {
    const bool pr = a.op[1] == FFGP_KOP_PRODUCT;
    gt0 = pr ? g * v[NL > 2 ? 2 : 0] : g;
    gv[NL > 2 ? 2 : 0] = pr ? g * t0 : g;
  }
  if (NL == 4) {
    const int i2 = NL > 2 ? 2 : 0, i3 = NL > 3 ? 3 : 0;
    const bool p1 = a.op[1] == FFGP_KOP_PRODUCT, p2 = a.op[2] == FFGP_KOP_PRODUCT;
    if (a.shape == FFGP_TREE_BALANCED) {
      const double t1 = tree_op(a.op[1], v[i2], v[i3]);
      gt0 = p2 ? g * t1 : g;
      const double gt1 = p2 ? g * t0 : g;
      gv[i2] = p1 ? gt1 * v[i3] : gt1;
      gv[i3] = p1 ? gt1 * v[i2] : gt1;
    } else {
      const double t1 = tree_op(a.op[1], t0, v[i2]);
      const double gt1 = p2 ? g * v[i3] : g;
      gv[i3] = p2 ? g * t1 : g;
      gt0 = p1 ? gt1 * v[i2] : gt1;
      gv[i2] = p1 ? gt1 * t0 : gt1;
    }
  }
  const bool p0 = a.op[0] == FFGP_KOP_PRODUCT;
  gv[0] = p0 ? gt0 * v[1] : gt0;
  gv[1] = p0 ? gt0 * v[0] : gt0;
}

template <int NL>
__global__ __launch_bounds__(256) void ffgp_assemble_pair_kernel(PairArgs a) {
  __shared__ double x1s[2][AT][DC + 1];
  __shared__ double x2t[2][DC][AT + 1];
  __shared__ double red[4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  int ti, tj;
  pair_tile_of(blockIdx.x, a.lower_only, a.tiles_n, ti, tj);
  const int r0 = ti * AT, c0 = tj * AT;
  double f[NL][4][4];
  pair_forms<NL>(a, r0, c0, tid, tx, ty, x1s, x2t, f);

  // every leaf's value in place (entries past the edge hold harmless values of the zero-padded inputs)
#pragma unroll
  for (int e = 0; e < NL; ++e) {
    const double amp = a.k[e].amp[0];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) f[e][i][j] = amp * pair_profile(a.k[e], f[e][i][j]);
  }
  const double dadd = a.diag_add ? a.diag_add[0] : 0.0;
  double tsum = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = r0 + ty + 16 * i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = c0 + tx + 16 * j;
      if (row < a.n1 && col < a.n2) {
        double v[NL];
#pragma unroll
        for (int e = 0; e < NL; ++e) v[e] = f[e][i][j];
        double k = tree_eval<NL>(a, v);
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

// per-entry weights of every leaf for an upstream tile g (f is overwritten: leaf e's form -> its weight W_e); optionally the
// amplitude / profile-parameter sums.  W_e = (d root / d leaf_e) * amp_e * (-2 phi') for a stationary leaf (0 on the clamp),
// (d root / d leaf_e) * amp_e for a linear one.
template <int NL, bool SUMS>
__device__ __forceinline__ void pair_weights(const PairArgs& a, int r0, int c0, int tx, int ty, const double (&gl)[4][4], double geff_add,
                                             double (&f)[NL][4][4], double (&s_amp)[NL], double (&s_kp)[NL]) {
  double amp[NL];
#pragma unroll
  for (int e = 0; e < NL; ++e) amp[e] = a.k[e].amp[0];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = r0 + ty + 16 * i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = c0 + tx + 16 * j;
      double wout[NL];
#pragma unroll
      for (int e = 0; e < NL; ++e) wout[e] = 0.0;
      if (row < a.n1 && (a.rect ? col < a.n2 : col <= row)) {
        const double sym = (!a.rect && col < row) ? 2.0 : 1.0;
        const double g = sym * (gl[i][j] + geff_add);
        double sc[NL], ev[NL], v[NL], gv[NL];
#pragma unroll
        for (int e = 0; e < NL; ++e) {
          const bool lin = a.k[e].kfun == FFGP_KFUN_LINEAR;
          sc[e] = lin ? f[e][i][j] : fmax(f[e][i][j], a.k[e].clamp);
          ev[e] = lin ? sc[e] : ffgp_kfun_val(a.k[e].kfun, a.k[e].rinv, sc[e]);
          v[e] = amp[e] * ev[e];
        }
        tree_back<NL>(a, v, g, gv);
#pragma unroll
        for (int e = 0; e < NL; ++e) {
          const bool lin = a.k[e].kfun == FFGP_KFUN_LINEAR;
          if (SUMS) {
            s_amp[e] += gv[e] * ev[e];
            if (a.k[e].kfun == FFGP_KFUN_RQ) s_kp[e] += gv[e] * amp[e] * ffgp_kfun_dparam(FFGP_KFUN_RQ, a.k[e].rinv, sc[e], ev[e]);
          }
          wout[e] = lin ? gv[e] * amp[e]
                        : ((f[e][i][j] >= a.k[e].clamp) ? gv[e] * amp[e] * ffgp_kfun_m2d(a.k[e].kfun, a.k[e].rinv, sc[e]) : 0.0);
        }
      }
#pragma unroll
      for (int e = 0; e < NL; ++e) f[e][i][j] = wout[e];
    }
  }
}

// per-dimension sums of one leaf's weights: stationary  sum W df^2;  linear  sum W p q  and  sum W (p + q)
template <int E>
__device__ __forceinline__ void pair_dim_sums(const PairArgs& a, int r0, int c0, int tid, int tx, int ty, double (*x1s)[DC + 1],
                                              double (*x2t)[AT + 1], double (*red)[2 * DC], const double (&fw)[4][4], double* out) {
  const bool lin = a.k[E].kfun == FFGP_KFUN_LINEAR;
  for (int d0 = 0; d0 < a.D; d0 += DC) {
    pair_stage(a, a.k[E], d0, r0, c0, tid, x1s, x2t);
    __syncthreads();
    double accd[DC], accc[DC];
#pragma unroll
    for (int dd = 0; dd < DC; ++dd) {
      double p[4], q2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) p[i] = x1s[ty + 16 * i][dd];
#pragma unroll
      for (int j = 0; j < 4; ++j) q2[j] = x2t[dd][tx + 16 * j];
      double s = 0.0, sc = 0.0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const double wv = fw[i][j];
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
      if (d0 + dd < a.D) out[which * a.D + d0 + dd] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    }
    __syncthreads();
  }
}

template <int NL>
__global__ __launch_bounds__(256) void ffgp_grad_pair_kernel(PairArgs a) {
  __shared__ double x1s[2][AT][DC + 1];
  __shared__ double x2t[2][DC][AT + 1];
  __shared__ double red[4][2 * DC];
  __shared__ double red1[4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  int ti, tj;
  pair_tile_of(blockIdx.x, !a.rect, (a.n2 + AT - 1) / AT, ti, tj);
  const int r0 = ti * AT, c0 = tj * AT;
  double f[NL][4][4];
  pair_forms<NL>(a, r0, c0, tid, tx, ty, x1s, x2t, f);

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
  double s_amp[NL], s_kp[NL];
#pragma unroll
  for (int e = 0; e < NL; ++e) s_amp[e] = s_kp[e] = 0.0;
  pair_weights<NL, true>(a, r0, c0, tx, ty, gl, geff_add, f, s_amp, s_kp);

  const int stride = 2 * a.D + 2;
  double* out = a.partial + (size_t)blockIdx.x * NL * stride;
  pair_dim_sums<0>(a, r0, c0, tid, tx, ty, x1s[0], x2t[0], red, f[0], out);
  pair_dim_sums<1>(a, r0, c0, tid, tx, ty, x1s[0], x2t[0], red, f[1], out + stride);
  if constexpr (NL > 2) pair_dim_sums<2>(a, r0, c0, tid, tx, ty, x1s[0], x2t[0], red, f[2], out + 2 * stride);
  if constexpr (NL > 3) pair_dim_sums<3>(a, r0, c0, tid, tx, ty, x1s[0], x2t[0], red, f[3], out + 3 * stride);
#pragma unroll
  for (int e = 0; e < NL; ++e) {
    double v = pair_block_sum(s_amp[e], red1, tid);
    if (tid == 0) out[e * stride + 2 * a.D] = v;
    v = pair_block_sum(s_kp[e], red1, tid);
    if (tid == 0) out[e * stride + 2 * a.D + 1] = v;
  }
}

// Input gradients of a composed kernel call: for an upstream dK [n1, n2], one pass writes every leaf's weight matrix
//     stationary leaf:  Wt_e = dK o (d root / d leaf_e) o amp_e (-2 phi'_e)    ->  dX1 = -w_e^2 o (rowsum(Wt_e) o X1 - Wt_e X2)
//     linear leaf:      Wt_e = dK o (d root / d leaf_e) o amp_e                ->  dX1 =  w_e^2 o (Wt_e (X2 - c_e))
// (two thin products per leaf on the matrix cores, done by the caller like the single-kernel ffgp_kernel_input_weights).
template <int NL>
__global__ __launch_bounds__(256) void ffgp_pair_wt_kernel(PairArgs a) {
  __shared__ double x1s[2][AT][DC + 1];
  __shared__ double x2t[2][DC][AT + 1];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int tn = (a.n2 + AT - 1) / AT;
  const int ti = blockIdx.x / tn, tj = blockIdx.x % tn;
  const int r0 = ti * AT, c0 = tj * AT;
  double f[NL][4][4];
  pair_forms<NL>(a, r0, c0, tid, tx, ty, x1s, x2t, f);
  double gl[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      gl[i][j] = a.G[(size_t)min(r0 + ty + 16 * i, a.n1 - 1) * a.ldg + min(c0 + tx + 16 * j, a.n2 - 1)];
  double s_amp[NL], s_kp[NL];
  pair_weights<NL, false>(a, r0, c0, tx, ty, gl, 0.0, f, s_amp, s_kp);
#pragma unroll
  for (int e = 0; e < NL; ++e)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r0 + ty + 16 * i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = c0 + tx + 16 * j;
        if (row < a.n1 && col < a.n2) a.Wt[e * a.wt_stride + (size_t)row * a.ldw + col] = f[e][i][j];
      }
    }
}

struct PairOut {
  double* g_w[TREE_MAX]; double* g_amp[TREE_MAX]; double* g_kparam[TREE_MAX]; double* g_center[TREE_MAX];
  const double* w[TREE_MAX]; int lin[TREE_MAX];
};

// deterministic second stage: one workgroup per (leaf, slot)
__global__ __launch_bounds__(256) void ffgp_grad_pair_finish(const double* __restrict__ partial, int blocks, int D, int nl, PairOut o) {
  __shared__ double red[4];
  const int stride = 2 * D + 2;
  const int e = blockIdx.x / stride, k = blockIdx.x % stride;
  double s = 0.0;
  for (int b = threadIdx.x; b < blocks; b += 256) s += partial[((size_t)b * nl + e) * stride + k];
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

static int pair_fill(PairArgs& a, const ffgp_ktree* t) {
  if (!t || !t->leaf || t->n_leaves < 2 || t->n_leaves > TREE_MAX) return FFGP_ERR_ARG;
  if (t->n_leaves == 4 && t->shape != FFGP_TREE_CHAIN && t->shape != FFGP_TREE_BALANCED) return FFGP_ERR_ARG;
  for (int i = 0; i + 1 < t->n_leaves; ++i)
    if (t->op[i] != FFGP_KOP_SUM && t->op[i] != FFGP_KOP_PRODUCT) return FFGP_ERR_ARG;
  const ffgp_kdesc* k = t->leaf;
  for (int e = 0; e < t->n_leaves; ++e) {
    if (k[e].kfun < FFGP_KFUN_SE || k[e].kfun > FFGP_KFUN_LINEAR || !k[e].w_dev || !k[e].amp_dev) return FFGP_ERR_ARG;
    a.k[e].w = k[e].w_dev;
    a.k[e].amp = k[e].amp_dev;
    a.k[e].center = (k[e].kfun == FFGP_KFUN_LINEAR) ? k[e].center_dev : nullptr;
    a.k[e].clamp = k[e].clamp_min;
    a.k[e].rinv = (k[e].kparam != 0.0) ? 1.0 / k[e].kparam : 1.0;
    a.k[e].kfun = k[e].kfun;
  }
  a.nl = t->n_leaves;
  a.shape = (t->n_leaves == 4) ? t->shape : FFGP_TREE_CHAIN;
  for (int i = 0; i < 3; ++i) a.op[i] = (i + 1 < t->n_leaves) ? t->op[i] : FFGP_KOP_SUM;
  return FFGP_OK;
}

#define PAIR_LAUNCH(KERNEL, NLV, GRID, H, ARGS)                                                          \
  do {                                                                                                   \
    if ((NLV) == 2) hipLaunchKernelGGL(KERNEL<2>, dim3(GRID), dim3(256), 0, (H)->stream, ARGS);          \
    else if ((NLV) == 3) hipLaunchKernelGGL(KERNEL<3>, dim3(GRID), dim3(256), 0, (H)->stream, ARGS);     \
    else hipLaunchKernelGGL(KERNEL<4>, dim3(GRID), dim3(256), 0, (H)->stream, ARGS);                     \
  } while (0)

int ffgp_assemble_pair_impl(ffgp_handle* h, const double* X1, int n1, const double* X2, int n2, int D, const ffgp_ktree* t,
                            const double* diag_add, const double* diag_vec, long diag_stride, const double* add_mat, int ld_add,
                            double add_all, double mean_jitter, double* K, int ldk, int lower_only) {
  if (n1 <= 0 || n2 <= 0) return FFGP_OK;
  if (!X1 || !X2 || !K || D <= 0 || ldk < n2) return FFGP_ERR_ARG;
  const bool symmetric = (X1 == X2 && n1 == n2);
  const bool extras = diag_add || diag_vec || add_mat || add_all != 0.0 || mean_jitter != 0.0;
  if ((extras || lower_only) && !symmetric) return FFGP_ERR_ARG;
  PairArgs a = {};
  FFGP_CHECK(pair_fill(a, t));
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
  PAIR_LAUNCH(ffgp_assemble_pair_kernel, a.nl, tiles, h, a);
  if (mean_jitter != 0.0)
    hipLaunchKernelGGL(ffgp_pair_mean_jitter_kernel, dim3((n1 + 255) / 256), dim3(256), 0, h->stream, K, ldk, n1, a.ksum, mean_jitter);
  if (hipGetLastError() != hipSuccess) return FFGP_ERR_HIP;
  return FFGP_OK;
}

size_t ffgp_grad_pair_partial_doubles(int n1, int n2, int D, int rect, int nl) {
  const size_t tm = (n1 + AT - 1) / AT, tn = (n2 + AT - 1) / AT;
  return (rect ? tm * tn : tm * (tm + 1) / 2) * (size_t)nl * (size_t)(2 * D + 2);
}

// rect = 0: G is the lower triangle of a symmetric [n1, n1] weight matrix (trG_dev / mj_coef: the mean-jitter chain of S2);
// rect = 1: a dense [n1, n2] upstream dK of a standalone kernel call.
int ffgp_grad_pair_impl(ffgp_handle* h, const double* X1, int n1, const double* X2, int n2, int D, const ffgp_ktree* t,
                        const double* G, int ldg, int rect, const double* trG_dev, double mj_coef, double* partial_ws,
                        const ffgp_kdesc_grads* g) {
  if (n1 <= 0 || n2 <= 0 || !g) return FFGP_OK;
  if (!X1 || !X2 || !G || D <= 0 || !partial_ws) return FFGP_ERR_ARG;
  PairArgs a = {};
  FFGP_CHECK(pair_fill(a, t));
  a.X1 = X1; a.n1 = n1; a.X2 = X2; a.n2 = n2; a.D = D;
  a.G = G; a.ldg = ldg; a.rect = rect ? 1 : 0; a.trG = trG_dev; a.mj_coef = trG_dev ? mj_coef : 0.0;
  a.partial = partial_ws;
  const int tm = (n1 + AT - 1) / AT, tn = (n2 + AT - 1) / AT;
  const int blocks = rect ? tm * tn : tm * (tm + 1) / 2;
  PairOut o = {};
  for (int e = 0; e < a.nl; ++e) {
    o.g_w[e] = g[e].g_w_dev; o.g_amp[e] = g[e].g_amp_dev; o.g_kparam[e] = g[e].g_kparam_dev; o.g_center[e] = g[e].g_center_dev;
    o.w[e] = t->leaf[e].w_dev;
    o.lin[e] = (t->leaf[e].kfun == FFGP_KFUN_LINEAR);
  }
  PAIR_LAUNCH(ffgp_grad_pair_kernel, a.nl, blocks, h, a);
  hipLaunchKernelGGL(ffgp_grad_pair_finish, dim3(a.nl * (2 * D + 2)), dim3(256), 0, h->stream, partial_ws, blocks, D, a.nl, o);
  if (hipGetLastError() != hipSuccess) return FFGP_ERR_HIP;
  return FFGP_OK;
}

int ffgp_pair_wt_impl(ffgp_handle* h, const double* X1, int n1, const double* X2, int n2, int D, const ffgp_ktree* t, const double* dK,
                      int ldk, double* Wt, int ldw, long leaf_stride) {
  if (n1 <= 0 || n2 <= 0) return FFGP_OK;
  if (!X1 || !X2 || !dK || !Wt || D <= 0 || ldk < n2 || ldw < n2 || leaf_stride < (long)n1 * ldw) return FFGP_ERR_ARG;
  PairArgs a = {};
  FFGP_CHECK(pair_fill(a, t));
  a.X1 = X1; a.n1 = n1; a.X2 = X2; a.n2 = n2; a.D = D;
  a.G = dK; a.ldg = ldk; a.rect = 1;
  a.Wt = Wt; a.ldw = ldw; a.wt_stride = (size_t)leaf_stride;
  const int tm = (n1 + AT - 1) / AT, tn = (n2 + AT - 1) / AT;
  PAIR_LAUNCH(ffgp_pair_wt_kernel, a.nl, tm * tn, h, a);
  if (hipGetLastError() != hipSuccess) return FFGP_ERR_HIP;
  return FFGP_OK;
}
