// One-kernel NLML (+ closed-form gradients) for the sizes the reference's own demos and experiments run (N = 16 ... 128;
// GaussianProcess/cigp_v10.py:76-79, FidelityFusion_Models/ResGP.py:115-164, Experiments/GAR_Aligned/exp_aligned.py:58-126): the
// blocked path of api.hip needs ~18 launches for work that one workgroup holds in LDS, and at these sizes a step IS its launches.
//
// One workgroup of 256 threads, everything in LDS (<= 136 KB): Sigma as a packed lower triangle that becomes L, L^-1, Sigma^-1 and G in
// place.  assemble (any radial profile, all Sigma extras incl. the mean(K) jitter) -> right-looking Cholesky (2-D thread tiling of the
// trailing update) -> L^-1 row by row -> Gamma = L^-1 Y, A = L^-T Gamma -> value (V1 / V2) -> Sigma^-1 = L^-T L^-1 -> G -> the
// gradient reductions of grad.hip (same formulas) -> optional raw-parameter links and output scale (ffgp_nlml_fused_raw).
// Limits: n <= 128, D <= 16, d <= 16; no composed kernel, no caller-built covariance, no g_cov.  Measured (tools/small_kernel_bench.py,
// forward + gradients, one call): n = 16: 0.050 ms against 0.099 on the blocked path, n = 32: 0.082 / 0.109, n = 48: 0.121 / 0.110,
// n = 128: 0.58 / 0.126 -- 2 n columns / rows of barrier-separated LDS phases on one CU lose to ~18 launches that use the whole chip
// beyond n ~ 40, so the library takes this path for n <= 40 only (option "small_max_n").
#include "ffgp_internal.h"

#define SM_N 128
#define SM_D 16
#define SM_Y 16
#define SM_T 256    // threads (1024 were measured: the 16-wave barriers cost more than the extra waves hide)
#define SM_T2 1024  // threads of the finishing launch (FROM_FACTOR)
#define SM_TG 16    // the trailing update's thread grid is SM_TG x SM_TG
#define SM_MAX_FAST_N 40   // above this the blocked path of api.hip (whole chip, MFMA kernels) is faster: tools/small_kernel_bench.py

struct SmallArgs {
  int n, D, d;
  const double* X; const double* Y;
  const double* w; const double* amp; const double* dadd;   // raw (links) or effective
  ffgp_links l; int has_links;
  double clamp; int kfun; double rinv;
  const double* diag_vec; long diag_stride; const double* add_mat; int ld_add; double add_all; double mean_jitter;
  double pi_const; int v2;
  double* nll;
  double* g_w; double* g_amp; double* g_dadd; double* g_Y; double* g_diag_vec; double* g_kparam;
  int want_grad;
  int* info;
  int info_max;         // 1: batch member (atomicMax into the shared status word instead of a plain store)
  const double* dinv;   // FROM_FACTOR: L^-1 as the blocked factorisation left it (one 128 x 128 block, lower part valid)
};

__device__ __forceinline__ double sm_link_val(int kind, double p, double c) {
  switch (kind) {
    case FFGP_LINK_INV_ABS_EPS: return 1.0 / (fabs(p) + c);
    case FFGP_LINK_EXP_NEG: return exp(-p) + c;
    case FFGP_LINK_INV: return 1.0 / p + c;
    case FFGP_LINK_ABS: return fabs(p);
    case FFGP_LINK_EXP_SQ: { const double e = exp(p); return e * e; }
    case FFGP_LINK_SQUARE: return p * p + c;
    default: return p;
  }
}
__device__ __forceinline__ double sm_link_der(int kind, double p, double c) {
  switch (kind) {
    case FFGP_LINK_INV_ABS_EPS: { const double a = fabs(p) + c; return ((p > 0.0) ? -1.0 : ((p < 0.0) ? 1.0 : 0.0)) / (a * a); }
    case FFGP_LINK_EXP_NEG: return -exp(-p);
    case FFGP_LINK_INV: return -1.0 / (p * p);
    case FFGP_LINK_ABS: return (p > 0.0) ? 1.0 : ((p < 0.0) ? -1.0 : 0.0);
    case FFGP_LINK_EXP_SQ: { const double e = exp(p); return 2.0 * e * e; }
    case FFGP_LINK_SQUARE: return 2.0 * p;
    default: return 1.0;
  }
}

__device__ __forceinline__ int sm_pk(int i, int j) { return i * (i + 1) / 2 + j; }   // packed lower triangle, j <= i
__device__ __forceinline__ void sm_unpk(int e, int& i, int& j) {
  i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
  while ((i + 1) * (i + 2) / 2 <= e) ++i;
  while (i * (i + 1) / 2 > e) --i;
  j = e - i * (i + 1) / 2;
}

// sum over the workgroup, result in every thread; red: T / 64 doubles of LDS
template <int T>
__device__ __forceinline__ double sm_bsum(double v, double* red, int tid) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int q = 0; q < T / 64; ++q) t += red[q];
  return t;
}

// FROM_FACTOR = false: everything here (n <= 40 by default).  FROM_FACTOR = true (40 < n <= 128): Sigma was assembled and factored by
// the blocked path's own kernels -- the diagonal-block kernel leaves L^-1 of a single block in the handle's store -- and this launch
// does the rest of the call (Gamma, A, value, Sigma^-1, G, every gradient, links, output scale): 7 launches per call instead of 21.
// (the finishing launch has no column-by-column phases, i.e. few barriers: 1024 threads hide the LDS latency that one wave per SIMD cannot)
template <bool FROM_FACTOR>
__device__ __forceinline__ void sm_body(const SmallArgs& a) {
  constexpr int T = FROM_FACTOR ? SM_T2 : SM_T;
  extern __shared__ double sm[];
  double* Sp = sm;                          // packed lower [n (n + 1) / 2]
  double* Xs = Sp + SM_N * (SM_N + 1) / 2;  // [n][17] scaled inputs
  double* Ym = Xs + SM_N * (SM_D + 1);      // [n][d]  Y
  double* Gam = Ym + SM_N * SM_Y;           // Gamma = L^-1 Y
  double* Am = Gam + SM_N * SM_Y;           // A = Sigma^-1 Y
  double* Bm = Am + SM_N * SM_Y;            // B = Sigma^-1 A (V2)
  double* rowb = Bm + SM_N * SM_Y;          // [n] a row / column of L in flight
  double* dinv = rowb + SM_N;               // [n] 1 / L_ii
  double* wv = dinv + SM_N;                 // [16] effective inverse length scales
  double* sc = wv + SM_D;                   // [8] scalars: amp, dadd, logdet, ...
  double* red = sc + 8;                     // [SM_T / 64]
  const int tid = threadIdx.x, n = a.n, D = a.D, d = a.d;
  const int npair = n * (n + 1) / 2;

  // ---- parameters
  if (tid < D) {
    const double r = a.w[(a.has_links && a.l.w_broadcast) ? 0 : tid];
    wv[tid] = a.has_links ? sm_link_val(a.l.w_link, r, a.l.w_c) : r;
  }
  if (tid == 0) {
    sc[0] = a.has_links ? sm_link_val(a.l.amp_link, a.amp[0], a.l.amp_c) : a.amp[0];
    sc[1] = a.dadd ? (a.has_links ? sm_link_val(a.l.dadd_link, a.dadd[0], a.l.dadd_c) : a.dadd[0]) : 0.0;
  }
  __syncthreads();
  for (int idx = tid; idx < n * D; idx += T) {
    const int i = idx / D, k = idx - i * D;
    Xs[i * (SM_D + 1) + k] = a.X[idx] * wv[k];
  }
  for (int idx = tid; idx < n * d; idx += T) Ym[idx] = a.Y[idx];
  __syncthreads();
  const double amp = sc[0], dadd = sc[1];

  double logdet = 0.0;
  if constexpr (FROM_FACTOR) {
    for (int e = tid; e < npair; e += T) {
      int i, j;
      sm_unpk(e, i, j);
      const double v = a.dinv[i * 128 + j];
      Sp[e] = v;
      if (i == j) logdet -= log(v);          // (every thread needs the total below: summed over the workgroup)
    }
    logdet = sm_bsum<T>(logdet, red, tid);
    __syncthreads();
  } else {
  // ---- Sigma (lower, packed) and sum(K) over the full matrix
    double ksum = 0.0;
    for (int e = tid; e < npair; e += T) {
      int i, j;
      sm_unpk(e, i, j);
      double sq = 0.0;
      for (int k = 0; k < D; ++k) {
        const double df = Xs[i * (SM_D + 1) + k] - Xs[j * (SM_D + 1) + k];
        sq = __builtin_fma(df, df, sq);
      }
      double kv = amp * ffgp_kfun_val(a.kfun, a.rinv, fmax(sq, a.clamp));
      ksum += (i == j) ? kv : 2.0 * kv;
      if (i == j) {
        kv += dadd;
        if (a.diag_vec) kv += a.diag_vec[(size_t)i * a.diag_stride];
      }
      if (a.add_mat) kv += a.add_mat[(size_t)i * a.ld_add + j];
      Sp[e] = kv + a.add_all;
    }
    if (a.mean_jitter != 0.0) {
      const double tot = sm_bsum<T>(ksum, red, tid);
      const double add = a.mean_jitter * tot / ((double)n * (double)n);
      __syncthreads();
      if (tid < n) Sp[sm_pk(tid, tid)] += add;
    }
    __syncthreads();

    // ---- Cholesky, right-looking; thread (ti, tk) of a 16 x 16 grid updates rows j+1+ti (+16..) x columns j+1+tk (+16..), k <= i.
    // ONE barrier per column: column j stays unscaled while its step runs (the update multiplies by 1 / d_jj itself) and is
    // scaled to L during step j + 1, when nobody reads it any more; 1 / L_jj by rsq + two Newton steps, the log-determinant
    // from the stored reciprocals afterwards.
    const int ti = tid / SM_TG, tk = tid % SM_TG;
    int bad = 0;
    double inv_prev = 0.0;
    for (int j = 0; j < n; ++j) {
      __syncthreads();                       // the previous column's trailing update is complete
      if (j > 0)
        for (int i = j + tid; i < n; i += T) Sp[sm_pk(i, j - 1)] *= inv_prev;     // column j - 1 becomes L's
      const double djj = Sp[sm_pk(j, j)];    // (the diagonal of L is never stored: the inverse below only needs 1 / L_jj)
      if (!(djj > 0.0) && bad == 0) bad = j + 1;
      double r = __builtin_amdgcn_rsq(djj);                       // 1 / sqrt(d_jj)
      r = __builtin_fma(0.5 * r, __builtin_fma(-djj * r, r, 1.0), r);
      r = __builtin_fma(0.5 * r, __builtin_fma(-djj * r, r, 1.0), r);
      const double inv2 = r * r;
      if (tid == 0) dinv[j] = r;
      inv_prev = r;
      for (int i = j + 1 + ti; i < n; i += SM_TG) {
        const int base = i * (i + 1) / 2;
        const double li = Sp[base + j] * inv2;
#pragma unroll 4
        for (int k = j + 1 + tk; k <= i; k += SM_TG) Sp[base + k] = __builtin_fma(-li, Sp[sm_pk(k, j)], Sp[base + k]);
      }
    }
    __syncthreads();
    if (tid == 0) {
      if (a.info_max) {
        if (bad) atomicMax(a.info, bad);     // one status word for a whole batch: any failing problem reports
      } else {
        a.info[0] = bad;
      }
    }
    {
      double ld = 0.0;
      for (int j = tid; j < n; j += T) ld -= log(dinv[j]);
      logdet = sm_bsum<T>(ld, red, tid);
    }

    // ---- L^-1 in place, row by row: X[i][j] = -(1 / L_ii) sum_{k = j}^{i-1} L[i][k] X[k][j]; 4 lanes share one j.  Row i of L is
    // copied aside first (its entries are overwritten by X's while other lanes still need them); the copy of row i + 1 is made
    // during row i's arithmetic into the other of two buffers: one barrier per row.
    {
      const int jq = tid >> 2, kp = tid & 3;          // T / 4 columns per pass: one pass
      double* rb[2] = {rowb, Gam};                    // (Gamma's storage is free until the inverse is complete)
      __syncthreads();
      for (int i = 0; i < n; ++i) {
        const int base = i * (i + 1) / 2;
        const double* rw = rb[i & 1];
        for (int j0 = 0; j0 < i; j0 += T / 4) {
          const int j = j0 + jq;
          double s = 0.0;
          if (j < i) {
#pragma unroll 4
            for (int k = j + kp; k < i; k += 4) s = __builtin_fma(rw[k], Sp[sm_pk(k, j)], s);
          }
          s += __shfl_xor(s, 1);
          s += __shfl_xor(s, 2);
          if (j < i && kp == 0) Sp[base + j] = -dinv[i] * s;
        }
        if (tid == 0) Sp[base + i] = dinv[i];
        if (i + 1 < n) {
          double* nx = rb[(i + 1) & 1];
          const int nb = (i + 1) * (i + 2) / 2;
          for (int k = tid; k <= i; k += T) nx[k] = Sp[nb + k];
        }
        __syncthreads();
      }
    }

}
  // ---- Gamma = L^-1 Y, A = L^-T Gamma
  for (int idx = tid; idx < n * d; idx += T) {
    const int i = idx / d, c = idx - i * d;
    const int base = i * (i + 1) / 2;
    double s = 0.0;
    for (int k = 0; k <= i; ++k) s = __builtin_fma(Sp[base + k], Ym[k * d + c], s);
    Gam[idx] = s;
  }
  __syncthreads();
  const bool needA = a.v2 || a.want_grad;
  if (needA) {
    for (int idx = tid; idx < n * d; idx += T) {
      const int i = idx / d, c = idx - i * d;
      double s = 0.0;
      for (int k = i; k < n; ++k) s = __builtin_fma(Sp[sm_pk(k, i)], Gam[k * d + c], s);
      Am[idx] = s;
    }
    __syncthreads();
  }
  // ---- value
  {
    const double* M = a.v2 ? Am : Gam;
    double ss = 0.0;
    for (int idx = tid; idx < n * d; idx += T) ss = __builtin_fma(M[idx], M[idx], ss);
    ss = sm_bsum<T>(ss, red, tid);
    const double oscale = (a.has_links && a.l.out_scale != 0.0) ? a.l.out_scale : 1.0;
    if (tid == 0) a.nll[0] = oscale * (0.5 * ss + (double)d * logdet + 0.5 * (double)n * (double)d * log(2.0 * a.pi_const));
  }
  if (!a.want_grad) return;

  // ---- Sigma^-1 = L^-T L^-1 (lower): every entry from the rows below it, all of them read before any is written
  {
    constexpr int NV = (SM_N * (SM_N + 1) / 2 + T - 1) / T;
    double vals[NV];
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int e = tid + T * c;
      double sacc = 0.0;
      if (e < npair) {
        int i, j;
        sm_unpk(e, i, j);
#pragma unroll 4
        for (int k = i; k < n; ++k) sacc = __builtin_fma(Sp[sm_pk(k, i)], Sp[sm_pk(k, j)], sacc);
      }
      vals[c] = sacc;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int e = tid + T * c;
      if (e < npair) Sp[e] = vals[c];
    }
    __syncthreads();
  }
  if (a.v2) {   // B = Sigma^-1 A
    for (int idx = tid; idx < n * d; idx += T) {
      const int i = idx / d, c = idx - i * d;
      double s = 0.0;
      for (int k = 0; k < n; ++k) s = __builtin_fma((k <= i) ? Sp[sm_pk(i, k)] : Sp[sm_pk(k, i)], Am[k * d + c], s);
      Bm[idx] = s;
    }
    __syncthreads();
  }
  // ---- G = d/2 Sigma^-1 - 1/2 A A^T   (V2: - 1/2 (A B^T + B A^T)), in place; trace and diagonal
  double tr = 0.0;
  for (int e = tid; e < npair; e += T) {
    int i, j;
    sm_unpk(e, i, j);
    double s = 0.0;
    if (!a.v2) {
      for (int c = 0; c < d; ++c) s = __builtin_fma(Am[i * d + c], Am[j * d + c], s);
    } else {
      for (int c = 0; c < d; ++c) s += Am[i * d + c] * Bm[j * d + c] + Bm[i * d + c] * Am[j * d + c];
    }
    const double gv = 0.5 * (double)d * Sp[e] - 0.5 * s;
    Sp[e] = gv;
    if (i == j) tr += gv;
  }
  const double trG = sm_bsum<T>(tr, red, tid);
  __syncthreads();
  const double oscale = (a.has_links && a.l.out_scale != 0.0) ? a.l.out_scale : 1.0;
  if (a.g_diag_vec && tid < n) a.g_diag_vec[tid] = oscale * Sp[sm_pk(tid, tid)];
  if (a.g_Y) {
    const double* M = a.v2 ? Bm : Am;
    for (int idx = tid; idx < n * d; idx += T) a.g_Y[idx] = oscale * M[idx];
  }
  // ---- kernel-parameter gradients: the reductions of grad.hip over the lower triangle (off-diagonal entries count twice)
  const double geff_add = (a.mean_jitter != 0.0) ? a.mean_jitter / ((double)n * (double)n) * trG : 0.0;
  double s_amp = 0.0, s_kp = 0.0, tot[SM_D];
#pragma unroll
  for (int k = 0; k < SM_D; ++k) tot[k] = 0.0;
  if (a.g_w || a.g_amp || a.g_kparam) {
    for (int e = tid; e < npair; e += T) {
      int i, j;
      sm_unpk(e, i, j);
      double df[SM_D], sq = 0.0;
#pragma unroll
      for (int k = 0; k < SM_D; ++k) {
        df[k] = (k < D) ? Xs[i * (SM_D + 1) + k] - Xs[j * (SM_D + 1) + k] : 0.0;
        sq = __builtin_fma(df[k], df[k], sq);
      }
      const double g = Sp[e] + geff_add, sym = (i == j) ? 1.0 : 2.0;
      const double scl = fmax(sq, a.clamp);
      const double ev = ffgp_kfun_val(a.kfun, a.rinv, scl);
      s_amp += sym * g * ev;
      if (a.kfun == FFGP_KFUN_RQ) s_kp += sym * g * amp * ffgp_kfun_dparam(a.kfun, a.rinv, scl, ev);
      const double wl = (sq >= a.clamp) ? sym * g * amp * ffgp_kfun_m2d(a.kfun, a.rinv, scl) : 0.0;
#pragma unroll
      for (int k = 0; k < SM_D; ++k) tot[k] = __builtin_fma(wl * df[k], df[k], tot[k]);
    }
  }
  s_amp = sm_bsum<T>(s_amp, red, tid);
  s_kp = sm_bsum<T>(s_kp, red, tid);
  double gw_eff = 0.0;       // thread k < D ends up with the effective-w gradient of dimension k
#pragma unroll
  for (int k = 0; k < SM_D; ++k) {
    const double t = sm_bsum<T>(tot[k], red, tid);
    if (tid == k && k < D) gw_eff = -t / wv[k];
  }
  // ---- chain to the raw parameters (identity links otherwise), output scale
  if (a.g_w) {
    if (a.has_links && a.l.w_broadcast) {
      double v = (tid < D) ? gw_eff : 0.0;
      v = sm_bsum<T>(v, red, tid);
      if (tid == 0) a.g_w[0] = oscale * v * sm_link_der(a.l.w_link, a.w[0], a.l.w_c);
    } else if (tid < D) {
      a.g_w[tid] = oscale * gw_eff * (a.has_links ? sm_link_der(a.l.w_link, a.w[tid], a.l.w_c) : 1.0);
    }
  }
  if (tid == 0) {
    if (a.g_amp) a.g_amp[0] = oscale * s_amp * (a.has_links ? sm_link_der(a.l.amp_link, a.amp[0], a.l.amp_c) : 1.0);
    if (a.g_dadd) a.g_dadd[0] = oscale * trG * ((a.has_links && a.dadd) ? sm_link_der(a.l.dadd_link, a.dadd[0], a.l.dadd_c) : 1.0);
    if (a.g_kparam) a.g_kparam[0] = oscale * s_kp;
  }
}

template <bool FROM_FACTOR>
__global__ __launch_bounds__(FROM_FACTOR ? SM_T2 : SM_T) void ffgp_small_nlml_kernel(SmallArgs a) {
  sm_body<FROM_FACTOR>(a);
}

// F independent problems in one launch (the per-fidelity / per-seed loops of Experiments/GAR_Aligned/exp_aligned.py:58-126 run
// models of this size one after the other): workgroup f does problem f, each on its own CU
#define SM_BATCH 8
struct SmallBatch {
  SmallArgs a[SM_BATCH];
};
__global__ __launch_bounds__(SM_T) void ffgp_small_batch_kernel(SmallBatch b) {
  sm_body<false>(b.a[blockIdx.x]);
}

#define SM_LDS_DOUBLES (SM_N * (SM_N + 1) / 2 + SM_N * (SM_D + 1) + 4 * SM_N * SM_Y + 2 * SM_N + SM_D + 8 + SM_T2 / 64)

// does the "blocked factorisation + one finishing kernel" path cover this call?  (one diagonal block: 40 < n <= 128; option
// "small_finish", off by default: tools/small_finish_bench.py measured a training step at n = 64 / 80 / 96 / 112 / 128 at
// 0.196 / 0.202 / 0.209 / 0.220 / 0.234 ms with it against 0.216 / 0.219 / 0.219 / 0.222 / 0.227 ms on the 21-launch path -- the
// single-workgroup finishing kernel takes about as long as the launches it saves)
bool ffgp_small2_ok(const ffgp_handle* h, const ffgp_problem* p, const ffgp_grads* g) {
  if (h->small_off || h->small2_off || p->n > SM_N || p->D > SM_D || p->d > SM_Y || p->cov_dev || p->pair || p->tree) return false;
  if (p->kfun < FFGP_KFUN_SE || p->kfun > FFGP_KFUN_RQ) return false;
  if (g && (g->g_cov_dev || g->g_pair)) return false;
  return !h->use_naive && h->timing == 0;
}

// does the one-kernel path cover this call?
bool ffgp_small_ok(const ffgp_handle* h, const ffgp_problem* p, const ffgp_grads* g) {
  if (h->small_off) return false;
  if (p->n > (h->small_max_n > 0 ? h->small_max_n : SM_MAX_FAST_N) || p->n > SM_N || p->D > SM_D || p->d > SM_Y || p->cov_dev || p->pair || p->tree)
    return false;
  if (p->kfun < FFGP_KFUN_SE || p->kfun > FFGP_KFUN_RQ) return false;
  if (g && (g->g_cov_dev || g->g_pair)) return false;
  return true;
}

// enqueue the kernel; the caller finishes like the blocked path (sticky info, D2H of the status word)
static void sm_fill(SmallArgs& a, ffgp_handle* h, const ffgp_problem* p, const ffgp_links* l, double* nll_dev, const ffgp_grads* g,
                    const double* dinv);

static int sm_attr(ffgp_handle* h) {
  static bool attr_set[64] = {false};
  if (h->device >= 0 && h->device < 64 && !attr_set[h->device]) {
    FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_small_nlml_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 SM_LDS_DOUBLES * (int)sizeof(double)));
    FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_small_nlml_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 SM_LDS_DOUBLES * (int)sizeof(double)));
    FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_small_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 SM_LDS_DOUBLES * (int)sizeof(double)));
    attr_set[h->device] = true;
  }
  return FFGP_OK;
}

int ffgp_small_enqueue(ffgp_handle* h, const ffgp_problem* p, const ffgp_links* l, double* nll_dev, const ffgp_grads* g, const double* dinv) {
  FFGP_CHECK(sm_attr(h));
  SmallArgs a;
  sm_fill(a, h, p, l, nll_dev, g, dinv);
  if (dinv) hipLaunchKernelGGL(ffgp_small_nlml_kernel<true>, dim3(1), dim3(SM_T2), SM_LDS_DOUBLES * sizeof(double), h->stream, a);
  else hipLaunchKernelGGL(ffgp_small_nlml_kernel<false>, dim3(1), dim3(SM_T), SM_LDS_DOUBLES * sizeof(double), h->stream, a);
  return hipGetLastError() == hipSuccess ? FFGP_OK : FFGP_ERR_HIP;
}

static void sm_fill(SmallArgs& a, ffgp_handle* h, const ffgp_problem* p, const ffgp_links* l, double* nll_dev, const ffgp_grads* g,
                    const double* dinv) {
  a.n = p->n; a.D = p->D; a.d = p->d;
  a.X = p->X_dev; a.Y = p->Y_dev; a.w = p->w_dev; a.amp = p->amp_dev; a.dadd = p->diag_add_dev;
  a.has_links = l ? 1 : 0;
  if (l) a.l = *l; else memset(&a.l, 0, sizeof(a.l));
  a.clamp = p->clamp_min; a.kfun = p->kfun; a.rinv = (p->kparam != 0.0) ? 1.0 / p->kparam : 1.0;
  a.diag_vec = p->diag_vec_dev; a.diag_stride = p->diag_stride; a.add_mat = p->add_mat_dev; a.ld_add = p->ld_add;
  a.add_all = p->add_all; a.mean_jitter = p->mean_jitter;
  a.pi_const = p->pi_const; a.v2 = (p->ll_variant == FFGP_LL_V2) ? 1 : 0;
  a.nll = nll_dev;
  a.g_w = g ? g->g_w_dev : nullptr; a.g_amp = g ? g->g_amp_dev : nullptr; a.g_dadd = g ? g->g_diag_add_dev : nullptr;
  a.g_Y = g ? g->g_Y_dev : nullptr; a.g_diag_vec = g ? g->g_diag_vec_dev : nullptr; a.g_kparam = g ? g->g_kparam_dev : nullptr;
  a.want_grad = (a.g_w || a.g_amp || a.g_dadd || a.g_Y || a.g_diag_vec || a.g_kparam) ? 1 : 0;
  a.info = h->d_info;
  a.info_max = 0;
  a.dinv = dinv;
}

// F problems (each must pass ffgp_small_batch_ok) in ceil(F / 8) launches; the caller zeroes the status word first and finishes
// like the other paths (sticky info, D2H of the status word)
int ffgp_small_batch_enqueue(ffgp_handle* h, int F, const ffgp_problem* p, const ffgp_links* l, double* nll_dev, const ffgp_grads* g) {
  FFGP_CHECK(sm_attr(h));
  for (int f0 = 0; f0 < F; f0 += SM_BATCH) {
    SmallBatch b;
    const int nb = (F - f0 < SM_BATCH) ? F - f0 : SM_BATCH;
    for (int f = 0; f < nb; ++f) {
      sm_fill(b.a[f], h, p + f0 + f, l ? l + f0 + f : nullptr, nll_dev + f0 + f, g ? g + f0 + f : nullptr, nullptr);
      b.a[f].info_max = 1;
    }
    for (int f = nb; f < SM_BATCH; ++f) b.a[f] = b.a[0];
    hipLaunchKernelGGL(ffgp_small_batch_kernel, dim3(nb), dim3(SM_T), SM_LDS_DOUBLES * sizeof(double), h->stream, b);
  }
  return hipGetLastError() == hipSuccess ? FFGP_OK : FFGP_ERR_HIP;
}

bool ffgp_small_batch_ok(const ffgp_problem* p, const ffgp_grads* g) {
  if (p->n <= 0 || p->n > SM_N || p->D <= 0 || p->D > SM_D || p->d <= 0 || p->d > SM_Y || p->cov_dev || p->pair || p->tree) return false;
  if (!p->X_dev || !p->Y_dev || !p->w_dev || !p->amp_dev) return false;
  if (p->kfun < FFGP_KFUN_SE || p->kfun > FFGP_KFUN_RQ) return false;
  if (p->ll_variant != FFGP_LL_V1 && p->ll_variant != FFGP_LL_V2) return false;
  if (g && (g->g_cov_dev || g->g_pair)) return false;
  return true;
}
