// K Adam steps of F small models in ONE launch (round 6) -- the reference's hot loop at the sizes its experiments run
// (FidelityFusion_Models/ResGP.py:78-112, GaussianProcess/cigp_v10.py:92-104: per fidelity 100-1000 iterations of zero_grad / loss =
// -negative_log_likelihood / backward / Adam step at N = 16 ... 128; Experiments/GAR_Aligned/exp_aligned.py:66-74: 100 low- against 4..32
// high-fidelity points).  ffgp_train_raw used to enqueue 13 dependent launches per step at n = 128 (0.096 ms per step, two thirds of
// it launch floors).  Here one PERSISTENT workgroup per model (gridDim.x = models, 512 threads) runs every step inside the kernel:
//
//   links (raw -> effective parameters)  ->  Sigma assembled straight into LDS as 16 x 16 blocks [16][17] (lower block triangle, 78 KiB)
//   ->  blocked Cholesky: the 16 x 16 diagonal block is factored AND inverted in registers by wave 0 on the DP-ALU DPP pivot step of the
//       diagonal-block kernel (f16_steps.h), the blocks below are solved with that inverse and the trailing blocks updated on the matrix
//       cores by all eight waves; only ceil(n / 16) stages run, so n = 32 costs a quarter of n = 128
//   ->  L^-1 in place by recursive doubling on the matrix cores  ->  Gamma = L^-1 Y, A = L^-T Gamma, the value
//   ->  Sigma^-1 = L^-T L^-1 block by block on the matrix cores, each 16 x 16 block consumed in its accumulators: G = d/2 Sigma^-1 - 1/2 A A^T,
//       the kernel re-evaluated for the entry, the gradient sums of grad.hip (amplitude, length scales, trace) -- Sigma^-1 is never stored
//   ->  the links' chain rule and torch.optim.Adam's update (operation for operation as ffgp_adam_kernel) on parameters and moments that
//       live in LDS for the whole call; the step's loss goes to the trace.
// A Sigma that is not positive definite stops THAT model at that step (its status word, NaN in its trace from there on, parameters as
// they were when the step began); the other models of the launch train on.
// Covers: n <= 128, D <= 16, d <= 16, one radial-profile kernel, V1 likelihood, diag_add and diag_vec (no matrix / all-entries /
// mean(K) extras, no learnable profile parameter): what cigp_v10.train_many produces.  Everything else stays on the launch-per-stage path.
#include "ffgp_internal.h"
#include "f16_steps.h"

#define TR_T 512
#define TR_BLD 17
#define TR_BLK (16 * TR_BLD)
#define TR_NBLK 36
#define TR_N 128
#define TR_D 16
#define TR_Y 16
#define TR_NRED 20        // values of the step's one workgroup reduction: ss, s_amp, tr, tot[16], spare

typedef double tr_d4 __attribute__((ext_vector_type(4)));

struct TrainModel {
  int n, D, d, nw;                       // nw: raw length scales (1 = one value broadcast over the D dimensions)
  const double* X; const double* Y;
  double* w; double* amp; double* dadd;  // RAW parameters, updated in place when the kernel ends
  const double* diag_vec; long diag_stride;
  ffgp_links l;
  double clamp, rinv, pi_const;
  int kfun;
  double* state;                         // [exp_avg (nw + 2) | exp_avg_sq (nw + 2)]
  double* trace;                         // [steps]
};
struct TrainCommon {
  int steps;
  double lr, b1, b2, eps;
  const double* bc;                      // [steps][2]: 1 - beta1^t, sqrt(1 - beta2^t) -- computed on the host with the C library's pow, as Python does
  int* info;                             // [models] status: 0, or the 1-based index of the first non-positive pivot of the step that failed
  int* fail_step;                        // [models] the step at which it happened
};

__device__ __forceinline__ double tr_link_val(int kind, double p, double c) {
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
__device__ __forceinline__ double tr_link_der(int kind, double p, double c) {
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

__device__ __forceinline__ int tr_blk(int bi, int bj) { return (bi * (bi + 1) / 2 + bj) * TR_BLK; }
// t-th block of the row-major enumeration of the lower block triangle -> (bi, bj)
__device__ __forceinline__ void tr_unblk(int t, int& bi, int& bj) {
  bi = 0;
#pragma unroll
  for (int q = 1; q < 8; ++q) bi += (t >= q * (q + 1) / 2) ? 1 : 0;
  bj = t - bi * (bi + 1) / 2;
}
__device__ __forceinline__ double tr_rsqrt(double d) {
  double y = __builtin_amdgcn_rsq(d);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const double e = __builtin_fma(-d * y, y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
  }
  return y;
}
// acc += P * Q^T on 16 x 16 blocks of the LDS image: P[m][k] at pa[m * 17 + k], Q[n][k] at pb[n * 17 + k]
__device__ __forceinline__ void tr_mma_nt(tr_d4& acc, const double* pa, const double* pb, int lane) {
#pragma unroll
  for (int kq = 0; kq < 4; ++kq) {
    const int k = kq * 4 + (lane >> 4);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[(lane & 15) * TR_BLD + k], pb[(lane & 15) * TR_BLD + k], acc, 0, 0, 0);
  }
}
// acc += P * Q: P[m][k] at pa[m * 17 + k], Q[k][n] at pb[k * 17 + n]
__device__ __forceinline__ void tr_mma_nn(tr_d4& acc, const double* pa, const double* pb, int lane) {
#pragma unroll
  for (int kq = 0; kq < 4; ++kq) {
    const int k = kq * 4 + (lane >> 4);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[(lane & 15) * TR_BLD + k], pb[k * TR_BLD + (lane & 15)], acc, 0, 0, 0);
  }
}
// acc += P^T * Q: P[k][m] at pa[k * 17 + m], Q[k][n] at pb[k * 17 + n]
__device__ __forceinline__ void tr_mma_tn(tr_d4& acc, const double* pa, const double* pb, int lane) {
#pragma unroll
  for (int kq = 0; kq < 4; ++kq) {
    const int k = kq * 4 + (lane >> 4);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k * TR_BLD + (lane & 15)], pb[k * TR_BLD + (lane & 15)], acc, 0, 0, 0);
  }
}
// accumulator (lane (g, c), register r = entry (g + 4 r, c)) -> the block's [16][17] home
__device__ __forceinline__ void tr_store(double* dst, const tr_d4& acc, int g, int c, double sign) {
#pragma unroll
  for (int r = 0; r < 4; ++r) dst[(g + 4 * r) * TR_BLD + c] = sign * acc[r];
}

// one level of the in-place inversion by recursive doubling (as inv_merge_level of potrf.hip): pairs of inverted diagonal super-blocks
// of S_ blocks are merged, X21 = -X22 (L21 X11).  One (pair, block column) item per wave (4 items per level); a wave keeps its
// column of T = L21 X11 in registers across the barrier that protects L21 from being overwritten while other waves still read it.
template <int S_>
__device__ __forceinline__ void tr_inv_level(double* S, int wave, int lane) {
  const bool act = wave < 4;
  const int pair = wave / S_, jl = wave % S_;
  const int b0 = pair * 2 * S_;
  const int j = b0 + jl;
  const int g = lane >> 4, c = lane & 15;
  tr_d4 T[S_];
  if (act) {
#pragma unroll
    for (int ii = 0; ii < S_; ++ii) {
      const int i = b0 + S_ + ii;
      tr_d4 acc = {0.0, 0.0, 0.0, 0.0};
      for (int k = j; k < b0 + S_; ++k) tr_mma_nn(acc, S + tr_blk(i, k), S + tr_blk(k, j), lane);
      T[ii] = acc;
    }
  }
  __syncthreads();
  if (act) {
#pragma unroll
    for (int ii = 0; ii < S_; ++ii) tr_store(S + tr_blk(b0 + S_ + ii, j), T[ii], g, c, 1.0);
    tr_d4 R[S_];
#pragma unroll
    for (int ii = 0; ii < S_; ++ii) {      // (reads column j of the rows the same wave just wrote: its own LDS stores, in order)
      const int i = b0 + S_ + ii;
      tr_d4 acc = {0.0, 0.0, 0.0, 0.0};
      for (int k = b0 + S_; k <= i; ++k) tr_mma_nn(acc, S + tr_blk(i, k), S + tr_blk(k, j), lane);
      R[ii] = acc;
    }
#pragma unroll
    for (int ii = 0; ii < S_; ++ii) tr_store(S + tr_blk(b0 + S_ + ii, j), R[ii], g, c, -1.0);
  }
  __syncthreads();
}

// LDS (doubles): S 36 * 272 | Xs [128][17] | Ym, Gam, Am [128][16] each | piv [128] | dvec [128] | small
#define TR_OFF_XS (TR_NBLK * TR_BLK)
#define TR_OFF_YM (TR_OFF_XS + TR_N * (TR_D + 1))
#define TR_OFF_GAM (TR_OFF_YM + TR_N * TR_Y)
#define TR_OFF_AM (TR_OFF_GAM + TR_N * TR_Y)
#define TR_OFF_PIV (TR_OFF_AM + TR_N * TR_Y)
#define TR_OFF_DVEC (TR_OFF_PIV + TR_N)
#define TR_OFF_SMALL (TR_OFF_DVEC + TR_N)
#define TR_SMALL_DOUBLES (16 + 3 * 20 + 8 + 8 * TR_NRED + TR_NRED + 8)
#define TR_LDS_DOUBLES (TR_OFF_SMALL + TR_SMALL_DOUBLES)

__global__ __launch_bounds__(TR_T) void ffgp_train_persist_kernel(const TrainModel* __restrict__ tab, TrainCommon cm) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const TrainModel M = tab[blockIdx.x];
  double* S = lds;
  double* Xs = lds + TR_OFF_XS;
  double* Ym = lds + TR_OFF_YM;
  double* Gam = lds + TR_OFF_GAM;
  double* Am = lds + TR_OFF_AM;
  double* piv = lds + TR_OFF_PIV;
  double* dvec = lds + TR_OFF_DVEC;
  double* wv = lds + TR_OFF_SMALL;         // [16] effective inverse length scales
  double* raw = wv + 16;                   // [20] raw parameters: w (nw) | amp | dadd
  double* mom = raw + 20;                  // [20] exp_avg
  double* mo2 = mom + 20;                  // [20] exp_avg_sq
  double* sc = mo2 + 20;                   // [8]  amp, dadd, logdet
  double* red = sc + 8;                    // [8][TR_NRED] per-wave partial sums
  double* tot = red + 8 * TR_NRED;         // [TR_NRED] the step's totals
  int* flags = reinterpret_cast<int*>(tot + TR_NRED);     // [0] bad pivot of the current step
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int n = M.n, D = M.D, d = M.d, nw = M.nw;
  const int nst = (n + 15) >> 4, nblk = nst * (nst + 1) / 2;
  const int npar = nw + 2;
  const double oscale = (M.l.out_scale != 0.0) ? M.l.out_scale : 1.0;
  ExpCoef ec;
  ffgp_exp_load(ec);

  // ---- once: targets, the diagonal extra, parameters and moments into LDS; identity padding of the blocks the factorisation never touches
  for (int idx = tid; idx < n * d; idx += TR_T) Ym[idx] = M.Y[idx];
  for (int i = tid; i < TR_N; i += TR_T) dvec[i] = (M.diag_vec && i < n) ? M.diag_vec[(size_t)i * M.diag_stride] : 0.0;
  if (tid < npar) {
    raw[tid] = (tid < nw) ? M.w[tid] : (tid == nw ? M.amp[0] : M.dadd[0]);
    mom[tid] = M.state[tid];
    mo2[tid] = M.state[npar + tid];
  }
  for (int t = wave; t < TR_NBLK; t += 8) {
    int bi, bj;
    tr_unblk(t, bi, bj);
    if (bi < nst) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) S[tr_blk(bi, bj) + (g + 4 * r) * TR_BLD + c] = (bi == bj && g + 4 * r == c) ? 1.0 : 0.0;
  }
  if (tid == 0) flags[0] = 0;
  __syncthreads();

  int failed = 0;
  for (int step = 0; step < cm.steps; ++step) {
    // ---- P0: effective parameters, scaled inputs
    if (tid < D) wv[tid] = tr_link_val(M.l.w_link, raw[M.l.w_broadcast ? 0 : tid], M.l.w_c);
    if (tid == 64) sc[0] = tr_link_val(M.l.amp_link, raw[nw], M.l.amp_c);
    if (tid == 65) sc[1] = tr_link_val(M.l.dadd_link, raw[nw + 1], M.l.dadd_c);
    __syncthreads();
    for (int idx = tid; idx < n * D; idx += TR_T) {
      const int i = idx / D, k = idx - i * D;
      Xs[i * (TR_D + 1) + k] = M.X[idx] * wv[k];
    }
    __syncthreads();
    const double amp = sc[0], dadd = sc[1];

    // ---- P1: Sigma, lower block triangle (diagonal blocks symmetric-full: the in-register factor wants both halves); rows / columns
    //      beyond n are identity
    for (int t = wave; t < nblk; t += 8) {
      int bi, bj;
      tr_unblk(t, bi, bj);
      double* dst = S + tr_blk(bi, bj);
      const int j = bj * 16 + c;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = bi * 16 + g + 4 * r;
        double kv = (i == j) ? 1.0 : 0.0;
        if (i < n && j < n) {
          double sq = 0.0;
          for (int k = 0; k < D; ++k) {
            const double df = Xs[i * (TR_D + 1) + k] - Xs[j * (TR_D + 1) + k];
            sq = __builtin_fma(df, df, sq);
          }
          const double s_ = fmax(sq, M.clamp);
          kv = amp * ((M.kfun == FFGP_KFUN_SE) ? ffgp_exp_fast(-0.5 * s_, ec) : ffgp_kfun_val(M.kfun, M.rinv, s_));
          if (i == j) kv += dadd + dvec[i];
        }
        dst[(g + 4 * r) * TR_BLD + c] = kv;
      }
    }
    __syncthreads();

    // ---- P2: blocked Cholesky over 16-column stages; the diagonal block's slot receives inv(L_jj), the pivots go to piv[]
    for (int jj = 0; jj < nst; ++jj) {
      if (wave == 0) {
        double* Dj = S + tr_blk(jj, jj);
        int cc = c, gg = g;
        asm volatile("" : "+v"(cc), "+v"(gg));      // (opaque per iteration: the stage loop must not be specialised per jj)
        double v[4], w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = Dj[(gg + 4 * r) * TR_BLD + cc];
          w[r] = (gg + 4 * r == cc) ? 1.0 : 0.0;
        }
        double rowA = bperm_d(v[0], cc);
        double rowW = (cc == 0) ? 1.0 : 0.0;
        {
          double hA = bperm_d(v[0], 16 + cc), hW = (cc == 1) ? 1.0 : 0.0;
          double pRow = 0.0, pt = 0.0, ptw = 0.0;
          double dcur = row_bcast64<0>(rowA), ycur = __builtin_amdgcn_rcp(dcur);
#define TR_F16(JJ) f16_step_dpp<JJ>(v, w, rowA, rowW, hA, hW, pRow, pt, ptw, dcur, ycur, cc, gg);
          TR_F16(0) TR_F16(1) TR_F16(2) TR_F16(3) TR_F16(4) TR_F16(5) TR_F16(6) TR_F16(7) TR_F16(8) TR_F16(9) TR_F16(10) TR_F16(11)
          TR_F16(12) TR_F16(13) TR_F16(14) TR_F16(15)
#undef TR_F16
        }
        const int q = c >> 2;
        const double dsel = (q == 0) ? v[0] : (q == 1) ? v[1] : (q == 2) ? v[2] : v[3];
        const double dcol = bperm_d(dsel, 16 * (c & 3) + c);      // pivot of column c
        const double rs = tr_rsqrt(dcol);
        const unsigned long long nonpos = __ballot(!(dcol > 0.0)) & 0xffffull;
        const int bad = nonpos ? __ffsll((long long)nonpos) : 0;
        if (g == 0) piv[jj * 16 + c] = dcol;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = g + 4 * r;
          const double rsi = bperm_d(rs, i);
          Dj[i * TR_BLD + c] = (i >= c) ? w[r] * rsi : 0.0;       // inv(L_jj)
        }
        if (bad && lane == 0 && jj * 16 + bad <= n && flags[0] == 0) flags[0] = jj * 16 + bad;
      }
      __syncthreads();
      if (jj + 1 >= nst) break;
      // solve: L[i][jj] = S[i][jj] inv(L_jj)^T for the block rows below
      for (int i = jj + 1 + wave; i < nst; i += 8) {
        tr_d4 acc = {0.0, 0.0, 0.0, 0.0};
        tr_mma_nt(acc, S + tr_blk(i, jj), S + tr_blk(jj, jj), lane);
        tr_store(S + tr_blk(i, jj), acc, g, c, 1.0);
      }
      __syncthreads();
      // update: S[i][k] -= L[i][jj] L[k][jj]^T for jj < k <= i
      const int m = nst - 1 - jj;
      for (int t = wave; t < m * (m + 1) / 2; t += 8) {
        int a, b;
        tr_unblk(t, a, b);
        const int i = jj + 1 + a, k = jj + 1 + b;
        tr_d4 acc = {0.0, 0.0, 0.0, 0.0};
        tr_mma_nt(acc, S + tr_blk(i, jj), S + tr_blk(k, jj), lane);
        double* dst = S + tr_blk(i, k);
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(g + 4 * r) * TR_BLD + c] -= acc[r];
      }
      __syncthreads();
    }
    if (flags[0] != 0) {       // (uniform: every thread reads the same word behind the barrier)
      failed = flags[0];
      if (tid == 0) {
        cm.info[blockIdx.x] = failed;
        cm.fail_step[blockIdx.x] = step;
      }
      for (int k = step + tid; k < cm.steps; k += TR_T) M.trace[k] = __builtin_nan("");
      break;
    }
    // ---- L^-1 in place
    if (nst > 1) tr_inv_level<1>(S, wave, lane);
    if (nst > 2) tr_inv_level<2>(S, wave, lane);
    if (nst > 4) tr_inv_level<4>(S, wave, lane);

    // ---- P3: Gamma = W Y, A = W^T Gamma (W = L^-1, lower; four lanes share an output, block columns dealt round robin)
    {
      const int q4 = tid & 3;
      for (int it = tid >> 2; it < n * d; it += TR_T / 4) {
        const int i = it / d, cc2 = it - i * d;
        const int bi = i >> 4;
        double s = 0.0;
        for (int kb = q4; kb <= bi; kb += 4) {
          const double* wr = S + tr_blk(bi, kb) + (i & 15) * TR_BLD;
#pragma unroll
          for (int k = 0; k < 16; ++k) s = __builtin_fma(wr[k], (kb * 16 + k < n) ? Ym[(kb * 16 + k) * d + cc2] : 0.0, s);
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (q4 == 0) Gam[it] = s;
      }
      __syncthreads();
      for (int it = tid >> 2; it < n * d; it += TR_T / 4) {
        const int i = it / d, cc2 = it - i * d;
        const int bi = i >> 4;
        double s = 0.0;
        for (int kb = bi + q4; kb < nst; kb += 4) {
          const double* wc = S + tr_blk(kb, bi) + (i & 15);
#pragma unroll
          for (int k = 0; k < 16; ++k) s = __builtin_fma(wc[k * TR_BLD], (kb * 16 + k < n) ? Gam[(kb * 16 + k) * d + cc2] : 0.0, s);
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (q4 == 0) Am[it] = s;
      }
      __syncthreads();
    }

    // ---- P5: per lane partial sums -- ss (value), s_amp, tr G, tot[k] (length scales); Sigma^-1 block by block on the matrix cores
    double ss = 0.0, s_amp = 0.0, trg = 0.0, tk[TR_D];
#pragma unroll
    for (int k = 0; k < TR_D; ++k) tk[k] = 0.0;
    for (int idx = tid; idx < n * d; idx += TR_T) ss = __builtin_fma(Gam[idx], Gam[idx], ss);
    for (int t = wave; t < nblk; t += 8) {
      int bi, bj;
      tr_unblk(t, bi, bj);
      tr_d4 acc = {0.0, 0.0, 0.0, 0.0};
      for (int kb = bi; kb < nst; ++kb) tr_mma_tn(acc, S + tr_blk(kb, bi), S + tr_blk(kb, bj), lane);
      const int j = bj * 16 + c;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = bi * 16 + g + 4 * r;
        if (i < n && j <= i) {
          double aa = 0.0;
          for (int q = 0; q < d; ++q) aa = __builtin_fma(Am[i * d + q], Am[j * d + q], aa);
          const double gv = 0.5 * (double)d * acc[r] - 0.5 * aa;
          const double sym = (i == j) ? 1.0 : 2.0;
          double df[TR_D], sq = 0.0;
#pragma unroll
          for (int k = 0; k < TR_D; ++k) {
            df[k] = (k < D) ? Xs[i * (TR_D + 1) + k] - Xs[j * (TR_D + 1) + k] : 0.0;
            sq = __builtin_fma(df[k], df[k], sq);
          }
          const double s_ = fmax(sq, M.clamp);
          double ev, m2;
          if (M.kfun == FFGP_KFUN_SE) {
            ev = ffgp_exp_fast(-0.5 * s_, ec);
            m2 = ev;
          } else {
            ev = ffgp_kfun_val(M.kfun, M.rinv, s_);
            m2 = ffgp_kfun_m2d(M.kfun, M.rinv, s_);
          }
          s_amp = __builtin_fma(sym * gv, ev, s_amp);
          if (i == j) trg += gv;
          const double wl = (sq >= M.clamp) ? sym * gv * amp * m2 : 0.0;
#pragma unroll
          for (int k = 0; k < TR_D; ++k) tk[k] = __builtin_fma(wl * df[k], df[k], tk[k]);
        }
      }
    }
    // ---- one workgroup reduction for all of them
    {
      double vals[TR_NRED];
      vals[0] = ss; vals[1] = s_amp; vals[2] = trg;
#pragma unroll
      for (int k = 0; k < TR_D; ++k) vals[3 + k] = tk[k];
      vals[19] = 0.0;
#pragma unroll
      for (int q = 0; q < 3 + TR_D; ++q) {
        double x = vals[q];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
        vals[q] = x;
      }
      if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 3 + TR_D; ++q) red[wave * TR_NRED + q] = vals[q];
      }
      __syncthreads();
      if (tid < 3 + TR_D) {
        double x = 0.0;
#pragma unroll
        for (int wv_ = 0; wv_ < 8; ++wv_) x += red[wv_ * TR_NRED + tid];
        tot[tid] = x;
      }
      if (tid == 64) {
        double ld = 0.0;
        for (int i = 0; i < n; ++i) ld += log(piv[i]);
        sc[2] = 0.5 * ld;                                   // sum_i log L_ii
      }
      __syncthreads();
    }
    // ---- P6: the loss of this step (before the update), the raw gradients through the links, Adam
    if (tid == 0)
      M.trace[step] = oscale * (0.5 * tot[0] + (double)d * sc[2] + 0.5 * (double)n * (double)d * log(2.0 * M.pi_const));
    if (tid < npar) {
      double gr;
      if (tid < nw) {
        if (!M.l.w_broadcast) {
          gr = oscale * (-tot[3 + tid] / wv[tid]) * tr_link_der(M.l.w_link, raw[tid], M.l.w_c);
        } else {
          double sg = 0.0;
          for (int k = 0; k < D; ++k) sg += -tot[3 + k] / wv[k];
          gr = oscale * sg * tr_link_der(M.l.w_link, raw[0], M.l.w_c);
        }
      } else if (tid == nw) {
        gr = oscale * tot[1] * tr_link_der(M.l.amp_link, raw[nw], M.l.amp_c);
      } else {
        gr = oscale * tot[2] * tr_link_der(M.l.dadd_link, raw[nw + 1], M.l.dadd_c);
      }
      const double bc1 = cm.bc[2 * step], bc2s = cm.bc[2 * step + 1];
      const double m1 = mom[tid] + (gr - mom[tid]) * (1.0 - cm.b1);          // exp_avg.lerp_(grad, 1 - beta1)
      const double v1 = mo2[tid] * cm.b2 + (1.0 - cm.b2) * gr * gr;           // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
      mom[tid] = m1;
      mo2[tid] = v1;
      const double denom = sqrt(v1) / bc2s + cm.eps;
      raw[tid] = raw[tid] + (-(cm.lr / bc1)) * (m1 / denom);                  // param.addcdiv_(exp_avg, denom, value = -step_size)
    }
    __syncthreads();
  }
  // ---- parameters and moments back to the caller's tensors (a failed step left them as they were when it began)
  if (tid < npar) {
    if (tid < nw) M.w[tid] = raw[tid];
    else if (tid == nw) M.amp[0] = raw[tid];
    else M.dadd[0] = raw[tid];
    M.state[tid] = mom[tid];
    M.state[npar + tid] = mo2[tid];
  }
  (void)failed;
}

// ---- host side ------------------------------------------------------------------------------------------------------
bool ffgp_train_persist_ok(const ffgp_handle* h, const ffgp_problem* p, const ffgp_links* l) {
  if (h->train_persist_off || h->use_naive || h->timing) return false;
  if (p->n <= 0 || p->n > TR_N || p->D <= 0 || p->D > TR_D || p->d <= 0 || p->d > TR_Y) return false;
  if (p->cov_dev || p->pair || p->tree || p->add_mat_dev || p->add_all != 0.0 || p->mean_jitter != 0.0) return false;
  if (!p->X_dev || !p->Y_dev || !p->w_dev || !p->amp_dev || !p->diag_add_dev) return false;
  if (p->kfun < FFGP_KFUN_SE || p->kfun > FFGP_KFUN_RQ || p->ll_variant != FFGP_LL_V1) return false;
  (void)l;
  return true;
}

// steps of F models (each must pass ffgp_train_persist_ok), one launch; synchronous.  Returns 0 or the pivot status of the first
// model (in the caller's order) whose Sigma was not positive definite at some step.
int ffgp_train_persist(ffgp_handle* h, int F, const ffgp_problem* p, const ffgp_links* l, int steps, const ffgp_adam* opt, double* state_dev,
                       long state_stride, long step0, double* trace_dev, long trace_stride) {
  static bool attr_set[64] = {false};
  if (h->device >= 0 && h->device < 64 && !attr_set[h->device]) {
    FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_train_persist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 TR_LDS_DOUBLES * (int)sizeof(double)));
    attr_set[h->device] = true;
  }
  // one device block: [F models | 2 steps bias corrections | 2 F status ints]
  const size_t tab_bytes = (size_t)F * sizeof(TrainModel);
  const size_t bc_off = (tab_bytes + 255) / 256 * 256;
  const size_t info_off = bc_off + ((size_t)2 * steps * sizeof(double) + 255) / 256 * 256;
  const size_t need = info_off + (size_t)2 * F * sizeof(int);
  if (need > h->train_tab_bytes) {
    if (h->train_tab) {
      FFGP_HIP(hipStreamSynchronize(h->stream));
      hipFree(h->train_tab);
      h->train_tab = nullptr;
      h->train_tab_bytes = 0;
    }
    if (hipMalloc(&h->train_tab, need + need / 2) != hipSuccess) {
      (void)hipGetLastError();
      return FFGP_ERR_ALLOC;
    }
    h->train_tab_bytes = need + need / 2;
  }
  std::vector<char> host(info_off + (size_t)2 * F * sizeof(int), 0);
  TrainModel* tm = reinterpret_cast<TrainModel*>(host.data());
  for (int f = 0; f < F; ++f) {
    const ffgp_problem& q = p[f];
    TrainModel& m = tm[f];
    m.n = q.n; m.D = q.D; m.d = q.d; m.nw = l[f].w_broadcast ? 1 : q.D;
    m.X = q.X_dev; m.Y = q.Y_dev;
    m.w = const_cast<double*>(q.w_dev); m.amp = const_cast<double*>(q.amp_dev); m.dadd = const_cast<double*>(q.diag_add_dev);
    m.diag_vec = q.diag_vec_dev; m.diag_stride = q.diag_stride;
    m.l = l[f];
    m.clamp = q.clamp_min; m.rinv = (q.kparam != 0.0) ? 1.0 / q.kparam : 1.0; m.pi_const = q.pi_const; m.kfun = q.kfun;
    m.state = state_dev + (size_t)f * state_stride;
    m.trace = trace_dev + (size_t)f * trace_stride;
  }
  double* bc = reinterpret_cast<double*>(host.data() + bc_off);
  for (int k = 0; k < steps; ++k) {
    const double t = (double)(step0 + k + 1);
    bc[2 * k] = 1.0 - std::pow(opt->beta1, t);
    bc[2 * k + 1] = std::sqrt(1.0 - std::pow(opt->beta2, t));
  }
  char* dev = reinterpret_cast<char*>(h->train_tab);
  FFGP_HIP(hipMemcpyAsync(dev, host.data(), host.size(), hipMemcpyHostToDevice, h->stream));
  FFGP_HIP(hipStreamSynchronize(h->stream));       // (the staging vector goes out of scope; a pageable copy may still be reading it)
  TrainCommon cm;
  cm.steps = steps; cm.lr = opt->lr; cm.b1 = opt->beta1; cm.b2 = opt->beta2; cm.eps = opt->eps;
  cm.bc = reinterpret_cast<const double*>(dev + bc_off);
  cm.info = reinterpret_cast<int*>(dev + info_off);
  cm.fail_step = cm.info + F;
  hipLaunchKernelGGL(ffgp_train_persist_kernel, dim3(F), dim3(TR_T), TR_LDS_DOUBLES * sizeof(double), h->stream,
                     reinterpret_cast<const TrainModel*>(dev), cm);
  if (hipGetLastError() != hipSuccess) return FFGP_ERR_HIP;
  std::vector<int> st(2 * F);
  FFGP_HIP(hipMemcpyAsync(st.data(), cm.info, (size_t)2 * F * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  FFGP_HIP(hipStreamSynchronize(h->stream));
  ffgp_invalidate(h);
  for (int f = 0; f < F; ++f)
    if (st[f] != 0) return st[f];
  return FFGP_OK;
}
