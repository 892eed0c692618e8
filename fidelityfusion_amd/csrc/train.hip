// One workgroup, one small GP: K Adam steps of F models in ONE launch, and the per-step likelihood + gradient call (round 6).
// The reference's hot loop runs at the sizes of its experiments (FidelityFusion_Models/ResGP.py:78-112, GaussianProcess/cigp_v10.py:92-104:
// per fidelity 100-1000 iterations of zero_grad / loss = -negative_log_likelihood / backward / Adam step at N = 16 ... 128;
// Experiments/GAR_Aligned/exp_aligned.py:66-74: 100 low- against 4..32 high-fidelity points).  ffgp_train_raw used to enqueue 13 dependent
// launches per step at n = 128 (0.096 ms per step, two thirds of it launch floors).  Here one PERSISTENT workgroup per model
// (gridDim.x = models, 512 threads) runs every step inside the kernel:
//
//   links (raw -> effective parameters)  ->  Sigma assembled straight into LDS as 16 x 16 blocks [16][17] (lower block triangle, 78 KiB),
//       the kernel values parked in a global scratch for the gradient pass
//   ->  blocked Cholesky AND inverse, two workgroup barriers per 16-column stage: wave 0 factors + inverts the diagonal block in
//       registers on the DP-ALU DPP pivot step of f16_steps.h while the helper waves apply the previous column and form the previous row
//       of the inverse; then all eight waves solve the column.  Only ceil(n / 16) stages run: n = 32 costs a quarter of n = 128.
//       (This loop became the factorisation's diagonal-block kernel, ffgp_potrf_diag128_v4 in potrf.hip.)
//   ->  Gamma = L^-1 Y, A = L^-T Gamma as MFMA block chains on zero-padded [128][16] images, the value
//   ->  Sigma^-1 = L^-T L^-1 block by block on the matrix cores, each 16 x 16 block consumed in its accumulators: G = d/2 Sigma^-1 - 1/2 A A^T,
//       the gradient sums of grad.hip (amplitude, length scales, trace) -- Sigma^-1 is never stored
//   ->  one workgroup reduction, the links' chain rule and torch.optim.Adam's update (operation for operation as ffgp_adam_kernel) on
//       parameters and moments that live in LDS for the whole call; the step's loss goes to the trace.
// A Sigma that is not positive definite stops THAT model at that step (its status word, NaN in its trace from there on, parameters as
// they were when the step began); the other models of the launch train on.
// EVALUATE mode (tr_body<DM, false>, ffgp_small_mfma_kernel): the same pass once, without Adam -- value, raw-parameter gradients, dL/dY and
// diag G written out -- for ffgp_nlml_fused_small_batch and ffgp_nlml_fused_raw at n <= 128.
// Covers: n <= 128, D <= 16, d <= 16, one radial-profile kernel, V1 likelihood, diag_add and diag_vec (no matrix / all-entries /
// mean(K) extras, no learnable profile parameter): what cigp_v10 produces.  Everything else keeps the paths it had.
#include "ffgp_internal.h"
#include "f16_steps.h"

#define TR_T 512
#define TR_BLD 17
#define TR_BLK (16 * TR_BLD)
#define TR_NBLK 36
#define TR_N 128
#define TR_D 16
#define TR_Y 16
#define TR_NRED 20        // values of the step's one workgroup reduction: ss, s_amp, tr G, log-det, tot[16]

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
  // evaluate mode (one likelihood + gradient call, no Adam: ffgp_nlml_fused_small_batch / ffgp_nlml_fused_raw at n <= 128)
  double* nll;                           // [1] value
  double* g_w; double* g_amp; double* g_dadd; double* g_Y; double* g_dvec;   // gradients (raw parameters, Y, diag_vec); any may be null
  int want_grad;
  double* kbuf;                          // [36][4][64] this model's kernel values K / amp, written by the assembly and read back by the
                                         // gradient pass of the same step (same lane, same slot: the exp is evaluated once per entry and step)
};
struct TrainCommon {
  int steps;
  double lr, b1, b2, eps;
  const double* bc;                      // [steps][2]: 1 - beta1^t, sqrt(1 - beta2^t) -- computed on the host with the C library's pow, as Python does
  int* info;                             // [models] status: 0, or the 1-based index of the first non-positive pivot of the step that failed
  int* fail_step;                        // [models] the step at which it happened
  int* shared_info;                      // evaluate mode: the handle's status word (batch: atomicMax of the failing pivot; single: plain store)
  int info_max;
  long* prof;                            // development (FFGP_TRAIN_TRACE=1): [12] wall_clock64 ticks per phase, summed over model 0's steps
};
// workgroup barrier that publishes LDS only: __syncthreads() also drains the wave's GLOBAL stores (the kernel values parked for the
// gradient pass, the trace), whose round trip to L2 would be exposed at every barrier behind them.  Nothing inside the step loop is
// handed from thread to thread through global memory.
#define TR_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define TR_PROF(k)                                                   \
  do {                                                               \
    if (cm.prof && tid == 0 && blockIdx.x == 0) {                    \
      const long now_ = wall_clock64();                              \
      cm.prof[k] += now_ - t_prof;                                   \
      t_prof = now_;                                                 \
    }                                                                \
  } while (0)

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

// sum over the 64 lanes of the wave, in all of them: four DPP steps inside the rows of 16 lanes, then the rows exchanged by gfx950's
// row swaps (the shuffle form, ds_bpermute, is an LDS crossbar round trip per step: 6 dependent trips per value)
template <int CTRL>
__device__ __forceinline__ double tr_dpp_add(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return x + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double tr_wsum(double x) {
  x = tr_dpp_add<0xB1>(x);
  x = tr_dpp_add<0x4E>(x);
  x = tr_dpp_add<0x141>(x);
  x = tr_dpp_add<0x140>(x);
  {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    x = __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
  }
  const int lo = __double2loint(x), hi = __double2hiint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

__device__ __forceinline__ int tr_blk(int bi, int bj) { return (bi * (bi + 1) / 2 + bj) * TR_BLK; }
// t-th block of the row-major enumeration of the lower block triangle -> (bi, bj)
__device__ __forceinline__ void tr_unblk(int t, int& bi, int& bj) {
  bi = 0;
#pragma unroll
  for (int q = 1; q < 8; ++q) bi += (t >= q * (q + 1) / 2) ? 1 : 0;
  bj = t - bi * (bi + 1) / 2;
}
// block dealt to `wave` in round q of the assembly / the Sigma^-1 pass: the row-major enumeration has the expensive blocks of the Sigma^-1
// pass first (block (bi, bj) costs nst - bi products), so the rounds run forwards and backwards in turn -- 16 products for the busiest
// wave at n = 128 instead of 19.  Both passes MUST deal alike: the kernel values travel from one to the other by (block, lane) slot.
__device__ __forceinline__ int tr_deal(int q, int wave) { return 8 * q + ((q & 1) ? 7 - wave : wave); }
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
// acc += sum_{kb = k0}^{k1 - 1} op(P_kb) * Q_kb over 16 x 16 blocks, the NEXT block's operands requested before this block's four MFMAs
// (a runtime loop of load-then-multiply rounds waits one LDS round trip per block).  P_kb at baseA + offA(kb): element (m, k) at
// [m * lda + k], or (TA) the transposed block: (m, k) at [k * lda + m]; Q_kb at baseB + offB(kb): element (k, n) at [k * ldb + n].
template <bool TA, class OA, class OB>
__device__ __forceinline__ void tr_chain(tr_d4& acc, int k0, int k1, const double* baseA, OA offA, int lda, const double* baseB, OB offB,
                                         int ldb, int lane) {
  if (k0 >= k1) return;
  const int m = lane & 15, g = lane >> 4;
  double a[4], b[4];
  {
    const double* pa = baseA + offA(k0);
    const double* pb = baseB + offB(k0);
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      const int k = kq * 4 + g;
      a[kq] = TA ? pa[k * lda + m] : pa[m * lda + k];
      b[kq] = pb[k * ldb + m];
    }
  }
  for (int kb = k0; kb < k1; ++kb) {
    double an[4], bn[4];
    const int kn = min(kb + 1, k1 - 1);      // (the last round re-reads its own block)
    const double* pa = baseA + offA(kn);
    const double* pb = baseB + offB(kn);
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      const int k = kq * 4 + g;
      an[kq] = TA ? pa[k * lda + m] : pa[m * lda + k];
      bn[kq] = pb[k * ldb + m];
    }
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kq], b[kq], acc, 0, 0, 0);
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      a[kq] = an[kq];
      b[kq] = bn[kq];
    }
  }
}
// accumulator (lane (g, c), register r = entry (g + 4 r, c)) -> the block's [16][17] home
__device__ __forceinline__ void tr_store(double* dst, const tr_d4& acc, int g, int c, double sign) {
#pragma unroll
  for (int r = 0; r < 4; ++r) dst[(g + 4 * r) * TR_BLD + c] = sign * acc[r];
}

// LDS (doubles): S 36 * 272 | Xs [128][17] | Ym, Gam, Am [128][16] each | piv [128] | dvec [128] | small
#define TR_OFF_XS (TR_NBLK * TR_BLK)
#define TR_OFF_YM (TR_OFF_XS + TR_N * (TR_D + 1))
#define TR_OFF_GAM (TR_OFF_YM + TR_N * TR_Y)
#define TR_OFF_AM (TR_OFF_GAM + TR_N * TR_Y)
#define TR_OFF_PIV (TR_OFF_AM + TR_N * TR_Y)
#define TR_OFF_DVEC (TR_OFF_PIV + TR_N)
#define TR_OFF_SMALL (TR_OFF_DVEC + TR_N)
#define TR_SMALL_DOUBLES (16 + 3 * 20 + 8 + 8 * TR_NRED + TR_NRED + 16)
#define TR_LDS_DOUBLES (TR_OFF_SMALL + TR_SMALL_DOUBLES)

// DM: the input dimensions the per-entry loops are unrolled for (8 or 16: every model of the launch has D <= DM)
// TRAIN: every step of the loop with Adam inside; !TRAIN ("evaluate"): ONE pass that writes the value and the gradients out
template <int DM, bool TRAIN>
__device__ __forceinline__ void tr_body(const TrainModel& M, const TrainCommon& cm) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
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
  // targets, Gamma and A live as [128][16] images, zero beyond (n, d): the matrix-core products read them without guards
  for (int idx = tid; idx < TR_N * TR_Y; idx += TR_T) {
    const int i = idx >> 4, q = idx & 15;
    Ym[idx] = (i < n && q < d) ? M.Y[i * d + q] : 0.0;
    Gam[idx] = 0.0;
    Am[idx] = 0.0;
  }
  for (int idx = tid; idx < TR_N * (TR_D + 1); idx += TR_T) Xs[idx] = 0.0;      // (columns >= D and rows >= n stay zero: the unrolled loops read them)
  for (int i = tid; i < TR_N; i += TR_T) dvec[i] = (M.diag_vec && i < n) ? M.diag_vec[(size_t)i * M.diag_stride] : 0.0;
  if (tid < npar) {
    raw[tid] = (tid < nw) ? M.w[tid] : (tid == nw ? M.amp[0] : (M.dadd ? M.dadd[0] : 0.0));
    if (TRAIN) {
      mom[tid] = M.state[tid];
      mo2[tid] = M.state[npar + tid];
    }
  }
  for (int t = wave; t < TR_NBLK; t += 8) {
    int bi, bj;
    tr_unblk(t, bi, bj);
    if (bi < nst) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) S[tr_blk(bi, bj) + (g + 4 * r) * TR_BLD + c] = (bi == bj && g + 4 * r == c) ? 1.0 : 0.0;
  }
  if (tid == 0) flags[0] = 0;
  if (lane == 0) flags[8 + wave] = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3;      // HW_REG_HW_ID, SIMD_ID (bits 5:4)
  __syncthreads();
  // helpers of the factorisation's stage [A]: the waves that do NOT share wave 0's SIMD (six, with two waves per SIMD)
  int hidx = -1, nh = 0;
  {
    const int s0 = flags[8];
    for (int w_ = 1; w_ < 8; ++w_) {
      const bool is_h = flags[8 + w_] != s0;
      if (is_h && w_ == wave) hidx = nh;
      nh += is_h ? 1 : 0;
    }
    if (nh == 0) {      // (every wave on one SIMD cannot happen with 8 waves on 4 SIMDs; keep the kernel correct anyway)
      nh = 7;
      hidx = wave - 1;
    }
    hidx = __builtin_amdgcn_readfirstlane(hidx);
    nh = __builtin_amdgcn_readfirstlane(nh);
  }
  if (tid < D) wv[tid] = tr_link_val(M.l.w_link, raw[M.l.w_broadcast ? 0 : tid], M.l.w_c);
  if (tid == 64) sc[0] = tr_link_val(M.l.amp_link, raw[nw], M.l.amp_c);
  if (tid == 65) sc[1] = M.dadd ? tr_link_val(M.l.dadd_link, raw[nw + 1], M.l.dadd_c) : 0.0;
  __syncthreads();

  int failed = 0;
  long t_prof = cm.prof ? wall_clock64() : 0;
  const int nsteps = TRAIN ? cm.steps : 1;
  const int lane_k = lane, wave_k = wave;
  for (int step = 0; step < nsteps; ++step) {
    // (the lane coordinates opaque per iteration: otherwise every per-lane offset of every phase is hoisted out of the step loop and kept
    //  alive across it -- 256 registers and 79 spilled ones, reloaded inside the phases)
    int lane = lane_k, wave = wave_k;
    asm volatile("" : "+v"(lane), "+s"(wave));
    const int g = lane >> 4, c = lane & 15;
    // ---- P0: scaled inputs (the effective parameters were refreshed by the threads that updated the raw ones)
    for (int idx = tid; idx < n * D; idx += TR_T) {      // (shifted by the first point: only differences enter the kernel)
      const int i = idx / D, k = idx - i * D;
      Xs[i * (TR_D + 1) + k] = (M.X[idx] - M.X[k]) * wv[k];
    }
    TR_BARRIER();
    const double amp = sc[0], dadd = sc[1];
    TR_PROF(0);

    // ---- P1: Sigma, lower block triangle (diagonal blocks symmetric-full: the in-register factor wants both halves); rows / columns
    //      beyond n are identity
    for (int q_ = 0; q_ < 5; ++q_) {
      const int t = tr_deal(q_, wave);
      if (t >= nblk) continue;
      int bi, bj;
      tr_unblk(t, bi, bj);
      double* dst = S + tr_blk(bi, bj);
      const int j = bj * 16 + c;
      double xj[DM], sq[4];
#pragma unroll
      for (int k = 0; k < DM; ++k) xj[k] = Xs[j * (TR_D + 1) + k];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double* xi = Xs + (bi * 16 + g + 4 * r) * (TR_D + 1);
        sq[r] = 0.0;
#pragma unroll
        for (int k = 0; k < DM; ++k) {
          const double df = xi[k] - xj[k];
          sq[r] = __builtin_fma(df, df, sq[r]);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = bi * 16 + g + 4 * r;
        const double s_ = fmax(sq[r], M.clamp);
        const double ev = (M.kfun == FFGP_KFUN_SE) ? ffgp_exp_fast(-0.5 * s_, ec) : ffgp_kfun_val(M.kfun, M.rinv, s_);
        M.kbuf[(t * 4 + r) * 64 + lane] = ev;
        double kv = amp * ev;
        if (i == j) kv += dadd + dvec[i];
        if (i >= n || j >= n) kv = (i == j) ? 1.0 : 0.0;
        dst[(g + 4 * r) * TR_BLD + c] = kv;
      }
    }
    TR_BARRIER();
    TR_PROF(1);

    // ---- P2: blocked Cholesky over 16-column stages AND the inverse, two barriers per stage.
    //   [A] wave 0 applies column jj - 1 to its diagonal block (jj, jj) and factors + inverts it in registers (F: the slot receives
    //       inv(L_jj), the pivots go to piv[]); in its shadow the helper waves -- every wave that does not share wave 0's SIMD: fp64
    //       MFMAs and the pivot loop's fp64 vector instructions use the same pipe -- apply column jj - 1 to all the other blocks and
    //       compute row block jj - 1 of the inverse, X[s][j] = -inv(L_s) sum_{k=j}^{s-1} L[s][k] X[k][j], into registers;
    //   [B] the inverse's row is stored over row jj - 1 of L (nobody reads it any more) and column jj is solved by all waves.
    for (int jj = 0; jj < nst; ++jj) {
      tr_d4 Xn[2];
      if (wave == 0) {
        double* Dj = S + tr_blk(jj, jj);
        tr_d4 upd = {0.0, 0.0, 0.0, 0.0};      // the one update still missing from this block: applied on the way into the registers
        if (jj > 0) tr_mma_nt(upd, S + tr_blk(jj, jj - 1), S + tr_blk(jj, jj - 1), lane);
        int cc = c, gg = g;
        asm volatile("" : "+v"(cc), "+v"(gg));      // (opaque per iteration: the stage loop must not be specialised per jj)
        double v[4], w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = Dj[(gg + 4 * r) * TR_BLD + cc] - upd[r];
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
      } else if (hidx >= 0 && jj > 0) {
        // column jj - 1 reaches every block (i, k), jj <= k <= i, but (jj, jj)
        const int m = nst - jj;
        for (int t = 1 + hidx; t < m * (m + 1) / 2; t += nh) {
          int a, b;
          tr_unblk(t, a, b);
          const int i = jj + a, k = jj + b;
          tr_d4 acc = {0.0, 0.0, 0.0, 0.0};
          tr_mma_nt(acc, S + tr_blk(i, jj - 1), S + tr_blk(k, jj - 1), lane);
          double* dst = S + tr_blk(i, k);
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[(g + 4 * r) * TR_BLD + c] -= acc[r];
        }
        // row block s = jj - 1 of the inverse, columns hidx and hidx + nh, kept in registers until row s of L is dead
        const int s_ = jj - 1;
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          const int j = hidx + q2 * nh;
          if (j >= s_) continue;
          tr_d4 T = {0.0, 0.0, 0.0, 0.0};
          tr_chain<false>(T, j, s_, S + tr_blk(s_, 0), [](int k) { return k * TR_BLK; }, TR_BLD, S, [j](int k) { return tr_blk(k, j); }, TR_BLD, lane);
          tr_d4 acc = {0.0, 0.0, 0.0, 0.0};
          const double* Ws = S + tr_blk(s_, s_);
#pragma unroll
          for (int kq = 0; kq < 4; ++kq) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ws[c * TR_BLD + kq * 4 + g], T[kq], acc, 0, 0, 0);
          Xn[q2] = acc;
        }
      }
      TR_BARRIER();
      TR_PROF(2);
      if (hidx >= 0 && jj > 0) {
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          const int j = hidx + q2 * nh;
          if (j < jj - 1) tr_store(S + tr_blk(jj - 1, j), Xn[q2], g, c, -1.0);
        }
      }
      // solve: L[i][jj] = S[i][jj] inv(L_jj)^T for the block rows below
      for (int i = jj + 1 + wave; i < nst; i += 8) {
        tr_d4 acc = {0.0, 0.0, 0.0, 0.0};
        tr_mma_nt(acc, S + tr_blk(i, jj), S + tr_blk(jj, jj), lane);
        tr_store(S + tr_blk(i, jj), acc, g, c, 1.0);
      }
      TR_BARRIER();
      TR_PROF(3);
    }
    if (flags[0] != 0) {       // (uniform: every thread reads the same word behind the barrier)
      failed = flags[0];
      if (TRAIN) {
        if (tid == 0) {
          cm.info[blockIdx.x] = failed;
          cm.fail_step[blockIdx.x] = step;
        }
        for (int k = step + tid; k < cm.steps; k += TR_T) M.trace[k] = __builtin_nan("");
      } else if (tid == 0) {      // (the blocked path goes on with a unit pivot and returns garbage; here the value is NaN)
        if (cm.info_max) atomicMax(cm.shared_info, failed);
        else cm.shared_info[0] = failed;
        M.nll[0] = __builtin_nan("");
      }
      break;
    }
    // ---- the last row block of the inverse (s = nst - 1), one column per wave
    if (nst > 1) {
      const int s_ = nst - 1, j = wave;
      tr_d4 X = {0.0, 0.0, 0.0, 0.0};
      if (j < s_) {
        tr_d4 T = {0.0, 0.0, 0.0, 0.0};
        tr_chain<false>(T, j, s_, S + tr_blk(s_, 0), [](int k) { return k * TR_BLK; }, TR_BLD, S, [j](int k) { return tr_blk(k, j); }, TR_BLD, lane);
        const double* Ws = S + tr_blk(s_, s_);
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) X = __builtin_amdgcn_mfma_f64_16x16x4f64(Ws[c * TR_BLD + kq * 4 + g], T[kq], X, 0, 0, 0);
      }
      TR_BARRIER();
      if (j < s_) tr_store(S + tr_blk(s_, j), X, g, c, -1.0);
      TR_BARRIER();
    }
    TR_PROF(5);

    // ---- P3: Gamma = W Y, A = W^T Gamma on the matrix cores (W = L^-1, lower; the d <= 16 target columns are one block column): wave w
    //      owns block row w of Gamma (w + 1 products) and block row 7 - w of A (w + 1 products)
    {
      if (wave < nst) {
        const int bi = wave;
        tr_d4 acc = {0.0, 0.0, 0.0, 0.0};
        tr_chain<false>(acc, 0, bi + 1, S + tr_blk(bi, 0), [](int kb) { return kb * TR_BLK; }, TR_BLD, Ym, [](int kb) { return kb * 16 * TR_Y; },
                        TR_Y, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) Gam[(bi * 16 + g + 4 * r) * TR_Y + c] = acc[r];
      }
      TR_BARRIER();
      if (7 - wave < nst) {
        const int bi = 7 - wave;
        tr_d4 acc = {0.0, 0.0, 0.0, 0.0};
        tr_chain<true>(acc, bi, nst, S, [bi](int kb) { return tr_blk(kb, bi); }, TR_BLD, Gam, [](int kb) { return kb * 16 * TR_Y; }, TR_Y, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) Am[(bi * 16 + g + 4 * r) * TR_Y + c] = acc[r];
      }
      TR_BARRIER();
    }
    TR_PROF(6);

    // ---- P5: per lane partial sums -- ss (value), s_amp, tr G, tot[k] (length scales); Sigma^-1 block by block on the matrix cores.
    // (The length-scale sums through the matrix cores as well -- tot_k = sum_j [V_jk + x_jk^2 c_j - 2 x_jk U_jk] with U | V = W_low^T [X | X^2]
    //  as one MFMA group per block, the block's weights used as the A operand straight from the accumulator layout, added into an LDS
    //  image -- was built, passed every test and measured SLOWER: 12.4 against 10.2 us at n = 128, D = 5; four LDS atomics per lane and block
    //  and two more barriers cost more than the 24 vector instructions per entry they replace.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this lane's parked kernel values have landed (long ago)
    double ss = 0.0, s_amp = 0.0, trg = 0.0, tk[DM];
#pragma unroll
    for (int k = 0; k < DM; ++k) tk[k] = 0.0;
    const double lpiv = (tid < n) ? log(piv[tid]) : 0.0;
#pragma unroll
    for (int q = 0; q < TR_N * TR_Y / TR_T; ++q) ss = __builtin_fma(Gam[tid + TR_T * q], Gam[tid + TR_T * q], ss);
    for (int q_ = 0; q_ < 5 && (TRAIN || M.want_grad); ++q_) {
      const int t = tr_deal(q_, wave);
      if (t >= nblk) continue;
      int bi, bj;
      tr_unblk(t, bi, bj);
      double evs[4];      // (requested before the block's products: the loads fly under the MFMA chain; written by this very lane)
#pragma unroll
      for (int r = 0; r < 4; ++r) evs[r] = M.kbuf[(t * 4 + r) * 64 + lane];
      tr_d4 acc = {0.0, 0.0, 0.0, 0.0};
      tr_chain<true>(acc, bi, nst, S, [bi](int kb) { return tr_blk(kb, bi); }, TR_BLD, S, [bj](int kb) { return tr_blk(kb, bj); }, TR_BLD, lane);
      const int j = bj * 16 + c;
      double xj[DM];
#pragma unroll
      for (int k = 0; k < DM; ++k) xj[k] = Xs[j * (TR_D + 1) + k];
#pragma unroll 1
      for (int r = 0; r < 4; ++r) {      // (one row at a time: four rows' differences in flight at once cost more registers than the chip has)
        const int i = bi * 16 + g + 4 * r;
        const double* xi = Xs + i * (TR_D + 1);
        const double accr = (r == 0) ? acc[0] : (r == 1) ? acc[1] : (r == 2) ? acc[2] : acc[3];
        const double ev = (r == 0) ? evs[0] : (r == 1) ? evs[1] : (r == 2) ? evs[2] : evs[3];
        double dsq[DM], sq = 0.0;
#pragma unroll
        for (int k = 0; k < DM; ++k) {
          const double df = xi[k] - xj[k];
          dsq[k] = df * df;
          sq += dsq[k];
        }
        double aa = 0.0;
        for (int q = 0; q < d; ++q) aa = __builtin_fma(Am[i * TR_Y + q], Am[j * TR_Y + q], aa);
        const bool live = (i < n && j <= i);
        const double gv = live ? 0.5 * (double)d * accr - 0.5 * aa : 0.0;
        const double sym = (i == j) ? 1.0 : 2.0;
        double m2 = ev;
        if (M.kfun != FFGP_KFUN_SE) m2 = ffgp_kfun_m2d(M.kfun, M.rinv, fmax(sq, M.clamp));
        s_amp = __builtin_fma(sym * gv, ev, s_amp);
        if (i == j) {
          trg += gv;
          if (!TRAIN && live) dvec[i] = gv;      // (evaluate mode: diag G for g_diag_vec; the diagonal extra was consumed by the assembly)
        }
        const double wl = (sq >= M.clamp) ? sym * gv * amp * m2 : 0.0;
#pragma unroll
        for (int k = 0; k < DM; ++k) tk[k] = __builtin_fma(wl, dsq[k], tk[k]);
      }
    }
    TR_PROF(7);
    // ---- one workgroup reduction for all of them (and the log-determinant: one pivot per thread)
    {
      constexpr int NV = 4 + DM;
      double vals[NV];
      vals[0] = ss; vals[1] = s_amp; vals[2] = trg;
      vals[3] = lpiv;
#pragma unroll
      for (int k = 0; k < DM; ++k) vals[4 + k] = tk[k];
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        vals[q] = tr_wsum(vals[q]);
      }
      if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) red[wave * TR_NRED + q] = vals[q];
      }
      TR_BARRIER();
      if (tid < NV) {
        double x = 0.0;
#pragma unroll
        for (int wv_ = 0; wv_ < 8; ++wv_) x += red[wv_ * TR_NRED + tid];
        tot[tid] = x;
      }
      TR_BARRIER();
    }
    TR_PROF(8);
    // ---- P6: the loss of this step (before the update), the raw gradients through the links, Adam
    const double value = oscale * (0.5 * tot[0] + (double)d * 0.5 * tot[3] + 0.5 * (double)n * (double)d * log(2.0 * M.pi_const));
    if (!TRAIN) {
      if (tid == 0) M.nll[0] = value;
      if (M.want_grad) {
        if (tid < npar) {
          double gr;
          if (tid < nw) {
            if (!M.l.w_broadcast) {
              gr = oscale * (-tot[4 + tid] / wv[tid]) * tr_link_der(M.l.w_link, raw[tid], M.l.w_c);
            } else {
              double sg = 0.0;
              for (int k = 0; k < D; ++k) sg += -tot[4 + k] / wv[k];
              gr = oscale * sg * tr_link_der(M.l.w_link, raw[0], M.l.w_c);
            }
            if (M.g_w) M.g_w[tid] = gr;
          } else if (tid == nw) {
            if (M.g_amp) M.g_amp[0] = oscale * tot[1] * tr_link_der(M.l.amp_link, raw[nw], M.l.amp_c);
          } else if (M.g_dadd) {
            M.g_dadd[0] = oscale * tot[2] * (M.dadd ? tr_link_der(M.l.dadd_link, raw[nw + 1], M.l.dadd_c) : 1.0);
          }
        }
        if (M.g_Y)
          for (int idx = tid; idx < n * d; idx += TR_T) M.g_Y[idx] = oscale * Am[(idx / d) * TR_Y + (idx % d)];
        if (M.g_dvec)
          for (int i = tid; i < n; i += TR_T) M.g_dvec[i] = oscale * dvec[i];
      }
      break;
    }
    if (tid == 0) M.trace[step] = value;
    if (tid < npar) {
      double gr;
      if (tid < nw) {
        if (!M.l.w_broadcast) {
          gr = oscale * (-tot[4 + tid] / wv[tid]) * tr_link_der(M.l.w_link, raw[tid], M.l.w_c);
        } else {
          double sg = 0.0;
          for (int k = 0; k < D; ++k) sg += -tot[4 + k] / wv[k];
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
      const double pnew = raw[tid] + (-(cm.lr / bc1)) * (m1 / denom);         // param.addcdiv_(exp_avg, denom, value = -step_size)
      raw[tid] = pnew;
      // the next step's effective value (nobody reads wv / sc between the reduction's barrier and the one below)
      if (tid < nw) {
        if (!M.l.w_broadcast) {
          wv[tid] = tr_link_val(M.l.w_link, pnew, M.l.w_c);
        } else {
          const double e_ = tr_link_val(M.l.w_link, pnew, M.l.w_c);
          for (int k = 0; k < D; ++k) wv[k] = e_;
        }
      } else if (tid == nw) {
        sc[0] = tr_link_val(M.l.amp_link, pnew, M.l.amp_c);
      } else {
        sc[1] = tr_link_val(M.l.dadd_link, pnew, M.l.dadd_c);
      }
    }
    TR_BARRIER();
    TR_PROF(9);
  }
  // ---- parameters and moments back to the caller's tensors (a failed step left them as they were when it began)
  if (TRAIN && tid < npar) {
    if (tid < nw) M.w[tid] = raw[tid];
    else if (tid == nw) M.amp[0] = raw[tid];
    else M.dadd[0] = raw[tid];
    M.state[tid] = mom[tid];
    M.state[npar + tid] = mo2[tid];
  }
  (void)failed;
}

template <int DM>
__global__ __launch_bounds__(TR_T) void ffgp_train_persist_kernel(const TrainModel* __restrict__ tab, TrainCommon cm) {
  const TrainModel M = tab[blockIdx.x];
  tr_body<DM, true>(M, cm);
}
// evaluate mode: up to eight models per launch, their descriptions by value (the enqueue-only entry points must not stage anything in
// host memory that a later call could overwrite)
#define TR_EVAL_BATCH 8
struct TrainEvalBatch {
  TrainModel m[TR_EVAL_BATCH];
};
template <int DM>
__global__ __launch_bounds__(TR_T) void ffgp_small_mfma_kernel(TrainEvalBatch b, TrainCommon cm) {
  tr_body<DM, false>(b.m[blockIdx.x], cm);
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
    FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_train_persist_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 TR_LDS_DOUBLES * (int)sizeof(double)));
    FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_train_persist_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 TR_LDS_DOUBLES * (int)sizeof(double)));
    attr_set[h->device] = true;
  }
  // one device block: [F models | 2 steps bias corrections | 2 F status ints | F x 36 x 256 kernel values]; its first three parts are
  // staged in pinned host memory owned by the handle (an asynchronous copy from a temporary would have to be waited for)
  const size_t tab_bytes = (size_t)F * sizeof(TrainModel);
  const size_t bc_off = (tab_bytes + 255) / 256 * 256;
  const size_t info_off = bc_off + ((size_t)2 * steps * sizeof(double) + 255) / 256 * 256;
  const size_t head = info_off + (size_t)2 * F * sizeof(int);
  const size_t k_off = (head + 255) / 256 * 256;
  const size_t need = k_off + (size_t)F * TR_NBLK * 256 * sizeof(double);
  if (need > h->train_tab_bytes) {
    FFGP_HIP(hipStreamSynchronize(h->stream));
    if (h->train_tab) hipFree(h->train_tab);
    if (h->train_host) hipHostFree(h->train_host);
    h->train_tab = nullptr;
    h->train_host = nullptr;
    h->train_tab_bytes = 0;
    const size_t cap = need + need / 2;
    if (hipMalloc(&h->train_tab, cap) != hipSuccess || hipHostMalloc(&h->train_host, cap, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      if (h->train_tab) hipFree(h->train_tab);
      h->train_tab = nullptr;
      return FFGP_ERR_ALLOC;
    }
    h->train_tab_bytes = cap;
  }
  char* host = reinterpret_cast<char*>(h->train_host);      // (the previous call on this handle was synchronous: the buffer is free)
  memset(host, 0, head);
  char* dev = reinterpret_cast<char*>(h->train_tab);
  TrainModel* tm = reinterpret_cast<TrainModel*>(host);
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
    m.kbuf = reinterpret_cast<double*>(dev + k_off) + (size_t)f * TR_NBLK * 256;
    m.want_grad = 1;
  }
  double* bc = reinterpret_cast<double*>(host + bc_off);
  for (int k = 0; k < steps; ++k) {
    const double t = (double)(step0 + k + 1);
    bc[2 * k] = 1.0 - std::pow(opt->beta1, t);
    bc[2 * k + 1] = std::sqrt(1.0 - std::pow(opt->beta2, t));
  }
  FFGP_HIP(hipMemcpyAsync(dev, host, head, hipMemcpyHostToDevice, h->stream));
  TrainCommon cm;
  cm.steps = steps; cm.lr = opt->lr; cm.b1 = opt->beta1; cm.b2 = opt->beta2; cm.eps = opt->eps;
  cm.bc = reinterpret_cast<const double*>(dev + bc_off);
  cm.info = reinterpret_cast<int*>(dev + info_off);
  cm.fail_step = cm.info + F;
  cm.shared_info = nullptr;
  cm.info_max = 0;
  cm.prof = nullptr;
  static const bool trace_on = getenv("FFGP_TRAIN_TRACE") && atoi(getenv("FFGP_TRAIN_TRACE")) != 0;
  if (trace_on) {
    FFGP_HIP(hipMalloc(&cm.prof, 12 * sizeof(long)));
    FFGP_HIP(hipMemset(cm.prof, 0, 12 * sizeof(long)));
  }
  int Dmax = 0;
  for (int f = 0; f < F; ++f) Dmax = std::max(Dmax, p[f].D);
  if (Dmax <= 8)
    hipLaunchKernelGGL(ffgp_train_persist_kernel<8>, dim3(F), dim3(TR_T), TR_LDS_DOUBLES * sizeof(double), h->stream,
                       reinterpret_cast<const TrainModel*>(dev), cm);
  else
    hipLaunchKernelGGL(ffgp_train_persist_kernel<16>, dim3(F), dim3(TR_T), TR_LDS_DOUBLES * sizeof(double), h->stream,
                       reinterpret_cast<const TrainModel*>(dev), cm);
  if (hipGetLastError() != hipSuccess) return FFGP_ERR_HIP;
  int* st = reinterpret_cast<int*>(host + info_off);
  FFGP_HIP(hipMemcpyAsync(st, cm.info, (size_t)2 * F * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  FFGP_HIP(hipStreamSynchronize(h->stream));
  ffgp_invalidate(h);
  if (cm.prof) {
    long pr[12];
    hipMemcpy(pr, cm.prof, sizeof(pr), hipMemcpyDeviceToHost);
    hipFree(cm.prof);
    static const char* const nm[10] = {"links+Xs", "assemble", "F(jj)", "solve", "update", "inverse", "Gamma/A", "Sigma^-1+grad", "reduce", "adam"};
    fprintf(stderr, "[ffgp] train_persist n=%d D=%d d=%d steps=%d, us per step:", p[0].n, p[0].D, p[0].d, steps);
    double tot = 0.0;
    for (int k = 0; k < 10; ++k) {
      fprintf(stderr, " %s %.2f", nm[k], pr[k] * 0.01 / steps);
      tot += pr[k] * 0.01 / steps;
    }
    fprintf(stderr, " | sum %.2f\n", tot);
  }
  for (int f = 0; f < F; ++f)
    if (st[f] != 0) return st[f];
  return FFGP_OK;
}

// ---- evaluate mode: one likelihood (+ gradients) of F small problems, ONE launch per eight of them ---------------------------------
// What ffgp_nlml_fused_small_batch and, for 40 < n <= 128, ffgp_nlml_fused_raw run instead of the scalar one-workgroup kernel of
// small.hip (0.58 ms at n = 128) or ~13 launches of the blocked path (0.098 ms): the trainer's pass without Adam.
bool ffgp_small_mfma_ok(const ffgp_handle* h, const ffgp_problem* p, const ffgp_grads* g) {
  if (h->train_persist_off || h->use_naive || h->timing || h->small_off) return false;
  if (p->n <= 0 || p->n > TR_N || p->D <= 0 || p->D > TR_D || p->d <= 0 || p->d > TR_Y) return false;
  if (p->cov_dev || p->pair || p->tree || p->add_mat_dev || p->add_all != 0.0 || p->mean_jitter != 0.0) return false;
  if (!p->X_dev || !p->Y_dev || !p->w_dev || !p->amp_dev) return false;
  if (p->kfun < FFGP_KFUN_SE || p->kfun > FFGP_KFUN_RQ || p->ll_variant != FFGP_LL_V1) return false;
  if (g && (g->g_cov_dev || g->g_pair || g->g_kparam_dev)) return false;
  return true;
}

int ffgp_small_mfma_enqueue(ffgp_handle* h, int F, const ffgp_problem* p, const ffgp_links* l, double* nll_dev, const ffgp_grads* g,
                            int info_max) {
  static bool attr_set[64] = {false};
  if (h->device >= 0 && h->device < 64 && !attr_set[h->device]) {
    FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_small_mfma_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 TR_LDS_DOUBLES * (int)sizeof(double)));
    FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_small_mfma_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 TR_LDS_DOUBLES * (int)sizeof(double)));
    attr_set[h->device] = true;
  }
  if (!h->small_kbuf) FFGP_HIP(hipMalloc(&h->small_kbuf, (size_t)TR_EVAL_BATCH * TR_NBLK * 256 * sizeof(double)));
  TrainCommon cm;
  memset(&cm, 0, sizeof(cm));
  cm.steps = 1;
  cm.shared_info = h->d_info;
  cm.info_max = info_max;
  for (int f0 = 0; f0 < F; f0 += TR_EVAL_BATCH) {
    const int nb = std::min(TR_EVAL_BATCH, F - f0);
    TrainEvalBatch b;
    memset(&b, 0, sizeof(b));
    int Dmax = 0;
    for (int z = 0; z < nb; ++z) {
      const ffgp_problem& q = p[f0 + z];
      const ffgp_grads* gg = g ? g + f0 + z : nullptr;
      TrainModel& m = b.m[z];
      m.n = q.n; m.D = q.D; m.d = q.d;
      m.X = q.X_dev; m.Y = q.Y_dev;
      m.w = const_cast<double*>(q.w_dev); m.amp = const_cast<double*>(q.amp_dev); m.dadd = const_cast<double*>(q.diag_add_dev);
      m.diag_vec = q.diag_vec_dev; m.diag_stride = q.diag_stride;
      if (l) {
        m.l = l[f0 + z];
      } else {      // effective parameters: identity links
        memset(&m.l, 0, sizeof(m.l));
        m.l.w_link = m.l.amp_link = m.l.dadd_link = FFGP_LINK_ID;
      }
      m.nw = m.l.w_broadcast ? 1 : q.D;
      m.clamp = q.clamp_min; m.rinv = (q.kparam != 0.0) ? 1.0 / q.kparam : 1.0; m.pi_const = q.pi_const; m.kfun = q.kfun;
      m.nll = nll_dev + f0 + z;
      if (gg) {
        m.g_w = gg->g_w_dev; m.g_amp = gg->g_amp_dev; m.g_dadd = gg->g_diag_add_dev; m.g_Y = gg->g_Y_dev; m.g_dvec = gg->g_diag_vec_dev;
      }
      m.want_grad = (m.g_w || m.g_amp || m.g_dadd || m.g_Y || m.g_dvec) ? 1 : 0;
      m.kbuf = h->small_kbuf + (size_t)z * TR_NBLK * 256;
      Dmax = std::max(Dmax, q.D);
    }
    if (Dmax <= 8)
      hipLaunchKernelGGL(ffgp_small_mfma_kernel<8>, dim3(nb), dim3(TR_T), TR_LDS_DOUBLES * sizeof(double), h->stream, b, cm);
    else
      hipLaunchKernelGGL(ffgp_small_mfma_kernel<16>, dim3(nb), dim3(TR_T), TR_LDS_DOUBLES * sizeof(double), h->stream, b, cm);
  }
  return hipGetLastError() == hipSuccess ? FFGP_OK : FFGP_ERR_HIP;
}
