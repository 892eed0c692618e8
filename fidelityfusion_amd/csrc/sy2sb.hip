// Symmetric eigensolver, stage 1: dense -> band (bandwidth 32) by blocked Householder transformations.
//
// Replaces the first half of LAPACK's syevd underneath `torch.linalg.eigh(K)` in the HOGP block of GAR
// (FidelityFusion_Models/two_fidelity_models/hogp_simple.py:15-19,97-100; MFGP_ver2023May/base_gp/hogp.py:20-24).
// CPU restatement of this file, stage by stage: the tests' numpy model eigh_twostage.py (sy2sb, tsqr_hr).
//
// Per panel p (32 columns j0 = 32p, acting on the m = n - j0 - 32 rows below the band):
//   1. TSQR of the m x 32 panel: leaf Householder QRs of 512 rows (one workgroup of 1024 threads each, the block in registers, a
//      half-wave per column), then one QR of the stacked R factors                                   [sy2sb_leaf_qr, sy2sb_top]
//   2. Householder reconstruction (Ballard et al. 2015): the panel's thin Q1 is never formed; its top 32 x 32 block is
//      assembled from 32 x 32 pieces, a modified LU of [I;0] - Q1 S gives the unit-lower Y1, the sign matrix S, U and
//      T = U Y1^-T; the rows below are Y = Q_leaf (-Q_top,i S U^-1) = [G_i; 0] - V_i (X_i G_i)      [sy2sb_top, sy2sb_form_y]
//      -> a compact-WY reflector I - Y T Y^T of the whole panel although no workgroup ever saw more than 512 of its rows.
//      (Cholesky-QR is not an option: kernel matrices are numerically rank deficient, the Gram step squares that.)
//   3. Yp = A22 Y (the fp64 matrix-core GEMM, split along k), G = Y^T Yp, M = T^T G T,  W = Yp T - 1/2 Y M   [sy2sb_w]
//   4. A22 <- A22 - [Y W] [W Y]^T  : ONE rank-64 GEMM, alpha = -1, beta = 1 (the fast form of gemm.hip)
// The band (diagonal blocks + the S R blocks) is collected in compact storage AB[c * 64 + (r - c)].
//
// n must be a multiple of 64 (ffgp_syevd pads with decoupled diagonal entries: every reflector component on a padded row is
// exactly zero, so they never mix).  Up to 16 leaves of 512 rows the TSQR has two levels; longer panels (n > 8224) get a middle level
// (the leaf kernel again, on stacks of 16 R factors), n <= 131072.
#include "ffgp_internal.h"
#include "syevd_internal.h"

#define QR_ROWS 512   // rows of one TSQR leaf
#ifdef FFGP_QR_STAMPS   // development probe (tools/native/qr_phases.hip): 100 MHz clock stamps of workgroup 0, thread 0
__device__ unsigned long long ffgp_qr_stamp[64];
#define QR_STAMP(k)                                                              \
  do {                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                           \
    __builtin_amdgcn_s_waitcnt(0);                                               \
    if (blockIdx.x == 0 && threadIdx.x == 0) ffgp_qr_stamp[k] = wall_clock64();  \
    __builtin_amdgcn_sched_barrier(0);                                           \
  } while (0)
#else
#define QR_STAMP(k)
#endif
#define QR_THREADS 1024

// (qr_dpp_add, qr_rdlane, qr_wsum32: syevd_internal.h)
__device__ __forceinline__ double qr_rcp(double d) {
  double y = __builtin_amdgcn_rcp(d);
  y = __builtin_fma(y, __builtin_fma(-d, y, 1.0), y);
  y = __builtin_fma(y, __builtin_fma(-d, y, 1.0), y);
  return y;
}

struct QrShared {
  double v[2][QR_ROWS];   // the current reflector, double-buffered: one barrier per column
  double tauc[2];
  double tau[32];
  double H[32][33];   // H[c][j] = V_c^T v_j  (c < j)
  double T[32][33];
};

// Householder QR of a (<= 512) x 32 block by 1024 threads: a half-wave (32 lanes) per column, lane i holds the rows i + 32 r
// (r = 0..15) in a[r]; rows beyond the block must be zero.  Every column sum is a half-wave reduction (no LDS partial sums), the
// reflector travels through LDS, one workgroup barrier per column.  On exit a[] holds V below the diagonal and R on / above it
// (a[0] of lane i <= c), sh.T the compact-WY T, sh.tau the scalar factors.  Ends with a barrier.
__device__ __forceinline__ void qr512(double (&a)[16], QrShared& sh, const int tid) {
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, i = lane & 31;
  const int c = wave * 2 + half;
  for (int j = 0; j < 32; ++j) {
    const int buf = j & 1;
    if (wave == (j >> 1)) {   // the wave that holds column j: its norm, the reflector, tau
      double ss = (i > j) ? a[0] * a[0] : 0.0;
#pragma unroll
      for (int r = 1; r < 16; ++r) ss = __builtin_fma(a[r], a[r], ss);
      const double sigma = qr_wsum32(ss, lane);
      const double alpha = qr_rdlane(a[0], (j & 1) * 32 + j);
      double tau = 0.0, beta = alpha, scale = 0.0;
      if (sigma != 0.0) {
        const double q = __builtin_fma(alpha, alpha, sigma);
        double rs = __builtin_amdgcn_rsq(q);
        rs = __builtin_fma(0.5 * rs, __builtin_fma(-q * rs, rs, 1.0), rs);
        rs = __builtin_fma(0.5 * rs, __builtin_fma(-q * rs, rs, 1.0), rs);
        double nrm = q * rs;
        nrm = __builtin_fma(0.5 * rs, __builtin_fma(-nrm, nrm, q), nrm);
        beta = (alpha >= 0.0) ? -nrm : nrm;
        tau = (beta - alpha) * qr_rcp(beta);
        scale = qr_rcp(alpha - beta);
      }
      if (half == (j & 1)) {
        const double v0 = (i > j) ? a[0] * scale : ((i == j) ? 1.0 : 0.0);
        sh.v[buf][i] = v0;
        a[0] = (i > j) ? v0 : ((i == j) ? beta : a[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) {
          a[r] *= scale;
          sh.v[buf][i + 32 * r] = a[r];
        }
        if (i == 0) {
          sh.tauc[buf] = tau;
          sh.tau[j] = tau;
        }
      }
    }
    __syncthreads();
    double vr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) vr[r] = sh.v[buf][i + 32 * r];
    double s = 0.0;
    if (c != j) {   // c > j: the column's projection on v;  c < j: V_c^T v_j for the T factor (v is zero above row j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s = __builtin_fma(vr[r], a[r], s);
    }
    const double dot = qr_wsum32(s, lane);
    if (c > j) {
      const double f = sh.tauc[buf] * dot;
#pragma unroll
      for (int r = 0; r < 16; ++r) a[r] = __builtin_fma(-f, vr[r], a[r]);
    } else if (c < j && i == 0) {
      sh.H[c][j] = dot;
    }
  }
  __syncthreads();
  {   // T row by row: T[t][j] = -tau_j sum_{cc = t}^{j-1} T[t][cc] H[cc][j].  The rows are independent recurrences: row t runs on
      // half-wave t with T[t][cc] in lane cc, every step one product per lane and a half-wave sum -- no barrier, no serial inner loop
      // (one row per LANE of a single wave, the entries in registers -- the form sy2sb_top's triangular solves take -- was measured
      //  slower here: 49 -> 55 us per leaf; the 32 extra registers spill beside the block's 16 rows per lane)
    const int t = c;
    double trow = (i == t) ? sh.tau[t] : 0.0;
#pragma unroll 4
    for (int j = 1; j < 32; ++j) {
      const double hv = sh.H[i][j];                       // (only read into the sum where i < j: the part the loop above wrote)
      const double prod = (i >= t && i < j) ? trow * hv : 0.0;
      const double ssum = qr_wsum32(prod, lane);
      if (i == j && j > t) trow = -sh.tau[j] * ssum;
    }
    sh.T[t][i] = trow;
  }
  __syncthreads();
}

struct LeafArgs {
  double* A; int lda;     // panel origin: row r0, column j0 of the matrix
  int m;                  // rows of the panel
  double* Rst;            // [L][32][32]
  double* Tst;            // [L][32][32]
};

__global__ __launch_bounds__(QR_THREADS) void sy2sb_leaf_qr(LeafArgs p) {
  __shared__ QrShared sh;
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, c = (tid >> 6) * 2 + (lane >> 5);
  const int row0 = blockIdx.x * QR_ROWS;
  const int nrows = min(QR_ROWS, p.m - row0);
  double* __restrict__ P = p.A + (size_t)row0 * p.lda;
  double a[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gr = i + 32 * r;
    a[r] = (gr < nrows) ? P[(size_t)gr * p.lda + c] : 0.0;
  }
  qr512(a, sh, tid);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gr = i + 32 * r;
    if (gr < nrows) P[(size_t)gr * p.lda + c] = a[r];
  }
  p.Rst[(size_t)blockIdx.x * 1024 + i * 32 + c] = (i <= c) ? a[0] : 0.0;
  p.Tst[(size_t)blockIdx.x * 1024 + tid] = sh.T[tid >> 5][tid & 31];
}

// ---- the same QR on 256 threads (option "sb_qr4", off by default) ----------------------------------------------------------------
// A half-wave holds FOUR columns (4 hw .. 4 hw + 3; lane i the rows i + 32 r in a[q][r]).  Against qr512: the reflector is read
// from LDS once per wave and used for four columns, and nobody waits for the owner's norm: the owner publishes the RAW column below
// the diagonal (zeros at and above it) plus row j of the block, and every half-wave derives sigma, tau and the scale itself, next to
// its own four dot products --
//   v = [1; s x],  v^T a_c = a_jc + s x^T a_c,   a_c <- a_c - tau (v^T a_c) v,   H[c][j] = v^T V_c by the same expression (c < j).
// One barrier per column.  Measured (N = 8192, kernel trace): 58.9 us per panel against 54.4 us for the 1024-thread kernel -- NOT
// faster.  tools/native/qr_phases.hip (clock stamps, workgroup 0): load 4.2, 32 columns 37.6 (1.16 per column: LDS 0.12, dots 0.16,
// five half-wave sums 0.36, scalars 0.16, updates 0.32, publish 0.12, barrier 0.16), T build 11.5 (four rows per half-wave, one
// after the other), store 3.8 us.  What it does do: a workgroup of one wave per SIMD finds room beside a running GEMM, so under
// "sb_lookahead" the leaves end before the trailing update instead of 60 us after it -- and still the stage is no faster (85.2 ms),
// because they take 127 us there and sy2sb_top (1024 threads) runs at the update's tail.
#define QR4_THREADS 256
struct Qr4Shared {
  double x[2][QR_ROWS];    // raw column j below the diagonal, double-buffered
  double rowj[2][32];      // row j of the block after reflector j - 1
  double tau[32];
  double H[32][33];
  double T[32][33];
};

__device__ __forceinline__ void qr512x4(double (&a)[4][16], Qr4Shared& sh, const int tid) {
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, i = lane & 31;
  const int hw = wave * 2 + half;
  if (hw == 0) {
    sh.x[0][i] = (i > 0) ? a[0][0] : 0.0;
#pragma unroll
    for (int r = 1; r < 16; ++r) sh.x[0][i + 32 * r] = a[0][r];
  }
  if (i == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) sh.rowj[0][4 * hw + q] = a[q][0];
  }
  __syncthreads();
  for (int jb = 0; jb < 8; ++jb) {
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) {
      const int j = 4 * jb + jq, buf = j & 1;
      if (jq == 0) QR_STAMP(8 + jb);                 // start of every block of four columns
      if (j == 16) QR_STAMP(20);
      double xr[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) xr[r] = sh.x[buf][i + 32 * r];
      const double alpha = sh.rowj[buf][j];
      double arow[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) arow[q] = sh.rowj[buf][4 * hw + q];
      if (j == 16) QR_STAMP(21);                     // reflector and row j are in registers
      double ss = 0.0, d[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        ss = __builtin_fma(xr[r], xr[r], ss);
#pragma unroll
        for (int q = 0; q < 4; ++q) d[q] = __builtin_fma(xr[r], a[q][r], d[q]);
      }
      if (j == 16) QR_STAMP(22);                     // five dot products per lane
      const double sigma = qr_wsum32(ss, lane);
#pragma unroll
      for (int q = 0; q < 4; ++q) d[q] = qr_wsum32(d[q], lane);
      if (j == 16) QR_STAMP(23);                     // five half-wave sums
      double tau = 0.0, beta = alpha, scale = 0.0;
      if (sigma != 0.0) {
        const double qq = __builtin_fma(alpha, alpha, sigma);
        double rs = __builtin_amdgcn_rsq(qq);
        rs = __builtin_fma(0.5 * rs, __builtin_fma(-qq * rs, rs, 1.0), rs);
        rs = __builtin_fma(0.5 * rs, __builtin_fma(-qq * rs, rs, 1.0), rs);
        double nrm = qq * rs;
        nrm = __builtin_fma(0.5 * rs, __builtin_fma(-nrm, nrm, qq), nrm);
        beta = (alpha >= 0.0) ? -nrm : nrm;
        tau = (beta - alpha) * qr_rcp(beta);
        scale = qr_rcp(alpha - beta);
      }
      if (j == 16) QR_STAMP(24);                     // beta, tau, scale
      const int nb = buf ^ 1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 4 * hw + q;
        const double w = __builtin_fma(scale, d[q], arow[q]);     // v^T a_c  (c > j)  =  v^T V_c  (c < j)
        if (c > j) {
          const double f = tau * w, g = f * scale;
#pragma unroll
          for (int r = 0; r < 16; ++r) a[q][r] = __builtin_fma(-g, xr[r], a[q][r]);
          if (i == j) a[q][0] -= f;
        } else if (c < j) {
          if (i == 0) sh.H[c][j] = w;
        }
      }
      if (j == 16) QR_STAMP(25);                     // four column updates
      if (hw == jb) {          // the owner: V below the diagonal, beta on it
        a[jq][0] = (i > j) ? xr[0] * scale : ((i == j) ? beta : a[jq][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) a[jq][r] = xr[r] * scale;
        if (i == 0) sh.tau[j] = tau;
      }
      if (j < 31) {            // publish column j + 1 (raw, below its diagonal) and row j + 1
        const int nq = (jq + 1) & 3;      // (a compile-time index once the jq loop is unrolled)
        const int nhw = (j + 1) >> 2;
        if (hw == nhw) {
          sh.x[nb][i] = (i > j + 1) ? a[nq][0] : 0.0;
#pragma unroll
          for (int r = 1; r < 16; ++r) sh.x[nb][i + 32 * r] = a[nq][r];
        }
        if (i == j + 1) {
#pragma unroll
          for (int q = 0; q < 4; ++q) sh.rowj[nb][4 * hw + q] = a[q][0];
        }
      }
      if (j == 16) QR_STAMP(26);                     // next column published
      __syncthreads();
      if (j == 16) QR_STAMP(27);                     // barrier
    }
  }
  QR_STAMP(16);
  // T: row t on a half-wave, four rows each (see qr512)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int t = 4 * hw + q;
    double trow = (i == t) ? sh.tau[t] : 0.0;
#pragma unroll 4
    for (int j = 1; j < 32; ++j) {
      const double hv = sh.H[i][j];
      const double ssum = qr_wsum32((i >= t && i < j) ? trow * hv : 0.0, lane);
      if (i == j && j > t) trow = -sh.tau[j] * ssum;
    }
    sh.T[t][i] = trow;
  }
  __syncthreads();
}

// leaf QR on qr512x4: every lane moves 32 bytes (its four columns) per row
__global__ __launch_bounds__(QR4_THREADS) void sy2sb_leaf_qr4(LeafArgs p) {
  __shared__ Qr4Shared sh;
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, hw = (tid >> 6) * 2 + (lane >> 5);
  const int row0 = blockIdx.x * QR_ROWS;
  const int nrows = min(QR_ROWS, p.m - row0);
  double* __restrict__ P = p.A + (size_t)row0 * p.lda + 4 * hw;
  QR_STAMP(0);
  double a[4][16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gr = i + 32 * r;
    d4_t v = d4_t{0.0, 0.0, 0.0, 0.0};
    if (gr < nrows) v = *reinterpret_cast<const d4_t*>(P + (size_t)gr * p.lda);
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q][r] = v[q];
  }
  QR_STAMP(1);
  qr512x4(a, sh, tid);
  QR_STAMP(2);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gr = i + 32 * r;
    if (gr < nrows) *reinterpret_cast<d4_t*>(P + (size_t)gr * p.lda) = d4_t{a[0][r], a[1][r], a[2][r], a[3][r]};
  }
  d4_t rv;
#pragma unroll
  for (int q = 0; q < 4; ++q) rv[q] = (i <= 4 * hw + q) ? a[q][0] : 0.0;
  *reinterpret_cast<d4_t*>(p.Rst + (size_t)blockIdx.x * 1024 + i * 32 + 4 * hw) = rv;
  for (int idx = tid; idx < 1024; idx += QR4_THREADS) p.Tst[(size_t)blockIdx.x * 1024 + idx] = sh.T[idx >> 5][idx & 31];
  QR_STAMP(3);
}

// ---- small dense helpers on 32 x 32 LDS matrices (leading dimension 33), 256 threads, 4 outputs per thread -------------
typedef double M33[32][33];
// C = alpha * A op(B)   (TB: B transposed)
template <bool TB>
__device__ __forceinline__ void mm32(M33& C, const M33& A, const M33& B, double alpha, int tid) {
  const int i = tid >> 3, j0 = (tid & 7) * 4;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int k = 0; k < 32; ++k) {
    const double av = A[i][k];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = __builtin_fma(av, TB ? B[j0 + q][k] : B[k][j0 + q], acc[q]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) C[i][j0 + q] = alpha * acc[q];
}

struct TopArgs {
  const double* A; int lda;   // panel origin (leaf V's in place)
  int m, L;
  const double* Rst; const double* Tst;   // Rst: the R stack the top QR factors ([L][32][32]);  Tst: T of the panel's first leaf
  const double* Vmid0; const double* Tmid0;   // three levels: first 32 rows of the first middle QR's V (in place in the leaves' R stack), its T
  int three;
  double* Vtst;               // [512][32]  V of the top-level QR
  double* small;              // Xt [1024] | S [32] | Uinv [1024]
  double* Tpan;               // [32][32]   T of this panel
  double* Y; int ldy;         // Y store, origin (r0, j0): rows 0..31 <- Y1
  double* AB;                 // band storage origin of column j0
  int use_tree;               // L > 1
};

// One workgroup: QR of the stacked R factors, top block of Q1, modified LU, T, U^-1.  (The QR runs on all 1024 threads, the
// 32 x 32 algebra after it on the first 256.)
__global__ __launch_bounds__(QR_THREADS) void sy2sb_top(TopArgs p) {
  __shared__ QrShared sh;
  __shared__ M33 V1, Xt, Q0, Wt;
  __shared__ double Ssign[32];
  M33& M1 = sh.H;   // free once qr512 has built T
  M33& M2 = sh.T;   // free once Tt has been copied
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, c = (tid >> 6) * 2 + (lane >> 5);
  const bool sm = tid < 256;   // the threads of the small-matrix phases
  double r0v;       // R[i][c] of the panel's R factor
  QR_STAMP(30);
  if (p.use_tree) {
    double a[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = (r < p.L) ? p.Rst[(size_t)(i + 32 * r) * 32 + c] : 0.0;   // [L][32][32] = stacked rows
    QR_STAMP(31);
    qr512(a, sh, tid);
    QR_STAMP(32);
    // explicit V of the top QR (unit diagonal, zeros above) for the leaf workgroups
    const double v0 = (i > c) ? a[0] : ((i == c) ? 1.0 : 0.0);
    p.Vtst[(size_t)i * 32 + c] = v0;
#pragma unroll
    for (int r = 1; r < 16; ++r) p.Vtst[(size_t)(i + 32 * r) * 32 + c] = a[r];
    r0v = (i <= c) ? a[0] : 0.0;
    V1[i][c] = v0;
    M1[tid >> 5][tid & 31] = sh.T[tid >> 5][tid & 31];
    __syncthreads();
    if (sm) mm32<true>(Xt, M1, V1, 1.0, tid);      // Xt = Tt V1^T
    __syncthreads();
    if (sm) mm32<false>(Q0, V1, Xt, -1.0, tid);    // Q0 = I - V1 Xt
    __syncthreads();
    if (tid < 32) Q0[tid][tid] += 1.0;
  } else {
    r0v = p.Rst[i * 32 + c];
    Xt[i][c] = 0.0;
    Q0[i][c] = (i == c) ? 1.0 : 0.0;
  }
  __syncthreads();
  QR_STAMP(33);
  // X0 = T0 V0top^T ; Wtop = (I - V0top X0) Q0
  {
    M1[i][c] = p.Tst[i * 32 + c];
    const double av = p.A[(size_t)i * p.lda + c];
    V1[i][c] = (i > c) ? av : ((i == c) ? 1.0 : 0.0);
  }
  __syncthreads();
  if (sm) mm32<true>(M2, M1, V1, 1.0, tid);        // X0
  __syncthreads();
  if (sm) mm32<false>(M1, V1, M2, -1.0, tid);      // -V0top X0
  __syncthreads();
  if (tid < 32) M1[tid][tid] += 1.0;
  __syncthreads();
  if (p.three) {   // a middle TSQR level sits between the leaves and the top: one more 32 x 32 factor I - Vm0 Xm0 behind leaf 0
    M2[i][c] = p.Tmid0[i * 32 + c];
    const double av = p.Vmid0[i * 32 + c];
    V1[i][c] = (i > c) ? av : ((i == c) ? 1.0 : 0.0);
    __syncthreads();
    if (sm) mm32<true>(Wt, M2, V1, 1.0, tid);      // Xm0 = Tm0 Vm0^T
    __syncthreads();
    if (sm) mm32<false>(M2, V1, Wt, -1.0, tid);    // -Vm0 Xm0
    __syncthreads();
    if (tid < 32) M2[tid][tid] += 1.0;
    __syncthreads();
    if (sm) mm32<false>(Wt, M1, M2, 1.0, tid);
    __syncthreads();
    M1[i][c] = Wt[i][c];
    __syncthreads();
  }
  if (sm) mm32<false>(Wt, M1, Q0, 1.0, tid);       // top block of Q1
  __syncthreads();
  QR_STAMP(34);
  // modified LU of [I;0] - Q1 S (top block): signs chosen so that every pivot is >= 1 in magnitude.  V1 <- Y1 (unit lower).
  V1[i][c] = (i == c) ? 1.0 : 0.0;
  __syncthreads();
  for (int j = 0; j < 32; ++j) {
    const double wjj = Wt[j][j];
    const double sj = (wjj >= 0.0) ? -1.0 : 1.0;
    const double piv = 1.0 - sj * wjj;
    if (tid < 32 && tid > j) V1[tid][j] = -sj * Wt[tid][j] / piv;
    if (tid == 0) Ssign[j] = sj;
    __syncthreads();
    if (i > j && c > j) Wt[i][c] = __builtin_fma(-V1[i][j], Wt[j][c], Wt[i][c]);   // thread (i, c): one entry each
    __syncthreads();
  }
  QR_STAMP(35);
  // U (upper) -> M1;  T = U Y1^-T -> M2 (row i: forward substitution over the columns);  U^-1 -> Q0 (column by column)
  M1[i][c] = (c >= i) ? (((i == c) ? 1.0 : 0.0) - Ssign[c] * Wt[i][c]) : 0.0;
  __syncthreads();
  // Both triangular solves, one ROW (of T = U Y1^-T, forward over the columns) or COLUMN (of U^-1, backward over the rows) per LANE with
  // its 32 entries in registers: every coefficient (Y1[j][i], U[ii][i]) is the same for all lanes -- one broadcast LDS read -- and a step
  // is a chain of plain multiply-adds.  (They ran as half-wave recurrences, one entry per lane and a half-wave sum per step, on all
  // sixteen waves: 2 x 32 steps of ~50 instructions of reduction, four waves deep on every SIMD -- 14 us; here wave 0 does T while
  // wave 1 does U^-1, 496 multiply-adds each.)
  if (tid < 32) {                    // row c = tid of T:  T[c][j] = U[c][j] - sum_{i<j} T[c][i] Y1[j][i]
    const int cr = tid;
    double t[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      double acc_ = M1[cr][j];
#pragma unroll
      for (int i2 = 0; i2 < j; ++i2) acc_ = __builtin_fma(-t[i2], V1[j][i2], acc_);
      t[j] = acc_;
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) M2[cr][j] = t[j];
  } else if (tid >= 64 && tid < 96) {   // column c = tid - 64 of U^-1:  X[ii][c] = (delta - sum_{i>ii} U[ii][i] X[i][c]) / U[ii][ii], zero below the diagonal
    const int cc_ = tid - 64;
    double q[32];
#pragma unroll
    for (int jj = 0; jj < 32; ++jj) {
      const int ii = 31 - jj;
      double acc_ = (ii == cc_) ? 1.0 : 0.0;
#pragma unroll
      for (int i2 = ii + 1; i2 < 32; ++i2) acc_ = __builtin_fma(-M1[ii][i2], q[i2], acc_);
      q[ii] = (ii > cc_) ? 0.0 : acc_ / M1[ii][ii];
    }
#pragma unroll
    for (int ii = 0; ii < 32; ++ii) Q0[ii][cc_] = q[ii];
  }
  __syncthreads();
  QR_STAMP(36);
  {
    const int idx = i * 32 + c;
    p.small[idx] = Xt[i][c];
    p.small[1024 + 32 + idx] = Q0[i][c];
    p.Tpan[idx] = M2[i][c];
    p.Y[(size_t)i * p.ldy + c] = V1[i][c];
    // band: row r0 + i, column j0 + c of S R, kept where i <= c: offset (32 + i - c)
    if (i <= c) p.AB[(size_t)c * SB_LDB + (32 + i - c)] = Ssign[i] * r0v;
  }
  if (tid < 32) p.small[1024 + tid] = Ssign[tid];
  QR_STAMP(37);
}

struct FormYArgs {
  const double* A; int lda;   // panel origin
  int m, L;
  const double* Tst; const double* Vtst; const double* small;
  const double* Vmid; const double* Tmid;   // three levels: the middle QRs' V (in place in the leaves' R stack, [L * 32][32]) and T ([groups][32][32])
  int three;
  double* Y; int ldy;
  int use_tree;
};

// rows of the reconstructed Y below the top block: Y = [G_i; 0] - V_i (X_i G_i),  G_i = -(Q_top,i S) U^-1.  Two workgroups
// per leaf (256 rows each); every workgroup rebuilds its leaf's 32 x 32 pieces.
__global__ __launch_bounds__(256) void sy2sb_form_y(FormYArgs p) {
  __shared__ M33 Ma, Mb, Mc, G, H;
  const int tid = threadIdx.x;
  const int leaf = blockIdx.x >> 1, half = blockIdx.x & 1;
  const int row0 = leaf * QR_ROWS;
  const int nrows = min(QR_ROWS, p.m - row0);
  if (half * 256 >= nrows) return;
  const double* Xt = p.small;
  const double* Ssign = p.small + 1024;
  const double* Uinv = p.small + 1024 + 32;
  // the 32 x 32 factor behind this leaf's thin Q:  Q_top,g = delta_g0 I - Vt_g Xt  (g = the leaf itself with two levels, its group of
  // 16 with three), times, with three levels, the middle factor  delta_l0 I - Vmid_g[block l] Xmid_g
  const int g = p.three ? (leaf >> 4) : leaf, l = leaf & 15;
  for (int idx = tid; idx < 1024; idx += 256) {
    const int i = idx >> 5, j = idx & 31;
    Ma[i][j] = p.use_tree ? p.Vtst[(size_t)(g * 32 + i) * 32 + j] : 0.0;
    Mb[i][j] = Xt[idx];
  }
  __syncthreads();
  mm32<false>(Mc, Ma, Mb, -1.0, tid);
  __syncthreads();
  if (tid < 32 && g == 0) Mc[tid][tid] += 1.0;
  __syncthreads();
  if (p.three) {
    const double* Vg = p.Vmid + (size_t)g * 16 * 1024;           // the group's stacked block (512 x 32), V in place
    for (int idx = tid; idx < 1024; idx += 256) {
      const int i = idx >> 5, j = idx & 31;
      Ma[i][j] = p.Tmid[(size_t)g * 1024 + idx];
      const double av = Vg[idx];
      Mb[i][j] = (i > j) ? av : ((i == j) ? 1.0 : 0.0);
    }
    __syncthreads();
    mm32<true>(G, Ma, Mb, 1.0, tid);                             // Xmid_g = Tmid_g Vmid_g[:32]^T
    __syncthreads();
    for (int idx = tid; idx < 1024; idx += 256) {
      const int i = idx >> 5, j = idx & 31;
      const double av = Vg[(size_t)l * 1024 + idx];
      Ma[i][j] = (l == 0) ? ((i > j) ? av : ((i == j) ? 1.0 : 0.0)) : av;
    }
    __syncthreads();
    mm32<false>(H, Ma, G, -1.0, tid);                            // -Vmid_g[block l] Xmid_g
    __syncthreads();
    if (tid < 32 && l == 0) H[tid][tid] += 1.0;
    __syncthreads();
    mm32<false>(G, H, Mc, 1.0, tid);                             // middle factor times top factor
    __syncthreads();
    for (int idx = tid; idx < 1024; idx += 256) Mc[idx >> 5][idx & 31] = G[idx >> 5][idx & 31];
    __syncthreads();
  }
  for (int idx = tid; idx < 1024; idx += 256) {
    const int i = idx >> 5, j = idx & 31;
    Ma[i][j] = -Mc[i][j] * Ssign[j];
    Mb[i][j] = Uinv[idx];
  }
  __syncthreads();
  mm32<false>(G, Ma, Mb, 1.0, tid);        // G_i
  __syncthreads();                         // every wave is done reading Ma, Mb before they are refilled (without this barrier a wave
                                           // that fell behind -- another kernel's waves on the CU are enough -- multiplied the NEW contents:
                                           // 256 rows of Y wrong in a few columns, found by running two band reductions at once)
  // X_i = T_i V_i[:32]^T
  const double* P = p.A + (size_t)row0 * p.lda;
  for (int idx = tid; idx < 1024; idx += 256) {
    const int i = idx >> 5, j = idx & 31;
    Ma[i][j] = p.Tst[(size_t)leaf * 1024 + idx];
    const double av = P[(size_t)i * p.lda + j];
    Mb[i][j] = (i > j) ? av : ((i == j) ? 1.0 : 0.0);
  }
  __syncthreads();
  mm32<true>(Mc, Ma, Mb, 1.0, tid);        // X_i
  __syncthreads();
  mm32<false>(H, Mc, G, 1.0, tid);         // H_i = X_i G_i
  __syncthreads();
  const int lr = half * 256 + tid;         // row inside the leaf
  if (lr >= nrows) return;
  if (leaf == 0 && lr < 32) return;        // the panel's top block holds Y1 (written by sy2sb_top)
  double vrow[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const double av = P[(size_t)lr * p.lda + k];
    vrow[k] = (lr < 32) ? ((lr > k) ? av : ((lr == k) ? 1.0 : 0.0)) : av;
  }
  double* yrow = p.Y + (size_t)(row0 + lr) * p.ldy;
  for (int j = 0; j < 32; ++j) {
    double s = (lr < 32) ? G[lr][j] : 0.0;
#pragma unroll
    for (int k = 0; k < 32; ++k) s = __builtin_fma(-vrow[k], H[k][j], s);
    yrow[j] = s;
  }
}

// Yp-partials = A22[:, chunk] Y[chunk, :]: m x m times m x 32, the one place of the band reduction that reads the trailing matrix
// without writing it.  On the general GEMM (64 x 64 tiles, half of every tile's 64 columns empty) this took two launches of 39 us per
// panel at N = 8192 (kernel trace), about 20 of the stage's 104 ms.  Here a workgroup owns 128 rows and one k chunk: the A tile of a
// k step (128 x 16) arrives by 16-byte loads, goes through LDS into the MFMA lane layout, and every wave multiplies its 32 rows by
// the chunk of Y (16 x 32, also in LDS): 16 MFMAs per wave and step.  Traced: 48 us per panel, 12.2 ms of the stage (about 3.7 TB/s
// on the average trailing matrix).  Two tiles in flight instead of one measured the same: not bound by load latency.
struct AvArgs {
  const double* A; int lda;      // A22, m x m row-major
  const double* Y; int ldy;      // Y, m x 32
  double* P; long sP;            // partials [parts][m][32]
  int m, kc;                     // chunk length (multiple of 16); chunk c covers k in [c kc, min(m, (c + 1) kc))
};

__global__ __launch_bounds__(256, 4) void sy2sb_av(AvArgs p) {
  __shared__ double As[128][17];
  __shared__ double Ys[16][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lq = lane >> 4;
  const int r0 = blockIdx.x * 128;
  const int k0 = blockIdx.y * p.kc, k1 = min(p.m, k0 + p.kc);
  if (k0 >= k1) return;
  // staging maps: A tile 128 x 16 -- thread t takes row t / 2, eight doubles at column 8 (t % 2); Y chunk 16 x 32 -- two doubles per thread
  const int arow = tid >> 1, acol = (tid & 1) * 8;
  const double* __restrict__ ap = p.A + (size_t)min(r0 + arow, p.m - 1) * p.lda + acol;
  const int yk = tid >> 4, yc = (tid & 15) * 2;
  const double* __restrict__ yp = p.Y + yc;
  struct Tile { d4_t a0, a1; d2_t y; };
  auto fetch = [&](int k, Tile& t) {
    t.a0 = *reinterpret_cast<const d4_t*>(ap + k);
    t.a1 = *reinterpret_cast<const d4_t*>(ap + k + 4);
    t.y = *reinterpret_cast<const d2_t*>(yp + (size_t)(k + yk) * p.ldy);
  };
  d4_t acc[2][2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = d4_t{0.0, 0.0, 0.0, 0.0};
  auto step = [&](const Tile& t) {
    __syncthreads();                       // the previous step's operand reads are done
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      As[arow][acol + q] = t.a0[q];
      As[arow][acol + 4 + q] = t.a1[q];
    }
    Ys[yk][yc] = t.y.x;
    Ys[yk][yc + 1] = t.y.y;
    __syncthreads();
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      double av[2], bv[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) av[rt] = As[32 * wave + 16 * rt + lr][4 * s4 + lq];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) bv[ct] = Ys[4 * s4 + lq][16 * ct + lr];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[rt], bv[ct], acc[rt][ct], 0, 0, 0);
    }
  };
  // two tiles in flight: the loads of steps k + 16 and k + 32 travel under the products of step k
  Tile ta, tb;
  fetch(k0, ta);
  if (k0 + 16 < k1) fetch(k0 + 16, tb);
  for (int k = k0; k < k1; k += 32) {
    step(ta);
    if (k + 32 < k1) fetch(k + 32, ta);
    if (k + 16 < k1) {
      step(tb);
      if (k + 48 < k1) fetch(k + 48, tb);
    }
  }
  double* __restrict__ out = p.P + (size_t)blockIdx.y * p.sP;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r0 + 32 * wave + 16 * rt + 4 * r + lq;
      if (row < p.m) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) out[(size_t)row * 32 + 16 * ct + lr] = acc[rt][ct][r];
      }
    }
}

// Yp-partials from the LOWER triangle of A22 alone (round 6, option "sb_lower", default).  The trailing update is bound by HBM (rank 64:
// 8 flop per byte of C), and A22 Y read the whole matrix once more per panel; A22 is symmetric, so the update now writes only its lower
// triangle (half the bytes: the GEMM's lower-trapezoid tile set) and this kernel reads only that -- every element A[i][k], k < i, serves
// both  Yp[i] += A[i][k] Y[k]  and  Yp[k] += A[i][k] Y[i].  A workgroup owns 128 rows x one k chunk as before (chunks right of its rows
// belong to other workgroups' transposed side and exit at once); per k step (128 x 16 tile through LDS, strictly-upper entries of a
// tile on the diagonal stored as zeros) waves 0-1 compute the direct product for 64 rows each, waves 2-3 the transposed one (16 columns
// of A x one half of Y's 32 columns each, summed over all 128 rows: Y's rows of the workgroup sit in registers for the whole launch) --
// 32 MFMAs per wave and step either way, no reduction across waves.  Direct partials P[chunk][row][32] as before; transposed partials
// PT[row block][column][32]; sy2sb_red adds, for row j, the chunks left of its row block's end and the row blocks from its own
// downwards, in that fixed order.
struct AvSymArgs {
  const double* A; int lda;      // A22 (lower triangle valid), m x m row-major
  const double* Y; int ldy;      // Y, m x 32
  double* P; long sP;            // direct partials [parts][m][32]
  double* PT; long sPT;          // transposed partials [row blocks][m][32]
  int m, kc;
};

// (workgroup barrier that publishes LDS only: __syncthreads() also waits for the wave's global loads -- the next tiles, which are meant to
//  stay in flight across the step -- and for the transposed partials' stores, which nobody in this launch reads)
#define AV_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__global__ __launch_bounds__(256, 3) void sy2sb_av_sym(AvSymArgs p) {
  __shared__ double As[128][17];
  __shared__ double Ys[16][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lq = lane >> 4;
  const int r0 = blockIdx.x * 128;
  const int k0 = blockIdx.y * p.kc, k1 = min(min(p.m, k0 + p.kc), r0 + 128);
  if (k0 >= k1) return;
  const int arow = tid >> 1, acol = (tid & 1) * 8;
  const double* __restrict__ ap = p.A + (size_t)min(r0 + arow, p.m - 1) * p.lda + acol;
  const int yk = tid >> 4, yc = (tid & 15) * 2;
  const double* __restrict__ yp = p.Y + yc;
  struct Tile { d4_t a0, a1; d2_t y; };
  auto fetch = [&](int k, Tile& t) {
    t.a0 = *reinterpret_cast<const d4_t*>(ap + k);
    t.a1 = *reinterpret_cast<const d4_t*>(ap + k + 4);
    t.y = *reinterpret_cast<const d2_t*>(yp + (size_t)(k + yk) * p.ldy);
  };
  const bool direct = wave < 2;
  const int ct = wave & 1;                 // transposed waves: their half of Y's columns
  d4_t acc[4][2];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) acc[rt][c2] = d4_t{0.0, 0.0, 0.0, 0.0};
  // transposed waves keep Y[r0 + 4 s + lq][16 ct + lr] (zero beyond the matrix), s = 0 .. 31, in the registers the direct waves use as
  // accumulators: acc[s >> 3][(s >> 2) & 1][s & 3]
  if (!direct) {
#pragma unroll
    for (int s_ = 0; s_ < 32; ++s_) {
      const int row = r0 + 4 * s_ + lq;
      acc[s_ >> 3][(s_ >> 2) & 1][s_ & 3] = (row < p.m) ? p.Y[(size_t)row * p.ldy + 16 * ct + lr] : 0.0;
    }
  }
  double* __restrict__ outT = p.PT + (size_t)blockIdx.x * p.sPT;
  auto step = [&](const Tile& t, const int k) {
    AV_BARRIER();                          // the previous step's operand reads are done
    const bool on_diag = k + 15 > r0;      // the tile reaches the diagonal: entries right of it are not part of the lower triangle
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool up0 = on_diag && (k + acol + q > r0 + arow), up1 = on_diag && (k + acol + 4 + q > r0 + arow);
      As[arow][acol + q] = up0 ? 0.0 : t.a0[q];
      As[arow][acol + 4 + q] = up1 ? 0.0 : t.a1[q];
    }
    Ys[yk][yc] = t.y.x;
    Ys[yk][yc + 1] = t.y.y;
    AV_BARRIER();
    if (direct) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        double av[4], bv[2];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) av[rt] = As[64 * wave + 16 * rt + lr][4 * s4 + lq];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) bv[c2] = Ys[4 * s4 + lq][16 * c2 + lr];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2) acc[rt][c2] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[rt], bv[c2], acc[rt][c2], 0, 0, 0);
      }
    } else {
      d4_t at[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};      // even / odd k steps: two independent chains, added at the end
#pragma unroll
      for (int s_ = 0; s_ < 32; ++s_) {
        double a = As[4 * s_ + lq][lr];
        if (on_diag && (r0 + 4 * s_ + lq == k + lr)) a = 0.0;      // the diagonal entry belongs to the direct side only
        at[s_ & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[s_ >> 3][(s_ >> 2) & 1][s_ & 3], at[s_ & 1], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) outT[(size_t)(k + lq + 4 * r) * 32 + 16 * ct + lr] = at[0][r] + at[1][r];
    }
  };
  Tile ta, tb;
  fetch(k0, ta);
  if (k0 + 16 < k1) fetch(k0 + 16, tb);
  for (int k = k0; k < k1; k += 32) {
    step(ta, k);
    if (k + 32 < k1) fetch(k + 32, ta);
    if (k + 16 < k1) {
      step(tb, k + 16);
      if (k + 48 < k1) fetch(k + 48, tb);
    }
  }
  if (direct) {
    double* __restrict__ out = p.P + (size_t)blockIdx.y * p.sP;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + 64 * wave + 16 * rt + 4 * r + lq;
        if (row < p.m) {
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2) out[(size_t)row * 32 + 16 * c2 + lr] = acc[rt][c2][r];
        }
      }
  }
}

struct RedArgs {
  const double* P; long sP; int parts;   // partial products A22[:, chunk] Y[chunk, :], [parts][m][32]
  const double* PT; long sPT; int kc;    // lower-triangle form (sy2sb_av_sym; null otherwise): transposed partials [row blocks][m][32], chunk length
  const double* Y; int ldy;
  double* Yp;                            // [m][32]  their sum
  double* Gpart;                         // [workgroups][32][32]  Y_blk^T Yp_blk
  int m;
};

// Yp = sum of the k-chunks' partial products (fixed order), and this workgroup's 128 rows' share of G = Y^T Yp.
// 1024 threads: 8 per row (4 columns each).
__global__ __launch_bounds__(1024) void sy2sb_red(RedArgs p) {
  __shared__ double Ys[128][33], Ps[128][33];
  const int tid = threadIdx.x, lrow = tid >> 3, c4 = (tid & 7) * 4;
  const int row = blockIdx.x * 128 + lrow;
  double yp[4] = {0.0, 0.0, 0.0, 0.0}, y[4] = {0.0, 0.0, 0.0, 0.0};
  if (row < p.m) {
    const int rbend = p.PT ? (row & ~127) + 128 : 0x7fffffff;      // (lower-triangle form: chunks right of the row block were never computed)
    for (int s = 0; s < p.parts; ++s) {
      if (s * p.kc >= rbend) break;
      const d4_t v = *reinterpret_cast<const d4_t*>(p.P + (size_t)s * p.sP + (size_t)row * 32 + c4);
#pragma unroll
      for (int q = 0; q < 4; ++q) yp[q] += v[q];
    }
    if (p.PT) {
      const int nrb = (p.m + 127) >> 7;
      for (int rb = row >> 7; rb < nrb; ++rb) {
        const d4_t v = *reinterpret_cast<const d4_t*>(p.PT + (size_t)rb * p.sPT + (size_t)row * 32 + c4);
#pragma unroll
        for (int q = 0; q < 4; ++q) yp[q] += v[q];
      }
    }
    d4_t o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      o[q] = yp[q];
      y[q] = p.Y[(size_t)row * p.ldy + c4 + q];
    }
    *reinterpret_cast<d4_t*>(p.Yp + (size_t)row * 32 + c4) = o;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    Ys[lrow][c4 + q] = y[q];
    Ps[lrow][c4 + q] = yp[q];
  }
  __syncthreads();
  {   // 1024 outputs, one per thread, k = 128 rows
    const int a = tid >> 5, b = tid & 31;
    double acc = 0.0;
#pragma unroll 8
    for (int r = 0; r < 128; ++r) acc = __builtin_fma(Ys[r][a], Ps[r][b], acc);
    p.Gpart[(size_t)blockIdx.x * 1024 + tid] = acc;
  }
}

struct MArgs {
  const double* Tpan; const double* Gpart; int ngp;   // T [32][32]; Gpart [ngp][32][32], their sum = G = Y^T Yp
  double* Mh;                                         // [32][32]  1/2 T^T G T
};

// one workgroup: G = sum of the pieces, Mh = 1/2 T^T (G T)
__global__ __launch_bounds__(1024) void sy2sb_m(MArgs p) {
  __shared__ M33 T, G, M1;
  const int tid = threadIdx.x, i = tid >> 5, j = tid & 31;
  double g = 0.0;
  {   // (eight loads in flight, added in the order of the plain loop: 12.8 -> 9.8 us per panel at n = 8192; sixteen: 10.2)
    int q = 0;
    for (; q + 8 <= p.ngp; q += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p.Gpart[(size_t)(q + u) * 1024 + tid];
#pragma unroll
      for (int u = 0; u < 8; ++u) g += v[u];
    }
    for (; q < p.ngp; ++q) g += p.Gpart[(size_t)q * 1024 + tid];
  }
  G[i][j] = g;
  T[i][j] = p.Tpan[tid];
  __syncthreads();
  double acc = 0.0;
  for (int k = 0; k < 32; ++k) acc = __builtin_fma(G[i][k], T[k][j], acc);
  M1[i][j] = acc;
  __syncthreads();
  acc = 0.0;
  for (int k = 0; k < 32; ++k) acc = __builtin_fma(T[k][i], M1[k][j], acc);
  p.Mh[tid] = 0.5 * acc;
}

struct WArgs {
  const double* Yp;   // [m][32]  A22 Y
  const double* Y; int ldy;
  const double* Tpan; const double* Mh;    // [32][32] each
  double* VW; double* WV;                  // [m][64] each
  int m;
};

// W = Yp T - Y Mh; the two rank-64 operands [Y W], [W Y] of the trailing update.  256 threads: 8 per row (4 columns each).
__global__ __launch_bounds__(256) void sy2sb_w(WArgs p) {
  __shared__ M33 T, Mh;
  __shared__ double Ys[32][33], Ps[32][33];
  const int tid = threadIdx.x, lrow = tid >> 3, c4 = (tid & 7) * 4;
  for (int idx = tid; idx < 1024; idx += 256) {
    T[idx >> 5][idx & 31] = p.Tpan[idx];
    Mh[idx >> 5][idx & 31] = p.Mh[idx];
  }
  const int row = blockIdx.x * 32 + lrow;
  const bool ok = row < p.m;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    Ys[lrow][c4 + q] = ok ? p.Y[(size_t)row * p.ldy + c4 + q] : 0.0;
    Ps[lrow][c4 + q] = ok ? p.Yp[(size_t)row * 32 + c4 + q] : 0.0;
  }
  __syncthreads();
  if (!ok) return;
  double w[4] = {0.0, 0.0, 0.0, 0.0};
  for (int k = 0; k < 32; ++k) {
    const double a = Ps[lrow][k], b = Ys[lrow][k];
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = __builtin_fma(a, T[k][c4 + q], __builtin_fma(-b, Mh[k][c4 + q], w[q]));
  }
  d4_t wv, yv;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    wv[q] = w[q];
    yv[q] = Ys[lrow][c4 + q];
  }
  double* vw = p.VW + (size_t)row * 64;
  double* wvp = p.WV + (size_t)row * 64;
  *reinterpret_cast<d4_t*>(vw + c4) = yv;
  *reinterpret_cast<d4_t*>(vw + 32 + c4) = wv;
  *reinterpret_cast<d4_t*>(wvp + c4) = wv;
  *reinterpret_cast<d4_t*>(wvp + 32 + c4) = yv;
}

// diagonal 32 x 32 blocks of the reduced matrix -> band storage (lower part)
__global__ void sy2sb_copy_diag(const double* __restrict__ A, int lda, double* __restrict__ AB, int n) {
  const int j0 = blockIdx.x * 32;
  for (int idx = threadIdx.x; idx < 1024; idx += blockDim.x) {
    const int i = idx >> 5, j = idx & 31;
    if (i >= j && j0 + i < n) AB[(size_t)(j0 + j) * SB_LDB + (i - j)] = A[(size_t)(j0 + i) * lda + j0 + j];
  }
}

#define AV_MAX_PARTS 32
static inline int sy2sb_max_leaves(int n) { return max(16, (n + QR_ROWS - 1) / QR_ROWS); }
size_t ffgp_sy2sb_ws_doubles(int n) {
  // Rst, Tst (all leaves), Rst2, Tst2 (middle level), Vtst, small (+ Mh), Yp, Gpart, VW, WV, partial products of the k-chunks
  return (size_t)sy2sb_max_leaves(n) * 1024 * 2 + (size_t)16 * 1024 * 2 + 512 * 32 + 4096 + (size_t)n * 32 + (size_t)(n / 128 + 1) * 1024 +
         (size_t)n * 64 * 2 +
         (size_t)(AV_MAX_PARTS + 1) * n * 32 + 64 +
         (size_t)(n / 128 + 1) * n * 32;      // transposed partials of the lower-triangle form (sy2sb_av_sym)
}

// A [n, n] full symmetric (destroyed), AB [n, 64] band out, Y [n, ldy] reflector store out (zero outside the staircase),
// Tpan [n/32][1024] the panels' T factors, ws: ffgp_sy2sb_ws_doubles(n)
int ffgp_sy2sb_impl(ffgp_handle* h, double* A, int n, int lda, double* AB, double* Y, int ldy, double* Tpan, double* ws) {
  if (n % 64 != 0 || n < 64 || n > 16 * 16 * QR_ROWS) return FFGP_ERR_ARG;
  hipStream_t st = h->stream;
  const int Lmax = sy2sb_max_leaves(n);
  double* Rst = ws;
  double* Tst = Rst + (size_t)Lmax * 1024;
  double* Rst2 = Tst + (size_t)Lmax * 1024;
  double* Tst2 = Rst2 + 16 * 1024;
  double* Vtst = Tst2 + 16 * 1024;
  double* small = Vtst + 512 * 32;
  double* Yp = small + 4096;
  double* Gpart = Yp + (size_t)n * 32;
  double* VW = Gpart + (size_t)(n / 128 + 1) * 1024;
  double* Mh = small + 3072;   // (small holds Xt | S | U^-1 in its first 2080 doubles)
  double* WV = VW + (size_t)n * 64;
  double* Ppart = WV + (size_t)n * 64;
  double* PTpart = Ppart + (size_t)(AV_MAX_PARTS + 1) * n * 32 + 64;
  // lower-triangle form: the trailing matrix is kept in its lower triangle only (the forms that read whole rows keep the full update)
  const bool lower = h->sb_lower && n >= h->sb_lower_min_n && !h->sb_av_gemm && !h->sb_lookahead;
  FFGP_HIP(hipMemsetAsync(AB, 0, (size_t)n * SB_LDB * sizeof(double), st));
  FFGP_HIP(hipMemsetAsync(Y, 0, (size_t)n * ldy * sizeof(double), st));
  const int npan = n / 32 - 1;
  // the panel factorisation (three small launches on <= 16 CUs) of panel p + 1 only needs panel p's update of ITS 32 columns: that
  // strip is updated first, then the QR chain runs on the side stream under the rest of the update
  const bool la = h->sb_lookahead != 0;
  hipStream_t side = h->aux;
  if (la && !h->sb_ev[0])
    for (int i = 0; i < 4; ++i) FFGP_HIP(hipEventCreateWithFlags(&h->sb_ev[i], hipEventDisableTiming));
  auto panel_qr = [&](int p, hipStream_t q) -> int {
    const int j0 = p * 32, r0 = j0 + 32, m = n - r0;
    const int L = (m + QR_ROWS - 1) / QR_ROWS;
    double* Ap = A + (size_t)r0 * lda + j0;
    double* Ypan = Y + (size_t)r0 * ldy + j0;
    LeafArgs la_;
    la_.A = Ap; la_.lda = lda; la_.m = m; la_.Rst = Rst; la_.Tst = Tst;
    if (h->sb_qr4) hipLaunchKernelGGL(sy2sb_leaf_qr4, dim3(L), dim3(QR4_THREADS), 0, q, la_);
    else hipLaunchKernelGGL(sy2sb_leaf_qr, dim3(L), dim3(QR_THREADS), 0, q, la_);
    // more than 16 leaves (m > 8192): a middle level -- the same kernel factors the leaves' R factors, 16 (= 512 rows) at a time
    const int three = (L > 16) ? 1 : 0;
    const int L2 = (L + 15) / 16;
    if (three) {
      LeafArgs lm;
      lm.A = Rst; lm.lda = 32; lm.m = L * 32; lm.Rst = Rst2; lm.Tst = Tst2;
      if (h->sb_qr4) hipLaunchKernelGGL(sy2sb_leaf_qr4, dim3(L2), dim3(QR4_THREADS), 0, q, lm);
      else hipLaunchKernelGGL(sy2sb_leaf_qr, dim3(L2), dim3(QR_THREADS), 0, q, lm);
    }
    TopArgs ta;
    ta.A = Ap; ta.lda = lda; ta.m = m; ta.L = three ? L2 : L; ta.Rst = three ? Rst2 : Rst; ta.Tst = Tst; ta.Vtst = Vtst; ta.small = small;
    ta.Vmid0 = Rst; ta.Tmid0 = Tst2; ta.three = three;
    ta.Tpan = Tpan + (size_t)p * 1024; ta.Y = Ypan; ta.ldy = ldy; ta.AB = AB + (size_t)j0 * SB_LDB; ta.use_tree = (L > 1) ? 1 : 0;
    hipLaunchKernelGGL(sy2sb_top, dim3(1), dim3(QR_THREADS), 0, q, ta);
    FormYArgs fa;
    fa.A = Ap; fa.lda = lda; fa.m = m; fa.L = L; fa.Tst = Tst; fa.Vtst = Vtst; fa.small = small; fa.Y = Ypan; fa.ldy = ldy;
    fa.Vmid = Rst; fa.Tmid = Tst2; fa.three = three;
    fa.use_tree = ta.use_tree;
    hipLaunchKernelGGL(sy2sb_form_y, dim3(2 * L), dim3(256), 0, q, fa);
    return hipGetLastError() == hipSuccess ? FFGP_OK : FFGP_ERR_HIP;
  };
  if (npan > 0) FFGP_CHECK(panel_qr(0, st));
  for (int p = 0; p < npan; ++p) {
    const int j0 = p * 32, r0 = j0 + 32, m = n - r0;
    double* Ypan = Y + (size_t)r0 * ldy + j0;
    double* Tp = Tpan + (size_t)p * 1024;
    double* A22 = A + (size_t)r0 * lda + r0;
    // Yp = A22 Y  (m x m times m x 32; A22 K-major, Y stored k x n), cut along k into `parts` chunks that run as ONE batched launch
    // (enough workgroups to fill the chip whatever m is); the chunks are summed in fixed order by sy2sb_red, which also leaves
    // the pieces of G = Y^T Yp
    int parts = min(lower ? AV_MAX_PARTS : 16, max(1, (h->sb_av_gemm ? 1024 : (lower ? h->sb_sym_wg : 2048)) / ((m + 63) / 64)));
    int kc = ((m + parts - 1) / parts + 31) & ~31;
    parts = m / kc;                       // full chunks; a shorter tail chunk runs as its own launch
    const int tail = m - parts * kc;
    const long sP = (long)m * 32;
    if (h->sb_av_gemm) {                  // (the general GEMM, kept for comparison: option "sb_av_gemm")
      if (parts > 0)
        FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, A22, lda, Ypan, ldy, Ppart, 32, m, 32, kc, 1.0, 0.0, 0, ALIAS_NONE, parts,
                                    (long)kc, (long)kc * ldy, sP));
      if (tail > 0)
        FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, A22 + (size_t)parts * kc, lda, Ypan + (size_t)parts * kc * ldy, ldy,
                                    Ppart + (size_t)parts * sP, 32, m, 32, tail, 1.0, 0.0));
    } else if (lower) {
      AvSymArgs av;
      av.A = A22; av.lda = lda; av.Y = Ypan; av.ldy = ldy; av.P = Ppart; av.sP = sP; av.PT = PTpart; av.sPT = sP; av.m = m; av.kc = kc;
      hipLaunchKernelGGL(sy2sb_av_sym, dim3((m + 127) / 128, parts + (tail > 0 ? 1 : 0)), dim3(256), 0, st, av);
    } else {
      AvArgs av;
      av.A = A22; av.lda = lda; av.Y = Ypan; av.ldy = ldy; av.P = Ppart; av.sP = sP; av.m = m; av.kc = kc;
      hipLaunchKernelGGL(sy2sb_av, dim3((m + 127) / 128, parts + (tail > 0 ? 1 : 0)), dim3(256), 0, st, av);
    }
    RedArgs ra;
    ra.P = Ppart; ra.sP = sP; ra.parts = parts + (tail > 0 ? 1 : 0); ra.Y = Ypan; ra.ldy = ldy; ra.Yp = Yp; ra.Gpart = Gpart; ra.m = m;
    ra.PT = lower ? PTpart : nullptr; ra.sPT = sP; ra.kc = kc;
    const int nred = (m + 127) / 128;
    hipLaunchKernelGGL(sy2sb_red, dim3(nred), dim3(1024), 0, st, ra);
    MArgs ma;
    ma.Tpan = Tp; ma.Gpart = Gpart; ma.ngp = nred; ma.Mh = Mh;
    hipLaunchKernelGGL(sy2sb_m, dim3(1), dim3(1024), 0, st, ma);
    WArgs wa;
    wa.Yp = Yp; wa.Y = Ypan; wa.ldy = ldy; wa.Tpan = Tp; wa.Mh = Mh; wa.VW = VW; wa.WV = WV; wa.m = m;
    hipLaunchKernelGGL(sy2sb_w, dim3((m + 31) / 32), dim3(256), 0, st, wa);
    if (hipGetLastError() != hipSuccess) return FFGP_ERR_HIP;
    // A22 -= [Y W] [W Y]^T
    const bool next = (p + 1 < npan);
    if (la && next && m > 64) {
      hipEvent_t ea = h->sb_ev[(p & 1) * 2], eq = h->sb_ev[(p & 1) * 2 + 1];
      FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_KMAJOR, TILES_FULL, 0, VW, 64, WV, 64, A22, lda, m, 32, 64, -1.0, 1.0));   // the next panel's columns
      FFGP_HIP(hipEventRecord(ea, st));
      FFGP_HIP(hipStreamWaitEvent(side, ea, 0));
      FFGP_CHECK(panel_qr(p + 1, side));
      FFGP_HIP(hipEventRecord(eq, side));
      FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_KMAJOR, TILES_FULL, 0, VW, 64, WV + 32 * 64, 64, A22 + 32, lda, m, m - 32, 64, -1.0, 1.0));
      FFGP_HIP(hipStreamWaitEvent(st, eq, 0));
    } else {
      FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_KMAJOR, lower ? TILES_LOWER : TILES_FULL, 0, VW, 64, WV, 64, A22, lda, m, m, 64, -1.0, 1.0));
      if (next) FFGP_CHECK(panel_qr(p + 1, st));
    }
  }
  hipLaunchKernelGGL(sy2sb_copy_diag, dim3(n / 32), dim3(256), 0, st, A, lda, AB, n);
  return hipGetLastError() == hipSuccess ? FFGP_OK : FFGP_ERR_HIP;
}
