// Blocked right-looking lower Cholesky for gfx950 (fp64).
//
//   outer step (width nb_outer, default 512):
//     panel: for each 128-wide sub-block
//        potrf_diag128   one workgroup, the 128x128 diagonal block LDS-resident (padded [128][130] image):
//                        left-looking over 16-column blocks -- MFMA updates, a register/readlane 16x16 factor in
//                        one wave, substitution TRSM with one lane per row -- then the block's INVERSE is formed
//                        in place (16x16 inverses + MFMA products) and written to the handle's Dinv store;
//        TRSM            A21 <- A21 * inv(L11)^T      = one NT GEMM on the matrix cores (in place);
//        panel update    A22p -= A21 * A21p^T         = one NT GEMM (lower-trapezoid tiles);
//     trailing update    A22 -= P * P^T  (K = nb_outer) = the SYRK instantiation of the GEMM kernel -- the
//                        kernel that carries ~95 % of the N^3/3 flops and the one the roofline is quoted on.
//
// The strictly-upper triangle of A is never read or written.  A failing pivot is reported as its 1-based
// global index (first failure wins) and the factorisation continues with a unit pivot so that no NaN/Inf
// propagates into later kernels' control flow.
#include "ffgp_internal.h"

#define NB FFGP_NB
// LDS image of the 128x128 diagonal block: only the 36 lower 16x16 blocks, each [16][17] doubles (the pad makes
// the MFMA operand reads bank-conflict-free).  78 KiB instead of 130 KiB for the dense image: the kernel must
// fit beside ONE resident GEMM workgroup (72 KiB of the CU's 160 KiB), otherwise the look-ahead panel factor
// would never be scheduled while the trailing update occupies the chip.
#define BLD 17
#define BLKSZ (16 * BLD)
#define NBLK_LOWER 36
#define DIAG_LDS_DOUBLES (NBLK_LOWER * BLKSZ + 128)
#define DIAG_LDS_BYTES (DIAG_LDS_DOUBLES * 8)
// 8 waves: wave 0 runs the serial 16x16 factor chain, the other seven do the MFMA work in its shadow
#define DIAG_THREADS 512
#define DIAG_WAVES (DIAG_THREADS / 64)

__device__ __forceinline__ int blk_off(int bi, int bj) { return (bi * (bi + 1) / 2 + bj) * BLKSZ; }

__device__ __forceinline__ double readlane_d(double x, int l) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

// value of x held by lane (16*(lane>>4) + J): DPP row broadcast inside each row of 16 lanes
template <int J>
__device__ __forceinline__ double row_bcast_d(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x150 + J, 0xf, 0xf, true);   // every lane receives data: no `old` value to seed
  hi = __builtin_amdgcn_mov_dpp(hi, 0x150 + J, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// value of x held by lane `src` (per-lane source): ds_bpermute
__device__ __forceinline__ double bperm_d(double x, int src) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_ds_bpermute(src << 2, lo);
  hi = __builtin_amdgcn_ds_bpermute(src << 2, hi);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double rsqrt_nr(double d) {
  double y = __builtin_amdgcn_rsq(d);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const double e = __builtin_fma(-d * y, y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
  }
  return y;
}

__device__ __forceinline__ double rcp_nr(double d) {
  double y = __builtin_amdgcn_rcp(d);
#pragma unroll
  for (int it = 0; it < 2; ++it) {  // v_rcp_f64 is good to ~2^-26; two Newton steps reach fp64 rounding level
    const double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
  }
  return y;
}

// 16x16 MFMA tile product helper: acc += Arows(16 x 16, K-major at pa[row*lda_ + k]) * B
//   KB = true : B given K-major  (B^T stored: element (n,k) at pb[n*ldb_ + k])
//   KB = false: B given N-major  (element (k,n) at pb[k*ldb_ + n])
template <bool KB>
__device__ __forceinline__ void mma16(d4_t& acc, const double* pa, int lda_, const double* pb, int ldb_, int lane) {
#pragma unroll
  for (int kq = 0; kq < 4; ++kq) {
    const int k = kq * 4 + (lane >> 4);
    const double a = pa[(lane & 15) * lda_ + k];
    const double b = KB ? pb[(lane & 15) * ldb_ + k] : pb[k * ldb_ + (lane & 15)];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
}

// one step of the in-register 16x16 Cholesky: the block is symmetric-full, lane (g = lane>>4, c = lane&15) holds
// rows g+4r (r = 0..3) of column c.  Pivot J: rank-1 downdate of the whole block with column J / d.
// The pivot-to-pivot dependency chain is  readlane -> rcp -> 3 fma -> fma  (fp64 VALU ops have ~24-cycle dependent
// latency, so every op removed from the chain is ~10 % of the block's factor time):
//   * 1/d = y0 (1 + e + e^2), e = 1 - d y0, with y0 = v_rcp_f64(d) (~2^-26): t = u + u*(e + e^2), u = A[J][c]*y0 --
//     three dependent fmas after the rcp instead of two Newton steps plus a multiply;
//   * the pivot is not sanitised on the chain: a non-positive pivot is flagged on the side and the block's
//     results are then garbage (as LAPACK's are), but no control flow depends on them;
//   * the lanes of column J park their unscaled column and pivot; 1/sqrt(d) scaling happens after the 16 steps;
//   * the same eliminations are applied to an identity block W (off the chain, in its issue gaps), so that
//     inv(L_jj) = diag(1/sqrt(d)) * W comes out with the factor and the rows below are solved on the matrix cores.
template <int J>
__device__ __forceinline__ void chol16_step(double (&v)[4], double (&w)[4], double (&out)[4], double& dmine, int lane,
                                            int& bad) {
  constexpr int PL = 16 * (J & 3) + J, PR = J >> 2;
  const double d = readlane_d(v[PR], PL);
  bad = (!(d > 0.0) && bad == 0) ? J + 1 : bad;
  const double y0 = __builtin_amdgcn_rcp(d);
  const double rowj = bperm_d(v[PR], 16 * (J & 3) + (lane & 15));  // A[J][c]
  const double wrow = bperm_d(w[PR], 16 * (J & 3) + (lane & 15));  // W[J][c]
  double colj[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) colj[r] = row_bcast_d<J>(v[r]);      // A[g+4r][J]
  const bool mine = (lane & 15) == J;
  dmine = mine ? d : dmine;
#pragma unroll
  for (int r = 0; r < 4; ++r) out[r] = mine ? v[r] : out[r];
  const double e = __builtin_fma(-d, y0, 1.0);
  const double f = __builtin_fma(e, e, e);
  const double u = rowj * y0;
  const double t = __builtin_fma(u, f, u);          // A[J][c] / d
  const double uw = wrow * y0;
  const double tw = __builtin_fma(uw, f, uw);       // W[J][c] / d
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = __builtin_fma(-colj[r], t, v[r]);
  // W rows above the pivot see multipliers that are rounding residue of already-eliminated entries (~1e-16 relative:
  // harmless); only the pivot row itself must be left alone
  colj[PR] = ((lane >> 4) == (J & 3)) ? 0.0 : colj[PR];
#pragma unroll
  for (int r = 0; r < 4; ++r) w[r] = __builtin_fma(-colj[r], tw, w[r]);
}

// one level of the in-place blocked inversion by recursive doubling: pairs of inverted S-block-wide diagonal
// blocks (S in units of 16) are merged,  X21 = -X22 * (L21 * X11).  Four (pair, block column) work items per
// level = one per wave; a wave keeps its column of T = L21*X11 in registers across the barrier that protects
// L21 from being overwritten while other waves still read it.
template <int S_>
__device__ __forceinline__ void inv_merge_level(double* S, int wave, int lane, double* __restrict__ Dinv, bool wr) {
  const bool act = wave < 4;   // four work items per level; any further waves only take part in the barriers
  const int pair = wave / S_, jl = wave % S_;
  const int b0 = pair * 2 * S_;
  const int j = b0 + jl;
  d4_t T[S_];
  if (act) {
#pragma unroll
    for (int ii = 0; ii < S_; ++ii) {
      const int i = b0 + S_ + ii;
      d4_t acc = {0.0, 0.0, 0.0, 0.0};
      for (int k = j; k < b0 + S_; ++k) mma16<false>(acc, S + blk_off(i, k), BLD, S + blk_off(k, j), BLD, lane);
      T[ii] = acc;
    }
  }
  __syncthreads();
  if (act) {
#pragma unroll
    for (int ii = 0; ii < S_; ++ii) {
      double* dst = S + blk_off(b0 + S_ + ii, j);
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[((lane >> 4) + 4 * r) * BLD + (lane & 15)] = T[ii][r];
    }
    d4_t R[S_];
#pragma unroll
    for (int ii = 0; ii < S_; ++ii) {
      const int i = b0 + S_ + ii;
      d4_t acc = {0.0, 0.0, 0.0, 0.0};
      for (int k = b0 + S_; k <= i; ++k) mma16<false>(acc, S + blk_off(i, k), BLD, S + blk_off(k, j), BLD, lane);
      R[ii] = acc;
    }
#pragma unroll
    for (int ii = 0; ii < S_; ++ii) {
      double* dst = S + blk_off(b0 + S_ + ii, j);
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[((lane >> 4) + 4 * r) * BLD + (lane & 15)] = -R[ii][r];
      // the block is final: it also goes straight to the Dinv store (no separate write-out pass)
      if (wr) {
        double* g = Dinv + (size_t)((b0 + S_ + ii) * 16 + (lane >> 4)) * NB + j * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) g[(size_t)4 * r * NB] = -R[ii][r];
      }
    }
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------------------
// potrf_diag128: factor one diagonal block (nb <= 128 valid rows/cols, identity-padded) and invert it.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DIAG_THREADS) void ffgp_potrf_diag128(double* __restrict__ A, int lda, int nb,
                                                          double* __restrict__ Dinv, int* info, int row_base,
                                                          int do_factor, int dbg, int prio) {
  // dbg: timing-only ablation mask (results are wrong when non-zero): 1 skip (b), 2 skip (c), 4 skip (a),
  //      8 skip phase 3, 16 skip phase 4, 32 skip phase 5, 64 skip phase 2, 128 skip phase 0
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* S = lds;                      // 36 lower blocks [16][17]
  double* rd = lds + NBLK_LOWER * BLKSZ;  // [128] reciprocals of the diagonal of L
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (prio) __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win issue arbitration against co-resident GEMM waves

  // ---- phase 0: load the lower blocks; diagonal blocks are completed symmetrically (mirror of the lower part),
  //      rows/cols beyond nb are identity
  if (!(dbg & 128)) {
    // 32 unconditional 16-byte loads per thread, all in flight before the first LDS store (rows are clamped into
    // the valid block; entries above the diagonal are fetched but never used)
    constexpr int NLOAD = 8192 / DIAG_THREADS;
    d2_t lv[NLOAD];
    const bool vec = !(lda & 1) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) {
      const int idx = tid + DIAG_THREADS * it;
      const int r = min(idx >> 6, nb - 1), c = min((idx & 63) * 2, (nb - 1) & ~1);
      const double* src = A + (size_t)r * lda + c;
      if (vec) {
        lv[it] = *reinterpret_cast<const d2_t*>(src);
      } else {
        lv[it].x = src[0];
        lv[it].y = (c + 1 < nb) ? src[1] : 0.0;
      }
    }
#pragma unroll
    for (int it = 0; it < NLOAD; ++it) {
      const int idx = tid + DIAG_THREADS * it;
      const int r = idx >> 6, c = (idx & 63) * 2;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int cc = c + e;
        if (cc <= r) {
          double v = (r == cc) ? 1.0 : 0.0;
          if (r < nb) v = e ? lv[it].y : lv[it].x;
          S[blk_off(r >> 4, cc >> 4) + (r & 15) * BLD + (cc & 15)] = v;
        }
      }
    }
    __syncthreads();
    for (int idx = tid; idx < 8 * 256; idx += DIAG_THREADS) {
      const int jj = idx >> 8, i = (idx >> 4) & 15, c = idx & 15;
      double* Dj = S + blk_off(jj, jj);
      if (c > i) Dj[i * BLD + c] = Dj[c * BLD + i];
    }
  }
  __syncthreads();

  if (do_factor) {
    // ---- phase 1: left-looking factorisation over 16-column blocks, with one block column of look-ahead:
    //      while wave 0 factors the 16x16 diagonal block jj in registers, waves 1-3 already apply blocks p < jj to
    //      block column jj+1, so that step jj+1 only has the single product with block column jj left on the chain
    const bool wr = !(dbg & 64);
    for (int jj = 0; jj < 8; ++jj) {
      d4_t T = {0.0, 0.0, 0.0, 0.0};   // this wave's block (j = wave - 1) of L[jj][:] * X, see the shadow work in (b)
      // (a) the one missing term: S[i][jj] -= S[i][jj-1] * S[jj][jj-1]^T   for block rows i = jj..7 (MFMA)
      if (jj > 0 && !(dbg & 4)) {
        for (int i = jj + wave; i < 8; i += DIAG_WAVES) {
          d4_t acc = {0.0, 0.0, 0.0, 0.0};
          mma16<true>(acc, S + blk_off(i, jj - 1), BLD, S + blk_off(jj, jj - 1), BLD, lane);
          double* dst = S + blk_off(i, jj);
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[((lane >> 4) + 4 * r) * BLD + (lane & 15)] -= acc[r];
        }
      }
      __syncthreads();
      // (b) wave 0: in-register 16x16 Cholesky + inverse (4 entries per lane, DPP / bpermute broadcasts).  L_jj goes
      //     straight to global memory; its LDS slot receives inv(L_jj), which is all that (c) and the block
      //     inversion need from it.
      if (wave == 0 && !(dbg & 1)) {
        double* Dj = S + blk_off(jj, jj);
        const int g = lane >> 4, c = lane & 15;
        double v[4], w[4], out[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = Dj[(g + 4 * r) * BLD + c];
          w[r] = (g + 4 * r == c) ? 1.0 : 0.0;
          out[r] = 0.0;
        }
        int bad = 0;
        double dmine = 1.0;
        chol16_step<0>(v, w, out, dmine, lane, bad);
        chol16_step<1>(v, w, out, dmine, lane, bad);
        chol16_step<2>(v, w, out, dmine, lane, bad);
        chol16_step<3>(v, w, out, dmine, lane, bad);
        chol16_step<4>(v, w, out, dmine, lane, bad);
        chol16_step<5>(v, w, out, dmine, lane, bad);
        chol16_step<6>(v, w, out, dmine, lane, bad);
        chol16_step<7>(v, w, out, dmine, lane, bad);
        chol16_step<8>(v, w, out, dmine, lane, bad);
        chol16_step<9>(v, w, out, dmine, lane, bad);
        chol16_step<10>(v, w, out, dmine, lane, bad);
        chol16_step<11>(v, w, out, dmine, lane, bad);
        chol16_step<12>(v, w, out, dmine, lane, bad);
        chol16_step<13>(v, w, out, dmine, lane, bad);
        chol16_step<14>(v, w, out, dmine, lane, bad);
        chol16_step<15>(v, w, out, dmine, lane, bad);
        const double rs = rsqrt_nr(dmine);  // 1/sqrt(pivot) of this lane's column
        if (g == 0) rd[jj * 16 + c] = rs;
        // L[i][c] = out * rs (rows >= c) -> global
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = g + 4 * r;
          const int gr = jj * 16 + i, gc = jj * 16 + c;
          if (i >= c && gr < nb && wr) A[(size_t)gr * lda + gc] = out[r] * rs;
        }
        // inv(L_jj)[i][c] = rs_i * W[i][c]; rs_i comes back through the LDS slot just written
        __builtin_amdgcn_s_waitcnt(0);       // rd[] visible to the whole wave (same-wave LDS write -> read)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = g + 4 * r;
          const double x = (i >= c) ? w[r] * rd[jj * 16 + i] : 0.0;
          Dj[i * BLD + c] = x;
          if (i >= c && !(dbg & 32)) Dinv[(size_t)(jj * 16 + i) * NB + jj * 16 + c] = x;
        }
        if (bad && lane == 0 && (jj * 16 + bad) <= nb) atomicCAS(info, 0, row_base + jj * 16 + bad);
      } else if (wave != 0 && jj > 0) {
        // in the shadow of the 16x16 factor (waves 1..3):
        // look-ahead: S[i][jj+1] -= sum_{p<jj} S[i][p] * S[jj+1][p]^T  for block rows i = jj+1..7
        if (jj < 7 && !(dbg & 4)) {
          for (int i = jj + 1 + (wave - 1); i < 8; i += DIAG_WAVES - 1) {
            d4_t acc = {0.0, 0.0, 0.0, 0.0};
            for (int p = 0; p < jj; ++p) mma16<true>(acc, S + blk_off(i, p), BLD, S + blk_off(jj + 1, p), BLD, lane);
            double* dst = S + blk_off(i, jj + 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[((lane >> 4) + 4 * r) * BLD + (lane & 15)] -= acc[r];
          }
        }
        // incremental inverse, row block jj: T[jj][j] = sum_{k=j}^{jj-1} L[jj][k] * X[k][j]  (X = inv(L), rows < jj
        // already sit in place of L's); the products with inv(L_jj) follow in (c).  T stays in registers.
        if (!(dbg & 16) && wave - 1 < jj) {
          const int j = wave - 1;
          for (int k = j; k < jj; ++k) mma16<false>(T, S + blk_off(jj, k), BLD, S + blk_off(k, j), BLD, lane);
        }
      }
      __syncthreads();
      // (c) rows below: X = B * inv(L_jj)^T on the matrix cores (4 MFMAs per 16-row block); the finished block of L
      //     goes to LDS (later steps read it) and straight to global memory
      if (!(dbg & 2)) {
        for (int i = jj + 1 + wave; i < 8; i += DIAG_WAVES) {
          d4_t acc = {0.0, 0.0, 0.0, 0.0};
          double* Bij = S + blk_off(i, jj);
          mma16<true>(acc, Bij, BLD, S + blk_off(jj, jj), BLD, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) Bij[((lane >> 4) + 4 * r) * BLD + (lane & 15)] = acc[r];
          if (wr) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int gr = i * 16 + (lane >> 4) + 4 * r;
              if (gr < nb) A[(size_t)gr * lda + jj * 16 + (lane & 15)] = acc[r];
            }
          }
        }
      }
      // (c') inverse, row block jj: X[jj][j] = -inv(L_jj) * T[jj][j].  The accumulator layout of T (lane group g,
      //      register r <-> row g + 4r) IS the MFMA B-operand layout of k-step r, so T never leaves its registers.
      if (wave != 0 && wave - 1 < jj && !(dbg & 16)) {
        const double* Wj = S + blk_off(jj, jj);
        const int j = wave - 1;
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kq = 0; kq < 4; ++kq)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Wj[(lane & 15) * BLD + kq * 4 + (lane >> 4)], T[kq], acc, 0, 0, 0);
        double* dst = S + blk_off(jj, j);
        double* g = Dinv + (size_t)(jj * 16 + (lane >> 4)) * NB + j * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dst[((lane >> 4) + 4 * r) * BLD + (lane & 15)] = -acc[r];
          if (!(dbg & 32)) g[(size_t)4 * r * NB] = -acc[r];
        }
      }
      __syncthreads();
    }
  } else {
    // inverse-only entry (Dinv refresh for a factor produced elsewhere): reciprocals of the diagonal
    if (tid < NB) rd[tid] = 1.0 / S[blk_off(tid >> 4, tid >> 4) + (tid & 15) * BLD + (tid & 15)];
  }
  __syncthreads();

  // ---- phase 3 (inverse-only entry; the factor entry already left inv(L_jj) in the diagonal slots):
  //      inverses of the eight 16x16 diagonal blocks, in place; 16 lanes per block (one per column)
  if (!do_factor && wave < 2 && !(dbg & 8)) {
    const int jj = wave * 4 + (lane >> 4), c = lane & 15;
    double* Lj = S + blk_off(jj, jj);
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (i == c) ? 1.0 : 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const double xi = x[i] * rd[jj * 16 + i];
      x[i] = xi;
#pragma unroll
      for (int k = i + 1; k < 16; ++k) x[k] = __builtin_fma(-xi, Lj[k * BLD + i], x[k]);
    }
    // every lane of the wave has finished reading L_jj (same instruction stream) before the block is overwritten
#pragma unroll
    for (int i = 0; i < 16; ++i) Lj[i * BLD + c] = x[i];
    if (!(dbg & 32)) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (i >= c) Dinv[(size_t)(jj * 16 + i) * NB + jj * 16 + c] = x[i];
    }
  }
  __syncthreads();

  // ---- phase 4: in-place blocked inversion by recursive doubling (16 -> 32 -> 64 -> 128), 2 barriers per level;
  //      every block is written to the Dinv store the moment it is final (the strictly-upper part of the store is
  //      zero from allocation)
  if (!do_factor && !(dbg & 16)) {   // (the factor entry built the inverse row block by row block, see (c'))
    const bool wi = !(dbg & 32);
    inv_merge_level<1>(S, wave, lane, Dinv, wi);   // (4 work items per level: waves 4..7 only keep the barriers)
    inv_merge_level<2>(S, wave, lane, Dinv, wi);
    inv_merge_level<4>(S, wave, lane, Dinv, wi);
  }
}

// ------------------------------------------------------------------------------------------------------------
// potrf_diag128_v2: the same factor + inverse as a wave-specialised PIPELINE instead of barrier-separated phases.
//
// The old kernel spends 8 x (3 workgroup barriers + a ~1.7 us in-register 16x16 factor + two short MFMA phases); every wave
// waits for every other wave three times per 16 columns.  Here wave 0 never meets a barrier after the load phase:
//
//   wave 0   F(jj): in-register 16x16 Cholesky + inverse of the diagonal block (16 pivots)
//            G(jj): Y = inv(L_jj) S[jj+1][jj]^T on the matrix cores (4 MFMAs); L[jj+1][jj] = Y^T goes to LDS / global;
//                   D = S[jj+1][jj+1] - Y^T Y (4 MFMAs with a = b = the Y registers: the accumulator layout is both the
//                   A layout of Y^T and the B layout of Y) -> the next diagonal block, already in the factor's layout
//            ... F(jj+1) ...                       serial chain: 8 x (F + G), nothing else
//   helpers  iteration jj, triggered by wave 0's flags in LDS (seqF: inv(L_jj) is in LDS; seqX: L[jj+1][jj] is):
//            A1 TRSM of the block rows jj+2.. of column jj          A2 row block jj of the inverse (from T, see B3)
//            B1 column jj+1 receives block column jj   B2 column jj+2 receives block columns 0..jj (left-looking,
//            one block column of look-ahead -- so G(jj+1) finds S[jj+2][jj+1], S[jj+2][jj+2] complete)  -> doneU
//            B3 T_j = sum_k L[jj+1][k] X[k][j] for the inverse's next row block (kept in registers)
//   Helpers synchronise among themselves with a counter barrier in LDS (two per iteration); wave 0 waits for doneU of
//   iteration jj-1 before G(jj) -- by then it has spent a whole F on its own, so it normally does not wait at all.
//
// The 16x16 factor itself is restructured so that no LDS-latency operation sits on the pivot-to-pivot chain: every lane
// keeps the CURRENT pivot row for its column (rowA, rowW); the next pivot row is fetched (ds_bpermute) one step ahead,
// before this step's update, and patched locally with this step's rank-1 term; pivot and multiplier broadcasts are DPP
// row shares.  Finished columns are parked in place (their multiplier is masked to 0), so no select instructions
// capture them.  Every polling loop is bounded: on a timeout the kernel raises `abort`, every wave leaves, and the host
// sees FFGP_DIAG_WATCHDOG in the status word instead of a hung queue.
// ------------------------------------------------------------------------------------------------------------
#define FFGP_DIAG_WATCHDOG 0x7ffffff0

// tools/diag_trace.py builds a second library with -DFFGP_DIAG_TRACE: wave 0 and helper 0 stamp the cycle counter at their
// phase boundaries (never compiled into libffgp.so)
#ifdef FFGP_DIAG_TRACE
__device__ unsigned long long* ffgp_diag_trace_buf = nullptr;
__device__ int ffgp_diag_trace_row = -1;          // -1: every launch stamps (the last one stays); else only the launch at this row
extern "C" int ffgp_debug_set_diag_trace(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(ffgp_diag_trace_buf), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
extern "C" int ffgp_debug_set_diag_trace_row(int row) {
  return hipMemcpyToSymbol(HIP_SYMBOL(ffgp_diag_trace_row), &row, sizeof(row)) == hipSuccess ? 0 : -1;
}
#define D2_TRACE(slot)                                                                          \
  do {                                                                                          \
    if (lane == 0 && ffgp_diag_trace_buf && (ffgp_diag_trace_row < 0 || ffgp_diag_trace_row == row_base))  \
      ffgp_diag_trace_buf[(slot)] = wall_clock64();                                             \
  } while (0)
#else
#define D2_TRACE(slot)
#endif

struct D2Flags {      // ints in LDS, behind the block image
  int seqF, seqX, doneU, sb, abort, pad[3];
};

#define D2_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define D2_COMPILER_FENCE() asm volatile("" ::: "memory")

// The flags live in LDS and are touched with explicit DS instructions.  A `volatile` access through the generic pointer
// compiles to flat_load / flat_store with `s_waitcnt vmcnt(0)`: every poll would first wait for all of the wave's
// outstanding GLOBAL stores (the L / inverse blocks it has just written -- a memory round trip per hand-off).
__device__ __forceinline__ int d2_ld(const volatile int* p) {
  int v;
  const unsigned off = (unsigned)(uintptr_t)p;          // low half of a generic LDS address = the LDS offset
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(off) : "memory");
  return v;
}
__device__ __forceinline__ void d2_st(volatile int* p, int v) {
  const unsigned off = (unsigned)(uintptr_t)p;
  asm volatile("ds_write_b32 %0, %1" : : "v"(off), "v"(v) : "memory");
}

__device__ __forceinline__ bool d2_wait_ge(volatile int* p, int target, volatile int* abort_flag, int* info) {
  int spins = 0;
  while (d2_ld(p) < target) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 63) == 0 && (spins > (1 << 21) || d2_ld(abort_flag))) {   // ~1 s of polling: something upstream died
      if (!d2_ld(abort_flag)) atomicExch(info, FFGP_DIAG_WATCHDOG);
      d2_st(abort_flag, 1);
      return false;
    }
  }
  D2_COMPILER_FENCE();
  return true;
}

// one pivot of the pipelined in-register factor (see the header comment).  lane (g = lane>>4, c = lane&15) holds rows
// g+4r of column c of the symmetric block (v) and of the eliminated identity (w); rowA / rowW = current row J of both.
template <int J>
__device__ __forceinline__ void f16_step(double (&v)[4], double (&w)[4], double& rowA, double& rowW, int c, int g) {
  double preA = 0.0, preW = 0.0;
  if constexpr (J < 15) {   // row J+1 as it stands BEFORE this pivot's update; patched below
    constexpr int PR1 = (J + 1) >> 2, G1 = (J + 1) & 3;
    preA = bperm_d(v[PR1], 16 * G1 + c);
    preW = bperm_d(w[PR1], 16 * G1 + c);
  }
  const double d = row_bcast_d<J>(rowA);                     // A[J][J]  (checked for positivity after the 16 steps)
  const double y0 = __builtin_amdgcn_rcp(d);
  const double e = __builtin_fma(-d, y0, 1.0);
  const double f = __builtin_fma(e, e, e);                   // 1/d = y0 (1 + e + e^2)
  const double u = rowA * y0;
  const double t = __builtin_fma(u, f, u);                   // A[J][c] / d
  const double uw = rowW * y0;
  const double tw = __builtin_fma(uw, f, uw);                // W[J][c] / d
  // registers whose four rows (g + 4r, g = 0..3) are all <= J hold finished rows: neither block is updated there
  constexpr int RMIN = (J + 1) >> 2;
  double colj[4];
#pragma unroll
  for (int r = RMIN; r < 4; ++r) colj[r] = row_bcast_d<J>(v[r]);   // A[g+4r][J]
  if constexpr (J < 15) {
    const double s = row_bcast_d<(J + 1) & 15>(rowA);        // A[J][J+1] = A[J+1][J]
    rowA = __builtin_fma(-s, t, preA);
    rowW = __builtin_fma(-s, tw, preW);
  }
  const double tm = (c > J) ? t : 0.0;                       // columns <= J are parked: they keep the unscaled L column
#pragma unroll
  for (int r = RMIN; r < 4; ++r) v[r] = __builtin_fma(-colj[r], tm, v[r]);
  constexpr int PR = J >> 2;
  if constexpr (PR >= RMIN) colj[PR] = (g == (J & 3)) ? 0.0 : colj[PR];   // the pivot row of W stays
#pragma unroll
  for (int r = RMIN; r < 4; ++r) w[r] = __builtin_fma(-colj[r], tw, w[r]);
}

template <int NW>
__global__ __launch_bounds__(NW * 64, NW / 2) void ffgp_potrf_diag128_v2(double* __restrict__ A, int lda, int nb, double* __restrict__ Dinv,
                                                                 int* info, int row_base, int prio) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* S = lds;
  volatile D2Flags* fl = reinterpret_cast<volatile D2Flags*>(lds + NBLK_LOWER * BLKSZ);
  constexpr int NT = NW * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave == 0) D2_TRACE(0);
  if (tid == 0) {
    d2_st(&fl->seqF, 0); d2_st(&fl->seqX, 0); d2_st(&fl->doneU, 0); d2_st(&fl->sb, 0); d2_st(&fl->abort, 0);
  }
  // ---- load phase (as in the barrier version): lower blocks, diagonal blocks completed symmetrically, identity padding
  {
    // two rounds of 8 loads per thread (all of a round in flight before its LDS stores): the kernel is held to 128 VGPRs
    // so that it fits on a CU BESIDE a resident trailing-update workgroup (the barrier version needs 256 -- a whole CU)
    constexpr int NLOAD = 8192 / NT, CH = NLOAD / 2;
    const bool vec = !(lda & 1) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      d2_t lv[CH];
#pragma unroll
      for (int it = 0; it < CH; ++it) {
        const int idx = tid + NT * (half * CH + it);
        const int r = min(idx >> 6, nb - 1), c = min((idx & 63) * 2, (nb - 1) & ~1);
        const double* src = A + (size_t)r * lda + c;
        if (vec) {
          lv[it] = *reinterpret_cast<const d2_t*>(src);
        } else {
          lv[it].x = src[0];
          lv[it].y = (c + 1 < nb) ? src[1] : 0.0;
        }
      }
#pragma unroll
      for (int it = 0; it < CH; ++it) {
        const int idx = tid + NT * (half * CH + it);
        const int r = idx >> 6, c = (idx & 63) * 2;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int cc = c + e;
          if (cc <= r) {
            double x = (r == cc) ? 1.0 : 0.0;
            if (r < nb) x = e ? lv[it].y : lv[it].x;
            S[blk_off(r >> 4, cc >> 4) + (r & 15) * BLD + (cc & 15)] = x;
          }
        }
      }
    }
    __syncthreads();
    for (int idx = tid; idx < 8 * 256; idx += NT) {
      const int jj = idx >> 8, i = (idx >> 4) & 15, c = idx & 15;
      double* Dj = S + blk_off(jj, jj);
      if (c > i) Dj[i * BLD + c] = Dj[c * BLD + i];
    }
  }
  // helper roles: every wave but wave 0.  (Keeping the helpers off wave 0's SIMD -- so that no MFMA shares a pipe with
  // the pivot chain's fp64 operations -- was measured: no difference.)
  const int hidx = wave - 1;
  constexpr int NH = NW - 1;
  __syncthreads();
  volatile int* ab = &fl->abort;
  const int g = lane >> 4, c = lane & 15;

  if (wave == 0) {
    // ================================ the serial chain ================================
    if (prio) __builtin_amdgcn_s_setprio(3);
    D2_TRACE(1);
    double v[4], w[4];
    {
      const double* D0 = S + blk_off(0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = D0[(g + 4 * r) * BLD + c];
    }
    for (int jj = 0; jj < 8; ++jj) {
      // ---- F(jj)
      // (lane coordinates made opaque per iteration: otherwise every lane-derived constant of the 16 unrolled pivots is
      //  hoisted out of this loop and the kernel spills ~60 registers around its serial chain)
      int cc = c, gg = g;
      asm volatile("" : "+v"(cc), "+v"(gg));
#pragma unroll
      for (int r = 0; r < 4; ++r) w[r] = (gg + 4 * r == cc) ? 1.0 : 0.0;
      double rowA = bperm_d(v[0], cc);           // row 0: lanes (0, c)
      double rowW = (cc == 0) ? 1.0 : 0.0;
      f16_step<0>(v, w, rowA, rowW, cc, gg);
      f16_step<1>(v, w, rowA, rowW, cc, gg);
      f16_step<2>(v, w, rowA, rowW, cc, gg);
      f16_step<3>(v, w, rowA, rowW, cc, gg);
      f16_step<4>(v, w, rowA, rowW, cc, gg);
      f16_step<5>(v, w, rowA, rowW, cc, gg);
      f16_step<6>(v, w, rowA, rowW, cc, gg);
      f16_step<7>(v, w, rowA, rowW, cc, gg);
      f16_step<8>(v, w, rowA, rowW, cc, gg);
      f16_step<9>(v, w, rowA, rowW, cc, gg);
      f16_step<10>(v, w, rowA, rowW, cc, gg);
      f16_step<11>(v, w, rowA, rowW, cc, gg);
      f16_step<12>(v, w, rowA, rowW, cc, gg);
      f16_step<13>(v, w, rowA, rowW, cc, gg);
      f16_step<14>(v, w, rowA, rowW, cc, gg);
      f16_step<15>(v, w, rowA, rowW, cc, gg);
      // operands of G(jj) that do not depend on this block's result are fetched now, under the post-processing below:
      // the helpers' updates of iteration jj-1 must have landed in S[jj+1][jj] and S[jj+1][jj+1]
      double sb[4];
      d4_t D;
      if (jj < 7) {
        if (jj > 0 && !d2_wait_ge(&fl->doneU, NH * jj, ab, info)) break;
        const double* Sb = S + blk_off(jj + 1, jj);
        const double* Sd = S + blk_off(jj + 1, jj + 1);
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) sb[kq] = Sb[c * BLD + kq * 4 + g];       // B operand of Y = inv(L_jj) S[jj+1][jj]^T
#pragma unroll
        for (int r = 0; r < 4; ++r) D[r] = Sd[(g + 4 * r) * BLD + c];
      }
      D2_TRACE(3 + 3 * jj);
      // the pivot of column c sits, unscaled, on the parked column's diagonal: lane (c & 3, c), register c >> 2
      const int q = c >> 2;
      const double dsel = (q == 0) ? v[0] : (q == 1) ? v[1] : (q == 2) ? v[2] : v[3];
      const double dcol = bperm_d(dsel, 16 * (c & 3) + c);
      const double rs = rsqrt_nr(dcol);          // 1 / sqrt(pivot of column c)
      // first non-positive pivot of the block (a NaN counts): lanes 0..15 carry columns 0..15
      const unsigned long long nonpos = __ballot(!(dcol > 0.0)) & 0xffffull;
      const int bad = nonpos ? __ffsll((long long)nonpos) : 0;
      double* Dj = S + blk_off(jj, jj);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = g + 4 * r;
        const double rsi = bperm_d(rs, i);       // 1 / sqrt(pivot of column i): the row scaling of the inverse
        const double x = (i >= c) ? w[r] * rsi : 0.0;
        Dj[i * BLD + c] = x;                                              // inv(L_jj) for the helpers and for G
        const int gr = jj * 16 + i, gc = jj * 16 + c;
        if (i >= c) {
          Dinv[(size_t)gr * NB + gc] = x;
          if (gr < nb) A[(size_t)gr * lda + gc] = v[r] * rs;              // L_jj
        }
      }
      if (bad && lane == 0 && (jj * 16 + bad) <= nb) atomicCAS(info, 0, row_base + jj * 16 + bad);
      D2_LDS_FENCE();
      if (lane == 0) d2_st(&fl->seqF, jj + 1);
      D2_TRACE(2 + 3 * jj);
      if (jj == 7) break;
      // ---- G(jj)
      d4_t Y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) Y = __builtin_amdgcn_mfma_f64_16x16x4f64(Dj[c * BLD + kq * 4 + g], sb[kq], Y, 0, 0, 0);
      {
        double* Xb = S + blk_off(jj + 1, jj);                  // L[jj+1][jj] = Y^T  (its old content sits in sb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int k = g + 4 * r;                             // Y[k][c] -> X[c][k]
          Xb[c * BLD + k] = Y[r];
          const int gr = (jj + 1) * 16 + c;
          if (gr < nb) A[(size_t)gr * lda + jj * 16 + k] = Y[r];
        }
      }
      D2_LDS_FENCE();
      if (lane == 0) d2_st(&fl->seqX, jj + 1);
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) D = __builtin_amdgcn_mfma_f64_16x16x4f64(Y[kq], Y[kq], D, 0, 0, 1);   // D -= Y^T Y
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = D[r];
      D2_TRACE(4 + 3 * jj);
    }
    return;
  }

  // ================================ helpers ================================
  int sb_target = 0;
  auto helper_barrier = [&]() -> bool {
    sb_target += NH;
    D2_LDS_FENCE();
    if (lane == 0) atomicAdd(const_cast<int*>(&fl->sb), 1);
    return d2_wait_ge(&fl->sb, sb_target, ab, info);
  };
  d4_t T[2];                                   // T_j for the columns j = hidx, hidx + NH of the inverse's next row block
  T[0] = (d4_t){0.0, 0.0, 0.0, 0.0};
  T[1] = T[0];
  for (int jj = 0; jj < 8; ++jj) {
    if (!d2_wait_ge(&fl->seqF, jj + 1, ab, info)) return;
    if (hidx == 0) D2_TRACE(32 + 4 * jj);
    const double* Wj = S + blk_off(jj, jj);    // inv(L_jj)
    // A1: L[i][jj] = S[i][jj] inv(L_jj)^T for the block rows wave 0 does not take itself
    for (int i = jj + 2 + hidx; i < 8; i += NH) {
      d4_t acc = {0.0, 0.0, 0.0, 0.0};
      double* Bij = S + blk_off(i, jj);
      mma16<true>(acc, Bij, BLD, Wj, BLD, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Bij[(g + 4 * r) * BLD + c] = acc[r];
        const int gr = i * 16 + g + 4 * r;
        if (gr < nb) A[(size_t)gr * lda + jj * 16 + c] = acc[r];
      }
    }
    // A2: X[jj][j] = -inv(L_jj) T_j (T from B3 of the previous iteration; accumulator layout = B-operand layout)
    for (int s = 0; s < 2; ++s) {
      const int j = hidx + s * NH;
      if (j < jj) {
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kq = 0; kq < 4; ++kq)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Wj[c * BLD + kq * 4 + g], s ? T[1][kq] : T[0][kq], acc, 0, 0, 0);
        double* dst = S + blk_off(jj, j);
        double* gd = Dinv + (size_t)(jj * 16 + g) * NB + j * 16 + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dst[(g + 4 * r) * BLD + c] = -acc[r];
          gd[(size_t)4 * r * NB] = -acc[r];
        }
      }
    }
    if (hidx == 0) D2_TRACE(33 + 4 * jj);
    if (jj == 7) break;
    if (!helper_barrier()) return;
    if (!d2_wait_ge(&fl->seqX, jj + 1, ab, info)) return;
    // B1 / B2, dealt round-robin in priority order: (B1 i, B2 i) for i = jj+2 .. 7
    {
      const int nrows = 6 - jj;                // block rows jj+2 .. 7
      for (int q = hidx; q < 2 * nrows; q += NH) {
        const int i = jj + 2 + (q >> 1);
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
        double* dst;
        if ((q & 1) == 0) {                    // B1: column jj+1 receives block column jj
          mma16<true>(acc, S + blk_off(i, jj), BLD, S + blk_off(jj + 1, jj), BLD, lane);
          dst = S + blk_off(i, jj + 1);
        } else {                               // B2: column jj+2 receives block columns 0 .. jj
          for (int p = 0; p <= jj; ++p) mma16<true>(acc, S + blk_off(i, p), BLD, S + blk_off(jj + 2, p), BLD, lane);
          dst = S + blk_off(i, jj + 2);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(g + 4 * r) * BLD + c] -= acc[r];
      }
    }
    D2_LDS_FENCE();
    if (lane == 0) atomicAdd(const_cast<int*>(&fl->doneU), 1);
    if (hidx == 0) D2_TRACE(34 + 4 * jj);
    // B3: T_j = sum_{k=j}^{jj} L[jj+1][k] X[k][j] for the owned columns j <= jj of the inverse's row block jj+1
    for (int s = 0; s < 2; ++s) {
      const int j = hidx + s * NH;
      d4_t acc = {0.0, 0.0, 0.0, 0.0};
      if (j <= jj)
        for (int k = j; k <= jj; ++k) mma16<false>(acc, S + blk_off(jj + 1, k), BLD, S + blk_off(k, j), BLD, lane);
      if (s) T[1] = acc; else T[0] = acc;
    }
    if (hidx == 0) D2_TRACE(35 + 4 * jj);
    if (!helper_barrier()) return;             // row block jj+1 of L is dead now: A2 of the next iteration overwrites it
  }
}

// ------------------------------------------------------------------------------------------------------------
// naive reference kernels (debug / on-device cross-check only; selected with option "naive")
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ffgp_potrf_naive(double* A, int lda, int n, int* info) {
  __shared__ double piv;
  __shared__ int failed;
  for (int j = 0; j < n; ++j) {
    if (threadIdx.x == 0) {
      double d = A[(size_t)j * lda + j];
      failed = 0;
      if (!(d > 0.0)) {
        atomicCAS(info, 0, j + 1);
        d = 1.0;
      }
      piv = sqrt(d);
      A[(size_t)j * lda + j] = piv;
    }
    __syncthreads();
    const double p = piv;
    for (int i = j + 1 + threadIdx.x; i < n; i += blockDim.x) A[(size_t)i * lda + j] /= p;
    __threadfence_block();
    __syncthreads();
    // rank-1 update of the trailing lower triangle
    const long m = n - j - 1;
    for (long e = threadIdx.x; e < m * m; e += blockDim.x) {
      const int r = j + 1 + (int)(e / m), c = j + 1 + (int)(e % m);
      if (c <= r) A[(size_t)r * lda + c] -= A[(size_t)r * lda + j] * A[(size_t)c * lda + j];
    }
    __threadfence_block();
    __syncthreads();
  }
}

// passenger rows, naive: row <- row * L^-T (one wave per row, serial substitution)
__global__ __launch_bounds__(64) void ffgp_trsm_rows_naive(double* A, int lda, int n) {
  double* x = A + (size_t)(n + blockIdx.x) * lda;
  for (int c = 0; c < n; ++c) {
    double part = 0.0;
    for (int k = threadIdx.x; k < c; k += 64) part += x[k] * A[(size_t)c * lda + k];
    for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o);
    if (threadIdx.x == 0) x[c] = (x[c] - part) / A[(size_t)c * lda + c];
    __threadfence_block();
    __syncthreads();
  }
}

// one workgroup per 128-block: dense inverse of the lower-triangular diagonal block by forward substitution
__global__ __launch_bounds__(128) void ffgp_dinv_naive(const double* L, int ldl, int n, double* Dinv) {
  const int b = blockIdx.x, c = threadIdx.x;
  const int r0 = b * NB;
  const int nb = min(NB, n - r0);
  double* X = Dinv + (size_t)b * NB * NB;
  for (int i = 0; i < NB; ++i) {
    double s = (i == c) ? 1.0 : 0.0;
    double dii = 1.0;
    if (i < nb) {
      dii = L[(size_t)(r0 + i) * ldl + r0 + i];
      for (int k = 0; k < i; ++k) s -= L[(size_t)(r0 + i) * ldl + r0 + k] * X[k * NB + c];
    }
    X[i * NB + c] = (i >= c) ? s / dii : 0.0;
  }
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
static int launch_diag(ffgp_handle* h, double* Ablk, int lda, int nb, double* Dinv_blk, int row_base, int do_factor) {
  if (!(h->diag_attr_set & 1)) {   // per handle = per device (the attribute lives in the device's context)
    FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_potrf_diag128),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, DIAG_LDS_BYTES));
    h->diag_attr_set |= 1;
  }
  if (do_factor && h->diag_v2 && !h->diag_dbg) {   // the pipelined kernel (the barrier version keeps the inverse-only entry)
    if (!(h->diag_attr_set & 2)) {
      FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_potrf_diag128_v2<8>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, DIAG_LDS_BYTES));
      h->diag_attr_set |= 2;
    }
    hipLaunchKernelGGL(ffgp_potrf_diag128_v2<8>, dim3(1), dim3(512), DIAG_LDS_BYTES, h->stream, Ablk, lda, nb, Dinv_blk, h->d_info,
                       row_base, h->aux_prio);
    return FFGP_OK;
  }
  hipLaunchKernelGGL(ffgp_potrf_diag128, dim3(1), dim3(DIAG_THREADS), DIAG_LDS_BYTES, h->stream, Ablk, lda, nb, Dinv_blk,
                     h->d_info, row_base, do_factor, h->diag_dbg, h->aux_prio);
  return FFGP_OK;
}

int ffgp_ensure_dinv(ffgp_handle* h, int n) {
  const int nblk = (n + NB - 1) / NB;
  const size_t need = (size_t)nblk * NB * NB * sizeof(double);
  if (need > h->dinv_bytes) {
    // grow with head-room and KEEP the content: a factor that gains rows (functional.Posterior.append) keeps the inverses
    // of its leading blocks, and ffgp_refresh_dinv only builds the new ones
    const size_t want = need + need / 4 + (size_t)8 * NB * NB * sizeof(double);
    double* fresh = nullptr;
    if (hipMalloc(&fresh, want) != hipSuccess) return FFGP_ERR_ALLOC;
    // strictly-upper parts stay zero forever.  The memset must be ORDERED with the kernels that fill the store: a plain
    // hipMemset runs on the NULL stream, which does not synchronise with the (non-blocking) streams this library works
    // on -- it could land after the first diagonal-block kernel had written its inverse and wipe it (seen as wrong
    // factors on the first use of a fresh handle only).
    if (hipMemsetAsync(fresh, 0, want, h->stream) != hipSuccess) return FFGP_ERR_HIP;
    if (h->dinv) {
      if (h->aux) hipStreamSynchronize(h->aux);   // work enqueued earlier on this handle may still read / write the old store
      if (hipMemcpyAsync(fresh, h->dinv, h->dinv_bytes, hipMemcpyDeviceToDevice, h->stream) != hipSuccess) return FFGP_ERR_HIP;
      hipStreamSynchronize(h->stream);
      hipFree(h->dinv);
    }
    h->dinv = fresh;
    h->dinv_bytes = want;
    ++h->alloc_epoch;
  }
  return FFGP_OK;
}

// (re)build the inverted diagonal blocks for a factor that is already in L (used when a caller hands us a
// factor this handle did not just produce)
int ffgp_refresh_dinv(ffgp_handle* h, const double* L, int n, int ldl) {
  // a factor that only GREW since the store was built (rows appended to the same buffer, functional.Posterior.append)
  // keeps its leading blocks: rebuild from the last, possibly partial, old block on -- unless the store must be
  // re-allocated, which drops its content
  int b_first = 0;
  if (!h->use_naive && h->dinv_L == L && h->dinv_ld == ldl && h->dinv_n > 0 && n > h->dinv_n)
    b_first = h->dinv_n / NB;
  FFGP_CHECK(ffgp_ensure_dinv(h, n));
  if (b_first == 0) h->sinv_L = nullptr;   // a factor that only grew keeps its leading super-block inverses as well
  const int nblk = (n + NB - 1) / NB;
  if (h->use_naive) {
    hipLaunchKernelGGL(ffgp_dinv_naive, dim3(nblk), dim3(128), 0, h->stream, L, ldl, n, h->dinv);
  } else {
    for (int b = b_first; b < nblk; ++b) {
      const int r0 = b * NB;
      FFGP_CHECK(launch_diag(h, const_cast<double*>(L) + (size_t)r0 * ldl + r0, ldl, min(NB, n - r0),
                             h->dinv + (size_t)b * NB * NB, r0, 0));
    }
  }
  h->dinv_L = L;
  h->dinv_n = n;
  h->dinv_ld = ldl;
  return FFGP_OK;
}

// factor one outer panel (columns k0 .. k0+w1) of the (mtot x n) matrix on h->stream: per 128-column block a
// diagonal factor+inverse, the TRSM of every row below as one GEMM, and the update of the panel's remaining columns
// `gate` (nullable): event the stream waits on before the panel's first update GEMM -- the look-ahead driver lets
// the first diagonal factor + TRSM start as soon as the panel's first 128 columns carry the trailing update
// `carry` (> 0): every update of the panel also covers the `carry` columns to the right of it -- the first block of the NEXT
// panel -- so that block is complete the moment this panel is and no strip update sits between two panels on the
// dependency chain; `gate2` (nullable) is waited for together with `gate` (the main stream's earlier contribution to
// those columns must have landed first)
static int factor_panel(ffgp_handle* h, double* A, int n, int mtot, int lda, int k0, int w1, hipEvent_t gate = nullptr, int carry = 0,
                        hipEvent_t gate2 = nullptr) {
  const int pend = k0 + w1;
  for (int j0 = k0; j0 < pend; j0 += NB) {
    const int jb = min(NB, n - j0);
    double* Ajj = A + (size_t)j0 * lda + j0;
    double* Dj = h->dinv + (size_t)(j0 / NB) * NB * NB;
    FFGP_CHECK(launch_diag(h, Ajj, lda, jb, Dj, j0, 1));
    const int mrows = mtot - (j0 + jb);
    if (mrows > 0) {
      double* A21 = A + (size_t)(j0 + jb) * lda + j0;
      // TRSM as GEMM: A21 <- A21 * Dj^T (in place: one column tile, each workgroup rewrites only rows it read)
      FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_KMAJOR, TILES_FULL, 0, A21, lda, Dj, NB, A21, lda, mrows, jb, jb, 1.0, 0.0, 0,
                                  ALIAS_A));
      const int wrem = pend - (j0 + jb) + carry;
      if (wrem > 0) {
        if (gate && j0 == k0) FFGP_HIP(hipStreamWaitEvent(h->stream, gate, 0));
        if (gate2 && j0 == k0) FFGP_HIP(hipStreamWaitEvent(h->stream, gate2, 0));
        double* C = A + (size_t)(j0 + jb) * lda + (j0 + jb);
        FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_KMAJOR, TILES_LOWER, 0, A21, lda, A21, lda, C, lda, mrows, wrem, jb, -1.0,
                                    1.0));
      }
    }
  }
  return FFGP_OK;
}

// Factor the leading n x n block of A in place; rows n..mtot-1 (if any) are "passenger" rows that receive the
// same right-hand transformations and come out as  A[n:, :] * L^-T  -- i.e. (L^-1 B)^T for B^T stored below
// Sigma.  The fused NLML/predict paths put Y^T and K_*^T there, so the triangular solves ride inside the
// factorisation's own GEMMs (no separate TRSM sweeps).
int ffgp_potrf_impl(ffgp_handle* h, double* A, int n, int mtot, int lda, int sync_info) {
  if (n <= 0) return FFGP_OK;
  if (!A || lda < n || mtot < n) return FFGP_ERR_ARG;
  if ((lda & 1) || (reinterpret_cast<uintptr_t>(A) & 15)) {
    fprintf(stderr, "[ffgp] potrf: A must be 16-byte aligned with an even leading dimension\n");
    return FFGP_ERR_ARG;
  }
  FFGP_CHECK(ffgp_ensure_dinv(h, n));
  FFGP_CHECK(ffgp_zero_async(h, h->d_info, sizeof(int)));
  h->dinv_L = nullptr;
  h->sinv_L = nullptr;   // super-block inverses belong to the factor that is about to be overwritten

  if (h->use_naive) {
    hipLaunchKernelGGL(ffgp_potrf_naive, dim3(1), dim3(256), 0, h->stream, A, lda, n, h->d_info);
    FFGP_CHECK(ffgp_refresh_dinv(h, A, n, lda));
    if (mtot > n) hipLaunchKernelGGL(ffgp_trsm_rows_naive, dim3(mtot - n), dim3(64), 0, h->stream, A, lda, n);
  } else {
    const int NB1 = h->nb_outer;
    // panel width at column k0: the wide block while more than nb_big_until columns remain (the SYRK's fixed per-tile
    // cost is amortised over a longer k loop where the chain still hides under it), nb_outer after that
    auto pw = [&](int k0) { return (h->nb_big > NB1 && n - k0 > h->nb_big_until) ? h->nb_big : NB1; };
    if (!h->lookahead || n <= NB1 || n <= h->la_min_n) {
      for (int k0 = 0; k0 < n; k0 += pw(k0)) {
        const int w1 = min(pw(k0), n - k0);
        const int pend = k0 + w1;  // end column of this outer panel
        FFGP_CHECK(factor_panel(h, A, n, mtot, lda, k0, w1));
        const int mt = n - pend;  // trailing columns; trailing rows include the passenger rows
        if (mt > 0) {
          double* P = A + (size_t)pend * lda + k0;
          double* C = A + (size_t)pend * lda + pend;
          FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_KMAJOR, TILES_LOWER, 1, P, lda, P, lda, C, lda, mtot - pend, mt, w1,
                                      -1.0, 1.0));
        }
      }
    } else if (h->la_carry == 1 || (h->la_carry == 2 && n <= 12288)) {
      // Look-ahead, "carry" form.  Panel k's own update kernels (one per 128-column block, K = 128) also cover Z(k+1) = the
      // first 128 columns of panel k+1, so the chain goes from the last TRSM of panel k straight into the first diagonal
      // block of panel k+1 -- no K = 512 strip update (S_a of the form below) and no event wait between two panels.
      // The main stream's trailing update of step k is cut into S_b (the rest of panel k+1's columns: gates the chain's
      // first update, as below) together with S_z (Z(k+2): panel k's contribution to the strip that chain k+1 will carry
      // into -- its right-hand neighbour, so the two are one launch) and S_ii (everything right of Z(k+2)).  Writers of any one column range
      // are ordered: Z(k+2) <- S_ii(<= k-1), S_z(k) on the main stream, then chain k+1 (after S_z's event).
      hipStream_t main_s = h->stream;
      auto carry_of = [&](int pend_) { return min(NB, n - pend_); };   // columns of the next panel's first block (0 at the end)
      FFGP_CHECK(factor_panel(h, A, n, mtot, lda, 0, min(pw(0), n), nullptr, max(0, carry_of(min(pw(0), n)))));
      FFGP_HIP(hipEventRecord(h->la_ev[6], main_s));
      FFGP_HIP(hipStreamWaitEvent(h->aux, h->la_ev[6], 0));
      int it = 0;
      hipEvent_t eb_prev = nullptr;
      for (int k0 = 0; k0 < n; k0 += pw(k0), ++it) {
        const int w1 = min(pw(k0), n - k0);
        const int pend = k0 + w1;
        const int mt = n - pend;
        if (mt <= 0) break;
        const int wn = min(pw(pend), mt);  // width of the next panel
        const int q = pend + wn;           // first column of panel k+2
        const int wz = max(0, min(NB, n - q));
        hipEvent_t eb = h->la_ev[(it & 1) * 3], eg = h->la_ev[(it & 1) * 3 + 1];
        const int wa = min(NB, wn);        // Z(k+1): already complete (carried by panel k)
        // main stream, once panel k is complete: S_b(k) and S_z(k) are neighbours (columns pend+wa .. q+wz): one launch, one event
        if (eb_prev) FFGP_HIP(hipStreamWaitEvent(main_s, eb_prev, 0));
        hipEvent_t gate = nullptr, gate2 = nullptr;
        if (wn - wa + wz > 0) {
          double* Pb = A + (size_t)(pend + wa) * lda + k0;
          double* Cb = A + (size_t)(pend + wa) * lda + (pend + wa);
          FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_KMAJOR, TILES_LOWER, 1, Pb, lda, Pb, lda, Cb, lda, mtot - pend - wa,
                                      wn - wa + wz, w1, -1.0, 1.0));
          FFGP_HIP(hipEventRecord(eg, main_s));
          gate = eg;
        }
        // side stream: panel k+1, carrying Z(k+2)
        h->stream = h->aux;
        int rc = factor_panel(h, A, n, mtot, lda, pend, wn, gate, wz, gate2);
        h->stream = main_s;
        FFGP_CHECK(rc);
        FFGP_HIP(hipEventRecord(eb, h->aux));
        eb_prev = eb;
        if (h->tri_hook_col > 0 && pend + wn == h->tri_hook_col) {   // the factor's columns < tri_hook_col are final from here on
          FFGP_HIP(hipEventRecord(h->tri_ev[0], h->aux));
          h->tri_hook_fired = 1;
        }
        // main stream: S_ii(k), everything right of Z(k+2)
        const int mt2 = mt - wn - wz;
        if (mt2 > 0) {
          double* P2 = A + (size_t)(q + wz) * lda + k0;
          double* C2 = A + (size_t)(q + wz) * lda + (q + wz);
          FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_KMAJOR, TILES_LOWER, 1, P2, lda, P2, lda, C2, lda, mtot - q - wz, mt2, w1, -1.0,
                                      1.0));
        }
      }
      if (eb_prev) FFGP_HIP(hipStreamWaitEvent(main_s, eb_prev, 0));
    } else {
      // Look-ahead.  The trailing update of step k is cut into S_a (the first 128 columns of panel k+1 -- all that its
      // first diagonal factor and TRSM read), S_b (the rest of panel k+1's columns) and S_ii (everything to the right).
      // The side stream (high priority) runs the dependency chain  S_a(k) -> panel k+1 (diag -> TRSM -> update, x4)
      // back to back; the main stream runs S_b(k), S_ii(k).  While the trailing matrix is large the chain hides under
      // S_ii; once it is small the chain IS the critical path -- which is why S_a sits on the chain's own stream:
      // a cross-stream event hand-off costs ~13 us, and the only waits left on the chain are for events that fired
      // long before (S_ii(k-1), S_b(k)).  Panel k+1 touches only its own columns; S_ii reads panel k and writes the
      // columns to the right of panel k+1, so the two streams never alias.
      hipStream_t main_s = h->stream;
      FFGP_CHECK(factor_panel(h, A, n, mtot, lda, 0, min(pw(0), n)));
      FFGP_HIP(hipEventRecord(h->la_ev[6], main_s));
      FFGP_HIP(hipStreamWaitEvent(h->aux, h->la_ev[6], 0));
      int it = 0;
      hipEvent_t eb_prev = nullptr, ei_prev = nullptr;
      for (int k0 = 0; k0 < n; k0 += pw(k0), ++it) {
        const int w1 = min(pw(k0), n - k0);
        const int pend = k0 + w1;
        const int mt = n - pend;
        if (mt <= 0) break;
        const int wn = min(pw(pend), mt);  // width of the next panel
        double* P = A + (size_t)pend * lda + k0;
        double* C = A + (size_t)pend * lda + pend;
        hipEvent_t eb = h->la_ev[(it & 1) * 3], eg = h->la_ev[(it & 1) * 3 + 1], ei = h->la_ev[(it & 1) * 3 + 2];
        const int wa = h->la_split ? min(NB, wn) : wn;
        // side stream: S_a(k) (after S_ii(k-1), which carried panel k-1 into these columns)
        if (ei_prev) FFGP_HIP(hipStreamWaitEvent(h->aux, ei_prev, 0));
        h->stream = h->aux;
        int rc = ffgp_gemm_launch(h, OP_KMAJOR, OP_KMAJOR, TILES_LOWER, 1, P, lda, P, lda, C, lda, mtot - pend, wa, w1, -1.0, 1.0);
        h->stream = main_s;
        FFGP_CHECK(rc);
        // main stream: S_b(k) once panel k is complete
        if (eb_prev) FFGP_HIP(hipStreamWaitEvent(main_s, eb_prev, 0));
        hipEvent_t gate = nullptr;
        if (wn > wa) {
          double* Pb = A + (size_t)(pend + wa) * lda + k0;
          double* Cb = A + (size_t)(pend + wa) * lda + (pend + wa);
          FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_KMAJOR, TILES_LOWER, 1, Pb, lda, Pb, lda, Cb, lda, mtot - pend - wa, wn - wa,
                                      w1, -1.0, 1.0));
          FFGP_HIP(hipEventRecord(eg, main_s));
          gate = eg;
        }
        // side stream: panel k+1
        h->stream = h->aux;
        rc = factor_panel(h, A, n, mtot, lda, pend, wn, gate);
        h->stream = main_s;
        FFGP_CHECK(rc);
        FFGP_HIP(hipEventRecord(eb, h->aux));
        eb_prev = eb;
        if (h->tri_hook_col > 0 && pend + wn == h->tri_hook_col) {   // the factor's columns < tri_hook_col are final from here on
          FFGP_HIP(hipEventRecord(h->tri_ev[0], h->aux));
          h->tri_hook_fired = 1;
        }
        // main stream: S_ii(k), the rest of the trailing matrix
        const int mt2 = mt - wn;
        ei_prev = nullptr;
        if (mt2 > 0) {
          double* P2 = A + (size_t)(pend + wn) * lda + k0;
          double* C2 = A + (size_t)(pend + wn) * lda + (pend + wn);
          FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_KMAJOR, TILES_LOWER, 1, P2, lda, P2, lda, C2, lda, mtot - pend - wn, mt2,
                                      w1, -1.0, 1.0));
          FFGP_HIP(hipEventRecord(ei, main_s));
          ei_prev = ei;
        }
      }
      if (eb_prev) FFGP_HIP(hipStreamWaitEvent(main_s, eb_prev, 0));
    }
    h->dinv_L = A;
    h->dinv_n = n;
    h->dinv_ld = lda;
  }
  if (!sync_info) return FFGP_OK;
  FFGP_HIP(hipMemcpyAsync(h->h_info, h->d_info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  FFGP_HIP(hipStreamSynchronize(h->stream));
  return ffgp_map_info(h->h_info[0]);
}

// status word -> return code: a pivot index passes through; the diagonal-block kernel's watchdog is a library error
int ffgp_map_info(int v) {
  if (v >= FFGP_DIAG_WATCHDOG) {
    fprintf(stderr, "[ffgp] potrf_diag128: a wave waited ~1 s for a hand-off inside the kernel and gave up (internal error)\n");
    return FFGP_ERR_HIP;
  }
  return v;
}
