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
#include "f16_steps.h"

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
#define FFGP_HANDOFF_WATCHDOG 0x7fffffe0     // a look-ahead gate gave up waiting for its hand-off (ffgp_handoff_gate)

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
    {                                                                                           \
      ffgp_diag_trace_buf[(slot)] = wall_clock64();                                             \
      if ((slot) == 0 || (slot) == 23) ffgp_diag_trace_buf[120 + ((slot) != 0)] = __builtin_readcyclecounter();   /* shader clock */ \
    }                                                                                           \
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
  // every lane read the same word: saying so (v_readfirstlane) lets the compiler keep the polling loops, the loop counters compared
  // with the flags and every branch on them SCALAR -- without it the whole helper section was compiled as divergent control flow
  // (exec-mask juggling around every task, loop counters in vector registers)
  return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void d2_st(volatile int* p, int v) {
  const unsigned off = (unsigned)(uintptr_t)p;
  asm volatile("ds_write_b32 %0, %1" : : "v"(off), "v"(v) : "memory");
}

__device__ __forceinline__ bool d2_wait_ge(volatile int* p, int target, volatile int* abort_flag, int* info, int site = 0) {
  int spins = 0;
  while (d2_ld(p) < target) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 63) == 0 && (spins > (1 << 21) || d2_ld(abort_flag))) {   // ~1 s of polling: something upstream died
      if (!d2_ld(abort_flag)) atomicExch(info, FFGP_DIAG_WATCHDOG + site);    // (site: which hand-off; ffgp_map_info prints it)
      d2_st(abort_flag, 1);
      return false;
    }
  }
  D2_COMPILER_FENCE();
  return true;
}

#ifdef FFGP_DEV_OPTIONS   // the round-3 pipelines (diag_v2 = 1, 3): development build only
template <int NW, bool DPP64>
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
      if constexpr (DPP64) {
        double hA = bperm_d(v[0], 16 + cc), hW = (cc == 1) ? 1.0 : 0.0;    // row 1 (lanes (1, c), register 0), one step ahead at the start only
        double pRow = 0.0, pt = 0.0, ptw = 0.0;
        double dcur = row_bcast64<0>(rowA), ycur = __builtin_amdgcn_rcp(dcur);
#define F16_S(JJ) f16_step_dpp<JJ>(v, w, rowA, rowW, hA, hW, pRow, pt, ptw, dcur, ycur, cc, gg);
        F16_S(0) F16_S(1) F16_S(2) F16_S(3) F16_S(4) F16_S(5) F16_S(6) F16_S(7) F16_S(8) F16_S(9) F16_S(10) F16_S(11) F16_S(12) F16_S(13)
        F16_S(14) F16_S(15)
#undef F16_S
      } else {
#define F16_S(JJ) f16_step<JJ>(v, w, rowA, rowW, cc, gg);
        F16_S(0) F16_S(1) F16_S(2) F16_S(3) F16_S(4) F16_S(5) F16_S(6) F16_S(7) F16_S(8) F16_S(9) F16_S(10) F16_S(11) F16_S(12) F16_S(13)
        F16_S(14) F16_S(15)
#undef F16_S
      }
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

#endif   // FFGP_DEV_OPTIONS

// ------------------------------------------------------------------------------------------------------------
// potrf_diag128_v3 (round 4): the same chain, with the helpers reorganised around what tools/native/f16_probe.hip measured.
//
//  * fp64 MFMAs and fp64 vector instructions share a SIMD's double-precision pipe: ONE MFMA-streaming wave on wave 0's own SIMD
//    takes the pivot loop from 148 to 270-280 cycles per pivot (s_setprio does not help: an issued 64-cycle MFMA is not
//    pre-empted); six streaming waves on the three OTHER SIMDs cost nothing (150).  An 8-wave workgroup puts two waves on each
//    SIMD, so the wave that shares wave 0's SIMD (found by HW_ID.SIMD_ID, not assumed) takes no part in the arithmetic: it leaves
//    at once.  Six helpers remain.
//  * No load phase and no workgroup barrier after the role hand-out: wave 0 reads its first diagonal block straight from global
//    memory into the factor's register layout and starts; every other 16 x 16 block has an OWNER wave (column-major round
//    robin) that loads it from global memory, keeps its running value in the block's LDS home as a REGISTER IMAGE of the
//    transposed block (lane (g, c), register r = S[c][g + 4r]: exactly the B operand of the triangular solve and of wave 0's
//    G, so no product ever needs a layout change), applies every update of that block itself, in order -- right-looking:
//    column s of L updates all later columns as soon as it exists -- and finally solves it (or hands it to wave 0).  No two
//    waves ever write the same block, so there are no barriers: a bit per block says "L[i][k] is final" (rows[k], bit i), two counters
//    hand wave 0 its next operands (hs, hd).  The round-3 kernel ran left-looking with one column of look-ahead and two
//    7-wave counter barriers per iteration: its helpers needed 3.0-3.3 us per iteration in the middle of the block, more than
//    wave 0's 2.6 -- the chain waited for them.
//  * The inverse's row block s, X[s][j] = -inv(L_s) sum_k L[s][k] X[k][j], overwrites row s of L in place; every helper counts
//    itself in (cntA[s]) once it has finished reading that row, and a row block is stored only when all six have.
// Stage s of a helper (s = 0 .. 7), after seqF >= s + 1 (inv(L_s) is in LDS):
//    (1) L[i][s]^T = inv(L_s) S[i][s]^T for its blocks of column s, i >= s + 2 (row s + 1 is wave 0's G)            -> rows[s] bit i
//    (2) S[i][k]^T -= L[k][s] L[i][s]^T for its blocks with k > s (column s + 1 first); a block whose updates are complete and
//        that wave 0 needs next is announced: (s + 2, s + 1) -> hs, the diagonal block (s + 2, s + 2) -> hd
//    (3) its columns of the inverse's row block s
// ------------------------------------------------------------------------------------------------------------
struct D3Flags {      // ints in LDS, behind the block image
  int seqF, h2, hs, hd, abort, roles, pad0[2];
  int simd[8];
  int cntA[8];
  int rows[8];        // rows[s]: bit i = L[i][s] is final and in LDS (bit s + 1 is set by wave 0's G(s), the others by the blocks' owners)
  int prog[8];        // [0] wave 0: 16 jj + phase; [1 + hidx]: the helper's current task index (post-mortem of a timed-out hand-off)
};

#define D3_NH 6
// The helpers' work as a STATIC task list per helper (the block structure is fixed, so is the schedule).  A first version walked
// nested loops over (stage, pass, owned block) with the ownership tests inline: on this machine a not-taken scalar branch costs as
// much as four vector instructions, and a stage with nothing to do took 2.5 us of pure control flow.  Now lane t of a helper holds
// descriptor t of its list (one load at the start), v_readlane fetches the next one, and one switch dispatches it.
//   descriptor: bits 1:0 type, 4:2 i, 7:5 k, 9:8 flag to raise (0 none, 1 hs, 2 hd, 3 h2), 12:10 stage s, 15:13 first column p0 of an
//   update (it applies columns p0 .. s); 0xffff ends the list
// Ownership: block t of the column-major enumeration (k = 0..7, i = k..7, without (0, 0)) belongs to helper t % 6.
// WHEN a block receives column p.  Right-looking (at stage p, the moment column p exists) puts 28 + 21 + 15 of the 77 updates into the
// first three stages, where the helpers then lag behind wave 0; left-looking (everything at stage k - 1) starves them early and
// makes the last stages long.  In between: block (i, k) takes column p at stage max(p, k - 3) -- the two columns wave 0 needs next
// stay current, column s + 3 catches up on columns 0 .. s in ONE task (one read and one write of the image, 4 (s + 1) products),
// later columns wait: 18, 19, 18, 15, 10, 3, 1 block visits per stage instead of 28, 21, 15, 10, 6, 3, 1.
// Order inside a stage s (what wave 0's G(s + 1) waits for comes first -- it needs (s + 2, s + 1), (s + 2, s + 2) and (s + 3, s + 1)
// with column s applied): STAGE (wait for inv(L_s)); the solve of (s + 3, s), the one operand of those three that wave 0 does not
// produce itself; those three updates; the other solves of column s; the other updates, next column first; the inverse's row s.
enum { D3_STAGE = 0, D3_TRSM = 1, D3_UPDATE = 2, D3_INVERSE = 3, D3_END = 0xffff, D3_MAXTASKS = 64 };
struct D3TaskTable { unsigned short t[D3_NH][D3_MAXTASKS]; };
constexpr int d3_owner(int i, int k) { return (k == 0 ? i - 1 : 8 * k - 1 - k * (k - 1) / 2 + (i - k)) % D3_NH; }
constexpr unsigned short d3_desc(int type, int i, int k, int flag, int s, int p0 = 0) {
  return (unsigned short)(type | (i << 2) | (k << 5) | (flag << 8) | (s << 10) | (p0 << 13));
}
constexpr int d3_flag_of(int i, int k, int s) {      // which of wave 0's next operands block (i, k) is once column s is applied
  return (k == s + 1 && i == s + 2) ? 1 : (k == s + 2 && i == k) ? 2 : (k == s + 1 && i == s + 3) ? 3 : 0;
}
constexpr D3TaskTable d3_make_tasks() {
  D3TaskTable T{};
  for (int h = 0; h < D3_NH; ++h) {
    int n = 0;
    for (int s = 0; s < 8; ++s) {
      T.t[h][n++] = d3_desc(D3_STAGE, 0, 0, 0, s);
      if (s + 3 < 8 && d3_owner(s + 3, s) == h) T.t[h][n++] = d3_desc(D3_TRSM, s + 3, s, 0, s);
      for (int pass = 0; pass < 2; ++pass) {            // pass 0: the three urgent updates; pass 1: the other solves, the other updates
        if (pass == 1)
          for (int i = s + 4; i < 8; ++i)
            if (d3_owner(i, s) == h) T.t[h][n++] = d3_desc(D3_TRSM, i, s, 0, s);
        for (int k = s + 1; k < 8 && k <= s + 3; ++k)
          for (int i = k; i < 8; ++i) {
            if (d3_owner(i, k) != h || (i == k && k == s + 1)) continue;      // (the diagonal block of column s + 1 is wave 0's)
            const int fl = d3_flag_of(i, k, s);
            if ((fl != 0) == (pass == 0)) T.t[h][n++] = d3_desc(D3_UPDATE, i, k, fl, s, (k == s + 3) ? 0 : s);
          }
      }
      if (s >= 1) T.t[h][n++] = d3_desc(D3_INVERSE, 0, 0, 0, s);
    }
    for (; n < D3_MAXTASKS; ++n) T.t[h][n] = (unsigned short)D3_END;
  }
  return T;
}
__device__ const D3TaskTable D3_TASKS = d3_make_tasks();

// LDS read-modify-write on a flag word, as explicit DS instructions (a generic-pointer atomic compiles to flat_atomic_*)
__device__ __forceinline__ void d3_or(volatile int* p, int v) {
  const unsigned off = (unsigned)(uintptr_t)p;
  asm volatile("ds_or_b32 %0, %1" : : "v"(off), "v"(v) : "memory");
}
__device__ __forceinline__ void d3_add(volatile int* p, int v) {
  const unsigned off = (unsigned)(uintptr_t)p;
  asm volatile("ds_add_u32 %0, %1" : : "v"(off), "v"(v) : "memory");
}

#define D3_IMG(bi, bj) (S + blk_off(bi, bj))     /* the block's home: register image [r][lane] while it accumulates, [16][17] once it is L */

__device__ __forceinline__ int d3_simd_id() {
  return (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3;     // HW_REG_HW_ID (4), SIMD_ID = bits 5:4
}

// element (r, cc) of the symmetric block (bi, bj) of A as the factor sees it: lower triangle of A, identity beyond nb
__device__ __forceinline__ double d3_elem(const double* __restrict__ A, int lda, int nb, int r, int cc) {
  const int hi = max(r, cc), lo = min(r, cc);
  if (hi >= nb) return (r == cc) ? 1.0 : 0.0;
  return A[(size_t)hi * lda + lo];
}

// stores acc = L[i][s]^T (lane (g, c), register r = L[i][s][c][g + 4r]) into the block's LDS home as L[i][s] and into A
__device__ __forceinline__ void d3_store_LT(double* home, double* __restrict__ A, int lda, int nb, int i, int s, const d4_t& acc, int g, int c) {
#pragma unroll
  for (int r = 0; r < 4; ++r) home[c * BLD + g + 4 * r] = acc[r];
  const int gr = i * 16 + c;
  if (gr < nb) {
    double* dst = A + (size_t)gr * lda + s * 16 + g;
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[4 * r] = acc[r];
  }
}

// the transposed register image of block (i, k) of A straight from global memory: lane (g, c), register r = S[c][g + 4r]
__device__ __forceinline__ void d3_load_image(double (&x)[4], const double* __restrict__ A, int lda, int nb, int i, int k, int g, int c) {
  const int rr = i * 16 + c;
  if (i == k) {
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = d3_elem(A, lda, nb, rr, k * 16 + g + 4 * r);      // mirrored from the lower triangle
  } else {
    const double* src = A + (size_t)min(rr, nb - 1) * lda + k * 16 + g;
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = (rr < nb) ? src[4 * r] : 0.0;
  }
}

// row block s of the inverse, columns hidx and hidx + 6, from registers into the LDS homes of row s of L and into the Dinv store
__device__ __forceinline__ void d3_store_inverse_rows(double* S, double* __restrict__ Dinv, const double (&Xn)[2][4], int s, int hidx, int g, int c) {
#pragma unroll
  for (int q2 = 0; q2 < 2; ++q2) {
    const int j = hidx + q2 * D3_NH;
    if (j >= s) continue;
    double* dst = S + blk_off(s, j);
    double* gd = Dinv + (size_t)(s * 16 + g) * NB + j * 16 + c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dst[(g + 4 * r) * BLD + c] = Xn[q2][r];
      gd[(size_t)4 * r * NB] = Xn[q2][r];
    }
  }
}

__device__ int ffgp_d3_dbg[32];     // state of the flags when a hand-off timed out (development aid)

// members of a ragged launch (blocks of different sizes sharing one chain, ffgp_potrf_ragged): each workgroup's own diagonal block
struct DiagRag {
  double* A[FFGP_RAG_MAX];
  double* Dinv[FFGP_RAG_MAX];
  int lda[FFGP_RAG_MAX];
  int nb[FFGP_RAG_MAX];
  int info[FFGP_RAG_MAX];      // index of the member's status word
  unsigned* pub;               // (every launch, ragged or not) a look-ahead hand-off this kernel publishes as it starts: everything
  unsigned pub_val;            // enqueued before it on its stream is complete then (la_record_deferred); nullptr = none
};

template <bool RAG>
__global__ __launch_bounds__(512, 4) void ffgp_potrf_diag128_v3(double* __restrict__ A, int lda, int nb, double* __restrict__ Dinv,
                                                                int* info, int row_base, int prio, long sA, long sD, int sInfo,
                                                                DiagRag rag) {
  if constexpr (RAG) {
    // (ragged batch: workgroup b factors member b's block -- own matrix, leading dimension, block size, Dinv slot, status word)
    A = rag.A[blockIdx.x];
    Dinv = rag.Dinv[blockIdx.x];
    lda = rag.lda[blockIdx.x];
    nb = rag.nb[blockIdx.x];
    info += rag.info[blockIdx.x];
  } else {
    // (batched factorisation: workgroup b factors block b -- its own matrix, Dinv store and status word)
    A += (size_t)blockIdx.x * sA;
    Dinv += (size_t)blockIdx.x * sD;
    info += blockIdx.x * sInfo;
  }
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* S = lds;
  volatile D3Flags* fl = reinterpret_cast<volatile D3Flags*>(lds + NBLK_LOWER * BLKSZ);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (uniform: scalar control flow)
  const int g = lane >> 4, c = lane & 15;
  // the previous panel is complete the moment this kernel runs (stream order): publish that to the update stream's waiting gate
  if (rag.pub && blockIdx.x == 0 && tid == 64) __hip_atomic_store(rag.pub, rag.pub_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (wave == 0) D2_TRACE(0);
  // wave 0's first operands need no update: it fetches them itself, straight into the registers of F(0) and G(0), and the loads fly
  // while the roles are handed out
  double v[4], sb[4], sb2[4];
  d4_t D;
  if (wave == 0) {
    double x[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = d3_elem(A, lda, nb, g + 4 * r, c);
    d3_load_image(sb, A, lda, nb, 1, 0, g, c);
    d3_load_image(sb2, A, lda, nb, 2, 0, g, c);
    d3_load_image(x, A, lda, nb, 1, 1, g, c);
#pragma unroll
    for (int r = 0; r < 4; ++r) D[r] = x[r];
  }
  // ---- roles: every wave publishes its SIMD; one barrier; the wave that shares wave 0's SIMD steps aside
  // d2_st is inline asm: the compiler's wait-count pass does not see the LDS store inside it and puts NO s_waitcnt in front of the
  // barrier -- a wave could pass the barrier with its store still in flight.  Under load that happened: a wave read simd[w] before
  // wave w's store had landed (0, which matched wave 0's SIMD 0), the waves disagreed about who the partner is, one helper role stayed
  // empty and the others waited for it until the watchdog fired (found by the co-running test leg; only ever seen beside two
  // eigensolvers).  Every flag store that a barrier is meant to publish is therefore drained explicitly.
  if (tid < (int)(sizeof(D3Flags) / sizeof(int))) d2_st(reinterpret_cast<volatile int*>(fl) + tid, 0);
  D2_LDS_FENCE();
  __syncthreads();
  if (lane == 0) d2_st(&fl->simd[wave], d3_simd_id());
  D2_LDS_FENCE();
  __syncthreads();
  int partner = 4;
  {
    int mine;      // lane l < 8 looks at wave l's SIMD: one LDS read for the whole search
    const unsigned off = (unsigned)(uintptr_t)&fl->simd[lane & 7];
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(mine) : "v"(off) : "memory");
    const int s0 = __builtin_amdgcn_readfirstlane(mine);
    const unsigned long long same = __ballot(mine == s0 && lane >= 1 && lane < 8);
    if (same) partner = __ffsll((long long)same) - 1;
  }
  volatile int* ab = &fl->abort;
  if (wave != 0 && wave == partner) return;      // (its SIMD now belongs to the pivot chain alone)
  const int hidx = (wave < partner) ? wave - 1 : wave - 2;

  if (wave == 0) {
    // ================================ the serial chain ================================
    if (prio) __builtin_amdgcn_s_setprio(3);
    double w[4];
    D2_TRACE(1);
    for (int jj = 0; jj < 8; ++jj) {
      // ---- F(jj)
      int cc = c, gg = g;
      asm volatile("" : "+v"(cc), "+v"(gg));      // (opaque per iteration: see v2)
#pragma unroll
      for (int r = 0; r < 4; ++r) w[r] = (gg + 4 * r == cc) ? 1.0 : 0.0;
      double rowA = bperm_d(v[0], cc);
      double rowW = (cc == 0) ? 1.0 : 0.0;
      {
        double hA = bperm_d(v[0], 16 + cc), hW = (cc == 1) ? 1.0 : 0.0;
        double pRow = 0.0, pt = 0.0, ptw = 0.0;
        double dcur = row_bcast64<0>(rowA), ycur = __builtin_amdgcn_rcp(dcur);
#define F16_S(JJ) f16_step_dpp<JJ>(v, w, rowA, rowW, hA, hW, pRow, pt, ptw, dcur, ycur, cc, gg);
        F16_S(0) F16_S(1) F16_S(2) F16_S(3) F16_S(4) F16_S(5) F16_S(6) F16_S(7) F16_S(8) F16_S(9) F16_S(10) F16_S(11) F16_S(12) F16_S(13)
        F16_S(14) F16_S(15)
#undef F16_S
      }
      D2_TRACE(64 + jj);     // the 16 pivots are done
      if (lane == 0) d2_st(&fl->prog[0], 16 * jj + 1);
      // operands of G(jj), register images left by their owners: requested now, under the post-processing below
      if (jj >= 1 && jj < 7) {
        if (!d2_wait_ge(&fl->hs, jj + 1, ab, info, 1) || !d2_wait_ge(&fl->hd, jj + 1, ab, info, 2)) break;
        if (jj < 6 && !d2_wait_ge(&fl->h2, jj + 1, ab, info, 3)) break;
        const double* Sb = D3_IMG(jj + 1, jj);
        const double* Sd = D3_IMG(jj + 1, jj + 1);
        const double* Sb2 = D3_IMG(min(jj + 2, 7), jj);
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) sb[kq] = Sb[kq * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) D[r] = Sd[r * 64 + lane];
        if (jj < 6) {
#pragma unroll
          for (int kq = 0; kq < 4; ++kq) sb2[kq] = Sb2[kq * 64 + lane];
        }
      }
      D2_TRACE(3 + 3 * jj);
      const int q = c >> 2;
      const double dsel = (q == 0) ? v[0] : (q == 1) ? v[1] : (q == 2) ? v[2] : v[3];
      const double dcol = bperm_d(dsel, 16 * (c & 3) + c);
      const double rs = rsqrt_nr(dcol);          // 1 / sqrt(pivot of column c)
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
      // ---- G(jj): the solves of block rows jj + 1 and jj + 2 (both feed the blocks wave 0 needs next, one flag hop away), then
      // the next diagonal block
      d4_t Y = {0.0, 0.0, 0.0, 0.0}, Y2 = {0.0, 0.0, 0.0, 0.0};
      double wa[4];
      D2_TRACE(72 + jj);
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) wa[kq] = Dj[c * BLD + kq * 4 + g];
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) Y = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[kq], sb[kq], Y, 0, 0, 0);
      if (jj < 6) {
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) Y2 = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[kq], sb2[kq], Y2, 0, 0, 0);
      }
      {   // L[jj+1][jj] = Y^T (its image sits in sb), L[jj+2][jj] = Y2^T: to LDS and announced first -- the helpers' next hop
        double* h1 = S + blk_off(jj + 1, jj);
        double* h2 = S + blk_off(min(jj + 2, 7), jj);
#pragma unroll
        for (int r = 0; r < 4; ++r) h1[c * BLD + g + 4 * r] = Y[r];
        if (jj < 6) {
#pragma unroll
          for (int r = 0; r < 4; ++r) h2[c * BLD + g + 4 * r] = Y2[r];
        }
      }
      D2_LDS_FENCE();
      if (lane == 0) d3_or(&fl->rows[jj], (jj < 6) ? (3 << (jj + 1)) : (1 << (jj + 1)));
      D2_TRACE(80 + jj);     // rows jj + 1, jj + 2 of column jj announced
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) D = __builtin_amdgcn_mfma_f64_16x16x4f64(Y[kq], Y[kq], D, 0, 0, 1);   // D -= Y^T Y
      {   // ... and to global memory while those products run
        const int gr = (jj + 1) * 16 + c;
        if (gr < nb) {
          double* dst = A + (size_t)gr * lda + jj * 16 + g;
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[4 * r] = Y[r];
        }
        if (jj < 6 && gr + 16 < nb) {
          double* dst = A + (size_t)(gr + 16) * lda + jj * 16 + g;
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[4 * r] = Y2[r];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = D[r];
      D2_TRACE(4 + 3 * jj);
    }
    return;
  }

  // ================================ helpers ================================
  // ---- prologue: the owned blocks come from global memory as transposed register images, all loads in flight at once
  // (one block after the other was one memory round trip each: 6.5 us before the first stage).  (1, 0), (1, 1), (2, 0) are wave 0's.
  {
    double x[6][4];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int t = hidx + u * D3_NH;                  // enumeration index -> (i, k)
      int k = 0, i = t + 1;
#pragma unroll
      for (int kk = 1; kk < 8; ++kk) {
        const int t0 = 8 * kk - 1 - kk * (kk - 1) / 2;
        if (t >= t0) { k = kk; i = kk + (t - t0); }
      }
      if (t > 34 || (k == 0 && i <= 2) || (k == 1 && i == 1)) continue;
      d3_load_image(x[u], A, lda, nb, i, k, g, c);
    }
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int t = hidx + u * D3_NH;
      int k = 0, i = t + 1;
#pragma unroll
      for (int kk = 1; kk < 8; ++kk) {
        const int t0 = 8 * kk - 1 - kk * (kk - 1) / 2;
        if (t >= t0) { k = kk; i = kk + (t - t0); }
      }
      if (t > 34 || (k == 0 && i <= 2) || (k == 1 && i == 1)) continue;
      double* img = D3_IMG(i, k);
#pragma unroll
      for (int r = 0; r < 4; ++r) img[r * 64 + lane] = x[u][r];
    }
    D2_LDS_FENCE();
  }
  if (hidx == 0) D2_TRACE(31);
  const int my_desc = D3_TASKS.t[hidx][lane];  // lane t holds descriptor t of this helper's list
  double Xn[2][4];                             // the row block of the inverse computed in a stage, stored once row s of L is dead
  const double* Ws = S;                        // inv(L_s)
  volatile int* rows = &fl->rows[0];           // rows[s], bit i: L[i][s] is final and in LDS
  int seen = 0;                                // lane p: the bits of rows[p] this wave has seen set (one poll serves every later task)
#pragma unroll 1
  for (int t = 0; t < D3_MAXTASKS; ++t) {
    const int desc = __builtin_amdgcn_readlane(my_desc, t);
    if (lane == 0) d2_st(&fl->prog[1 + hidx], 1 + t);      // (post-mortem of a timed-out hand-off: where every helper stood)
    if (desc == D3_END) break;
    const int type = desc & 3, i = (desc >> 2) & 7, k = (desc >> 5) & 7, flag = (desc >> 8) & 3, s = (desc >> 10) & 7;
    if (type == D3_STAGE) {
      if (hidx == 0 && s >= 1) D2_TRACE(31 + 4 * s);          // (slot 35 + 4 (s - 1): the previous stage is complete)
      if (!d2_wait_ge(&fl->seqF, s + 1, ab, info, 4)) return;
      if (hidx == 0) D2_TRACE(32 + 4 * s);
      Ws = S + blk_off(s, s);
      rows = &fl->rows[s];
    } else if (type == D3_TRSM) {
      // ---- triangular solve of an owned block of column s (its updates were completed in the earlier stages)
      double* home = D3_IMG(i, s);
      double b[4];
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) b[kq] = home[kq * 64 + lane];
      d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ws[c * BLD + kq * 4 + g], b[kq], acc, 0, 0, 0);
      d3_store_LT(home, A, lda, nb, i, s, acc, g, c);
      D2_LDS_FENCE();
      if (lane == 0) d3_or(rows, 1 << i);
    } else if (type == D3_UPDATE) {
      // ---- columns p0 .. s of L update a later block this wave owns:  image -= L[k][p] L[i][p]^T  (the transposed update)
      const int p0 = (desc >> 13) & 7;
      const int need = (1 << i) | (1 << k);
      double* img = D3_IMG(i, k);
      d4_t acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = img[r * 64 + lane];
#pragma unroll 1
      for (int p = p0; p <= s; ++p) {
        // rows i and k of column p must be final: lane p of `seen` remembers what this wave saw of rows[p] (bits only ever get set)
        int have = __builtin_amdgcn_readlane(seen, p);
        if ((have & need) != need) {
          int spins = 0;
          while (((have = d2_ld(&fl->rows[p])) & need) != need) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 63) == 0 && (spins > (1 << 21) || d2_ld(ab))) {
              if (!d2_ld(ab)) {
                atomicExch(info, FFGP_DIAG_WATCHDOG + 5);
                if (lane == 0) {     // post-mortem: which hand-off never came (printed by ffgp_map_info)
                  ffgp_d3_dbg[0] = hidx; ffgp_d3_dbg[1] = s; ffgp_d3_dbg[2] = p; ffgp_d3_dbg[3] = i; ffgp_d3_dbg[4] = k; ffgp_d3_dbg[5] = have;
                  ffgp_d3_dbg[6] = d2_ld(&fl->seqF); ffgp_d3_dbg[7] = d2_ld(&fl->hs); ffgp_d3_dbg[8] = d2_ld(&fl->hd); ffgp_d3_dbg[9] = d2_ld(&fl->h2);
                  for (int z = 0; z < 8; ++z) ffgp_d3_dbg[10 + z] = d2_ld(&fl->rows[z]);
                  for (int z = 0; z < 8; ++z) ffgp_d3_dbg[18 + z] = d2_ld(&fl->cntA[z]);
                  ffgp_d3_dbg[26] = t;
                  for (int z = 0; z < 5; ++z) ffgp_d3_dbg[27 + z] = d2_ld(&fl->prog[z]) | (d2_ld(&fl->prog[z + (z < 3 ? 5 : 0)]) << 16);
                }
              }
              d2_st(ab, 1);
              return;
            }
          }
          D2_COMPILER_FENCE();
          seen = (lane == p) ? have : seen;
        }
        const double* Lk = S + blk_off(k, p);
        const double* Li = S + blk_off(i, p);
#pragma unroll
        for (int kq = 0; kq < 4; ++kq)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Lk[c * BLD + kq * 4 + g], Li[c * BLD + kq * 4 + g], acc, 0, 0, 1);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) img[r * 64 + lane] = acc[r];
      if (flag) {                              // complete as far as its owner is concerned: an operand of G(s + 1)
        D2_LDS_FENCE();
        if (lane == 0) d2_st((flag == 1) ? &fl->hs : (flag == 2) ? &fl->hd : &fl->h2, s + 2);
      }
    } else {
      // ---- row block s of the inverse, columns j = hidx and hidx + 6:  X[s][j] = -inv(L_s) T_j,  T_j = sum_{k=j}^{s-1} L[s][k] X[k][j].
      // It overwrites row s of L in place, so it may only be stored once all six helpers have finished reading that row (cntA[s]):
      // the values stay in registers for one stage and are stored at the START of the next stage's inverse task -- a helper that
      // waited here for the slowest one was late for the next column's urgent updates.
      if (hidx == 0) D2_TRACE(34 + 4 * s);
      if (s >= 2) {
        if (!d2_wait_ge(&fl->cntA[s - 1], D3_NH, ab, info, 6)) return;
        d3_store_inverse_rows(S, Dinv, Xn, s - 1, hidx, g, c);
        D2_LDS_FENCE();
      }
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const int j = hidx + q2 * D3_NH;
        if (j >= s) continue;
        d4_t T = {0.0, 0.0, 0.0, 0.0};
        for (int kk = j; kk < s; ++kk) mma16<false>(T, S + blk_off(s, kk), BLD, S + blk_off(kk, j), BLD, lane);
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ws[c * BLD + kq * 4 + g], T[kq], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) Xn[q2][r] = -acc[r];
      }
      // every read of row s of L by this wave is behind it (the updates of column s's blocks ran in earlier stages)
      D2_LDS_FENCE();
      if (lane == 0) d3_add(&fl->cntA[s], 1);
      if (s == 7) {                            // the last row block: nothing follows, store it now
        if (!d2_wait_ge(&fl->cntA[7], D3_NH, ab, info, 7)) return;
        d3_store_inverse_rows(S, Dinv, Xn, 7, hidx, g, c);
      }
    }
  }
  if (hidx == 0) D2_TRACE(63);
}

// ------------------------------------------------------------------------------------------------------------
// potrf_diag128_v4 (round 6): the stage loop of the one-launch trainer (train.hip) as the diagonal-block kernel.
//
// v3 is a flag-driven pipeline: wave 0 carries the pivot chain AND the solves of the next two block rows AND the next diagonal
// block's update (G: 12 dependent MFMAs and two LDS round trips per stage) while six helpers follow task lists behind LDS flags -- 3.5 us
// per 16 columns, co-limited between wave 0 and the helpers.  Here every stage has TWO workgroup barriers and nothing else to wait for:
//   [A] wave 0 applies column jj - 1 to its diagonal block on the way into the registers and factors + inverts it (the same DP-ALU DPP
//       pivot step); in its shadow the helper waves (those that do not share wave 0's SIMD) apply column jj - 1 to every other block and
//       compute row block jj - 1 of the inverse into registers;
//   [B] the inverse's row is stored (LDS + Dinv store), all eight waves solve column jj with inv(L_jj) (one block each) and write L.
// Only ceil(nb / 16) stages run.  25.2 us per 128-block alone against v3's 28.6 (rocprofv3 kernel trace, profiles/r06g_*).  A variant with
// ONE workgroup barrier per stage -- wave 0 solving block row jj itself in v3's way (Y = inv(L) S^T, D -= Y^T Y in registers) and the
// helpers meeting at an LDS counter that wave 0 only arrives at -- was built and measured equal (docs/experiments.md, round 6).  The factor differs from v3's in the last bits (the updates reach a
// block in a different order); both are held to LAPACK (1e-11) by the same tests.
// ------------------------------------------------------------------------------------------------------------
#define DIAG4_LDS_DOUBLES (NBLK_LOWER * BLKSZ + 128 + 16)
#ifndef FFGP_D4_DBG
#define FFGP_D4_DBG 0      // timing-only ablations of v4 (results wrong when non-zero): 1 = helpers idle in [A], 2 = wave 0 skips its pivots
#endif
#define DIAG4_LDS_BYTES (DIAG4_LDS_DOUBLES * 8)

// (workgroup barrier that publishes LDS only: the kernel's global stores -- L, the Dinv store -- are read by nobody inside it, and
//  __syncthreads() would expose their round trip to L2 at every one of the 2 x 8 barriers)
#define D4_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <bool RAG>
__global__ __launch_bounds__(512, 2) void ffgp_potrf_diag128_v4(double* __restrict__ A, int lda, int nb, double* __restrict__ Dinv,
                                                                int* info, int row_base, int prio, long sA, long sD, int sInfo,
                                                                DiagRag rag) {
  if constexpr (RAG) {
    A = rag.A[blockIdx.x];
    Dinv = rag.Dinv[blockIdx.x];
    lda = rag.lda[blockIdx.x];
    nb = rag.nb[blockIdx.x];
    info += rag.info[blockIdx.x];
  } else {
    A += (size_t)blockIdx.x * sA;
    Dinv += (size_t)blockIdx.x * sD;
    info += blockIdx.x * sInfo;
  }
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* S = lds;
  int* flags = reinterpret_cast<int*>(lds + NBLK_LOWER * BLKSZ + 128);      // [0..7] SIMD of wave w
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  if (rag.pub && blockIdx.x == 0 && tid == 64) __hip_atomic_store(rag.pub, rag.pub_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (prio) __builtin_amdgcn_s_setprio(3);
  const int nst = (nb + 15) >> 4;
  if (lane == 0) flags[wave] = d3_simd_id();
  D4_BARRIER();
  int hidx = -1, nh = 0;      // helpers: the waves that do not share wave 0's SIMD (fp64 MFMAs and the pivot loop's DP-ALU work share a pipe)
  {
    const int s0 = flags[0];
    for (int w_ = 1; w_ < 8; ++w_) {
      const bool is_h = flags[w_] != s0;
      if (is_h && w_ == wave) hidx = nh;
      nh += is_h ? 1 : 0;
    }
    if (nh == 0) { nh = 7; hidx = wave - 1; }
    hidx = __builtin_amdgcn_readfirstlane(hidx);
    nh = __builtin_amdgcn_readfirstlane(nh);
  }
  // ---- loads: wave 0 takes block (0, 0) straight into the factor's registers; the other waves bring every other lower block of the
  //      first nst block rows into LDS ([16][17] images; diagonal blocks mirrored to full), all loads of a wave in flight at once
  double v0[4];
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v0[r] = d3_elem(A, lda, nb, g + 4 * r, c);
  } else {
    const int nblk = nst * (nst + 1) / 2;
    double x[6][4];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int t = wave + 7 * u;      // blocks 1 .. nblk - 1 of the row-major enumeration over seven waves
      if (t >= nblk) continue;
      int bi = 0;
#pragma unroll
      for (int q = 1; q < 8; ++q) bi += (t >= q * (q + 1) / 2) ? 1 : 0;
      const int bj = t - bi * (bi + 1) / 2;
#pragma unroll
      for (int r = 0; r < 4; ++r) x[u][r] = d3_elem(A, lda, nb, bi * 16 + g + 4 * r, bj * 16 + c);
    }
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int t = wave + 7 * u;
      if (t >= nblk) continue;
      int bi = 0;
#pragma unroll
      for (int q = 1; q < 8; ++q) bi += (t >= q * (q + 1) / 2) ? 1 : 0;
      const int bj = t - bi * (bi + 1) / 2;
      double* dst = S + blk_off(bi, bj);
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(g + 4 * r) * BLD + c] = x[u][r];
    }
    // rows beyond the stages that run: identity in the Dinv store (a previous, larger block may have left its inverse there)
    for (int t = nst * (nst + 1) / 2 + (wave - 1); t < NBLK_LOWER; t += 7) {
      int bi = 0;
#pragma unroll
      for (int q = 1; q < 8; ++q) bi += (t >= q * (q + 1) / 2) ? 1 : 0;
      const int bj = t - bi * (bi + 1) / 2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = g + 4 * r;
        if (bi != bj || i >= c) Dinv[(size_t)(bi * 16 + i) * NB + bj * 16 + c] = (bi == bj && i == c) ? 1.0 : 0.0;
      }
    }
  }
  d4_t Tl[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
  for (int jj = 0; jj < nst; ++jj) {
    d4_t Xn[2];
    if (wave == 0) {
      double* Dj = S + blk_off(jj, jj);
      d4_t upd = {0.0, 0.0, 0.0, 0.0};
      if (jj > 0) mma16<true>(upd, S + blk_off(jj, jj - 1), BLD, S + blk_off(jj, jj - 1), BLD, lane);
      int cc = c, gg = g;
      asm volatile("" : "+v"(cc), "+v"(gg));      // (opaque per iteration: see v2)
      double v[4], w[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = (jj == 0) ? v0[r] : Dj[(gg + 4 * r) * BLD + cc] - upd[r];
        w[r] = (gg + 4 * r == cc) ? 1.0 : 0.0;
      }
      double rowA = bperm_d(v[0], cc);
      double rowW = (cc == 0) ? 1.0 : 0.0;
      {
        double hA = bperm_d(v[0], 16 + cc), hW = (cc == 1) ? 1.0 : 0.0;
        double pRow = 0.0, pt = 0.0, ptw = 0.0;
        double dcur = row_bcast64<0>(rowA), ycur = __builtin_amdgcn_rcp(dcur);
#define F16_S(JJ) f16_step_dpp<JJ>(v, w, rowA, rowW, hA, hW, pRow, pt, ptw, dcur, ycur, cc, gg);
        if (!(FFGP_D4_DBG & 2)) {
        F16_S(0) F16_S(1) F16_S(2) F16_S(3) F16_S(4) F16_S(5) F16_S(6) F16_S(7) F16_S(8) F16_S(9) F16_S(10) F16_S(11) F16_S(12) F16_S(13)
        F16_S(14) F16_S(15)
        }
#undef F16_S
      }
      const int q = c >> 2;
      const double dsel = (q == 0) ? v[0] : (q == 1) ? v[1] : (q == 2) ? v[2] : v[3];
      const double dcol = bperm_d(dsel, 16 * (c & 3) + c);
      const double rs = rsqrt_nr(dcol);
      const unsigned long long nonpos = __ballot(!(dcol > 0.0)) & 0xffffull;
      const int bad = nonpos ? __ffsll((long long)nonpos) : 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = g + 4 * r;
        const double rsi = bperm_d(rs, i);
        const double x = (i >= c) ? w[r] * rsi : 0.0;
        Dj[i * BLD + c] = x;                                              // inv(L_jj)
        const int gr = jj * 16 + i, gc = jj * 16 + c;
        if (i >= c) {
          Dinv[(size_t)gr * NB + gc] = x;
          if (gr < nb) A[(size_t)gr * lda + gc] = v[r] * rs;              // L_jj
        }
      }
      if (bad && lane == 0 && (jj * 16 + bad) <= nb) atomicCAS(info, 0, row_base + jj * 16 + bad);
    } else if (hidx >= 0 && jj > 0 && !(FFGP_D4_DBG & 1)) {
      // column jj - 1 reaches every block (i, k), jj <= k <= i, but (jj, jj).  (Two tasks at a time -- both tasks' operands requested before
      // either product, the two MFMA chains alternating -- was measured SLOWER: 25.9 against 25.2 us per block.  By ablation
      // (-DFFGP_D4_DBG) the kernel is co-limited: without wave 0's pivots it takes the same time, with idle helpers 2.9 us less.)
      const int m = nst - jj;
      for (int t = 1 + hidx; t < m * (m + 1) / 2; t += nh) {
        int a = 0;
#pragma unroll
        for (int q = 1; q < 8; ++q) a += (t >= q * (q + 1) / 2) ? 1 : 0;
        const int b = t - a * (a + 1) / 2;
        const int i = jj + a, k = jj + b;
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
        mma16<true>(acc, S + blk_off(i, jj - 1), BLD, S + blk_off(k, jj - 1), BLD, lane);
        double* dst = S + blk_off(i, k);
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(g + 4 * r) * BLD + c] -= acc[r];
      }
      const int s_ = jj - 1;      // row block s_ of the inverse, columns hidx and hidx + nh: X = -inv(L_s) sum_k L[s][k] X[k][j]
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const int j = hidx + q2 * nh;
        if (j >= s_) continue;
        d4_t T = {0.0, 0.0, 0.0, 0.0};
        for (int k = j; k < s_; ++k) mma16<false>(T, S + blk_off(s_, k), BLD, S + blk_off(k, j), BLD, lane);
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
        const double* Ws = S + blk_off(s_, s_);
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ws[c * BLD + kq * 4 + g], T[kq], acc, 0, 0, 0);
        Xn[q2] = acc;
      }
      if (jj == nst - 1) {      // last stage: the terms k <= s - 2 of the inverse's LAST row as well (only k = s - 1 is left for the tail)
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          const int j = hidx + q2 * nh;
          d4_t T = {0.0, 0.0, 0.0, 0.0};
          if (j < jj - 1)
            for (int k = j; k < jj - 1; ++k) mma16<false>(T, S + blk_off(jj, k), BLD, S + blk_off(k, j), BLD, lane);
          Tl[q2] = T;
        }
      }
    }
    D4_BARRIER();
    if (hidx >= 0 && jj > 0) {      // row jj - 1 of L is dead now: the inverse's row takes its place (and goes to the Dinv store)
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const int j = hidx + q2 * nh;
        if (j >= jj - 1) continue;
        double* dst = S + blk_off(jj - 1, j);
        double* gd = Dinv + (size_t)((jj - 1) * 16 + g) * NB + j * 16 + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dst[(g + 4 * r) * BLD + c] = -Xn[q2][r];
          gd[(size_t)4 * r * NB] = -Xn[q2][r];
        }
      }
    }
    for (int i = jj + 1 + wave; i < nst; i += 8) {      // solve: L[i][jj] = S[i][jj] inv(L_jj)^T
      d4_t acc = {0.0, 0.0, 0.0, 0.0};
      mma16<true>(acc, S + blk_off(i, jj), BLD, S + blk_off(jj, jj), BLD, lane);
      double* dst = S + blk_off(i, jj);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dst[(g + 4 * r) * BLD + c] = acc[r];
        const int gr = i * 16 + g + 4 * r;
        if (gr < nb) A[(size_t)gr * lda + jj * 16 + c] = acc[r];
      }
    }
    D4_BARRIER();
  }
  if (nst > 1 && hidx >= 0) {      // the last row block of the inverse: the helpers' columns, one product left per column
    const int s_ = nst - 1;
    const double* Ws = S + blk_off(s_, s_);
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      const int j = hidx + q2 * nh;
      if (j >= s_) continue;
      d4_t T = Tl[q2];
      mma16<false>(T, S + blk_off(s_, s_ - 1), BLD, S + blk_off(s_ - 1, j), BLD, lane);      // (block (s - 1, s - 1) holds inv(L_{s-1}) = X[s-1][s-1])
      d4_t X = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) X = __builtin_amdgcn_mfma_f64_16x16x4f64(Ws[c * BLD + kq * 4 + g], T[kq], X, 0, 0, 0);
      double* gd = Dinv + (size_t)(s_ * 16 + g) * NB + j * 16 + c;
#pragma unroll
      for (int r = 0; r < 4; ++r) gd[(size_t)4 * r * NB] = -X[r];
    }
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
// The chain's TRSM as its own kernel (round 5): X = A21 * Dinv^T in place for one full 128-column block.
// The general GEMM spends 128 MFMAs per wave on a 32 x 128 tile of this product -- half of them on Dinv's zero upper triangle -- behind
// eight k-tile barriers and a 40 KiB LDS stage.  Here a workgroup owns 16 rows: they go to LDS once (one barrier) and from there into
// registers as MFMA operands in one batch; wave w owns the 16-column blocks w and 7 - w of the output (4 (w + 1) + 4 (8 - w) = 36 MFMA
// k-steps for every wave: the triangle is balanced across the four waves) and fetches those blocks' rows of Dinv straight from L2 into
// registers (no LDS image of Dinv: the kernel must fit beside a resident trailing-update workgroup; 17 KiB of LDS does).
// The values are the general GEMM's bit for bit: per output element the same v_mfma_f64_16x16x4 chain over k = 0, 4, 8, ... from a zero
// accumulator, minus the k-steps in which every Dinv operand is a structural zero (adding 0 * a changes nothing) -- so this kernel and
// the general one are interchangeable launch by launch and member by member: partial last blocks, odd leading dimensions and very tall
// panels stay on the general GEMM, and a shared chain (ffgp_nlml_fused_batch) still equals the single call whichever kernel each took.
// Measured (profiles/r05e_*): 7.7 us per launch against 11.0 (an empty dependent launch with the same stores: 5.0); forward at
// N = 1024 / 2048 / 4096 / 8192: -5.6 / -5.4 / -4.8 / -1.6 %.
// ------------------------------------------------------------------------------------------------------------
#define TRSM_LDS_LD 130   // doubles per LDS row: 2 mod 32 -> the 16 rows x 2 k-values a 32-lane half reads hit 32 distinct 8-byte banks

// One 16-byte load per lane fetches the k-pair {8p + 2kk, 8p + 2kk + 1} of the lane's row (kk = lane / 16: 64 contiguous bytes per row and
// instruction -- half the cache-line requests of one 8-byte load per MFMA k-step); the MFMA operand of k-step 2p wants k = 8p + kk in lane
// group kk, that of k-step 2p + 1 wants 8p + 4 + kk.  Two of gfx950's row swaps per 32-bit half re-deal the four 16-lane rows:
// v_permlane16_swap (x.row1 <-> y.row0, x.row3 <-> y.row2), then v_permlane32_swap (x.rows23 <-> y.rows01).
__device__ __forceinline__ void kpair_to_ksteps(d2_t in, double& u, double& v) {
  const unsigned long long xb = __builtin_bit_cast(unsigned long long, (double)in.x), yb = __builtin_bit_cast(unsigned long long, (double)in.y);
  auto l1 = __builtin_amdgcn_permlane16_swap((unsigned)xb, (unsigned)yb, false, false);
  auto l2 = __builtin_amdgcn_permlane32_swap(l1[0], l1[1], false, false);
  auto h1 = __builtin_amdgcn_permlane16_swap((unsigned)(xb >> 32), (unsigned)(yb >> 32), false, false);
  auto h2 = __builtin_amdgcn_permlane32_swap(h1[0], h1[1], false, false);
  u = __builtin_bit_cast(double, ((unsigned long long)h2[0] << 32) | l2[0]);
  v = __builtin_bit_cast(double, ((unsigned long long)h2[1] << 32) | l2[1]);
}

// members of one launch (blockIdx.y): MEMB = 0 one matrix or a batch at fixed strides (ffgp_nlml_fused_batch's uniform chain), MEMB = 1 up
// to FFGP_RAG_MAX matrices of different sizes from a by-value table (the ragged chain; a member's surplus workgroups exit at once)
struct TrsmSet {
  double* A[FFGP_RAG_MAX];
  const double* D[FFGP_RAG_MAX];
  int lda[FFGP_RAG_MAX];
  int mrows[FFGP_RAG_MAX];
};

template <int MEMB>
__global__ __launch_bounds__(256) void ffgp_trsm128_kernel(double* __restrict__ A, int lda, int mrows, const double* __restrict__ Dinv,
                                                           int prio, long sA_, long sD_, TrsmSet set) {
  __shared__ __attribute__((aligned(16))) double sA[16 * TRSM_LDS_LD];
  if (prio) __builtin_amdgcn_s_setprio(2);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = blockIdx.x * 16;
  if (MEMB) {
    const int f = blockIdx.y;
    A = set.A[f];
    Dinv = set.D[f];
    lda = set.lda[f];
    mrows = set.mrows[f];
    if (r0 >= mrows) return;
  } else {
    A += (size_t)blockIdx.y * sA_;
    Dinv += (size_t)blockIdx.y * sD_;
  }
  d2_t va[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i, row = idx >> 6, c2 = idx & 63;
    const int gr = r0 + row;
    va[i] = (d2_t){0.0, 0.0};
    if (gr < mrows) va[i] = *reinterpret_cast<const d2_t*>(A + (size_t)gr * lda + 2 * c2);
  }
  const int bL = wave, bH = 7 - wave;
  const int j = lane & 15, kk = lane >> 4;
  const double* dH = Dinv + (size_t)(16 * bH + j) * NB + 2 * kk;
  const double* dL = Dinv + (size_t)(16 * bL + j) * NB + 2 * kk;
  d2_t rh[16], rl[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c <= bH) {
      rh[2 * c] = *reinterpret_cast<const d2_t*>(dH + 16 * c);
      rh[2 * c + 1] = *reinterpret_cast<const d2_t*>(dH + 16 * c + 8);
    }
    if (c < 4 && c <= bL) {
      rl[2 * c] = *reinterpret_cast<const d2_t*>(dL + 16 * c);
      rl[2 * c + 1] = *reinterpret_cast<const d2_t*>(dL + 16 * c + 8);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i, row = idx >> 6, c2 = idx & 63;
    *reinterpret_cast<d2_t*>(sA + row * TRSM_LDS_LD + 2 * c2) = va[i];
  }
  __syncthreads();
  // every A operand of the wave's k range into registers in one batch (an LDS round trip per MFMA would double the chain)
  const double* aP = sA + j * TRSM_LDS_LD + kk;
  double a[32];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c <= bH) {
#pragma unroll
      for (int q = 0; q < 4; ++q) a[4 * c + q] = aP[16 * c + 4 * q];
    }
  }
  d4_t accL = {0.0, 0.0, 0.0, 0.0}, accH = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c <= bH) {
      double bh[4], bl[4];
      kpair_to_ksteps(rh[2 * c], bh[0], bh[1]);
      kpair_to_ksteps(rh[2 * c + 1], bh[2], bh[3]);
      const bool low = (c < 4 && c <= bL);
      if (low) {
        kpair_to_ksteps(rl[(2 * c) & 7], bl[0], bl[1]);
        kpair_to_ksteps(rl[(2 * c + 1) & 7], bl[2], bl[3]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        accH = __builtin_amdgcn_mfma_f64_16x16x4f64(a[4 * c + q], bh[q], accH, 0, 0, 0);
        if (low) accL = __builtin_amdgcn_mfma_f64_16x16x4f64(a[4 * c + q], bl[q], accL, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = r0 + kk + 4 * r;
    if (row < mrows) {
      double* out = A + (size_t)row * lda + j;
      out[16 * bL] = accL[r];
      out[16 * bH] = accH[r];
    }
  }
}

// the dedicated kernel takes full 128-column blocks with 16-byte aligned rows; very tall panels stay on the general GEMM (from ~8000 rows
// -- of all members of a shared chain's launch together -- the 16-row workgroups outnumber the chip's slots twice over and the launch is
// no longer latency-bound: measured 0.5 % slower at C3, 1.1 % slower on 8 chained blocks of N = 8192)
static bool trsm128_fits(const ffgp_handle* h, const double* A21, int lda, int jb, int mrows) {
  return h->trsm128 && jb == NB && mrows <= h->trsm128_max_m && !(lda & 1) && ((uintptr_t)A21 & 15) == 0;
}

static int launch_trsm128(ffgp_handle* h, double* A21, int lda, int mrows, const double* Dj) {
  const int prio = (h->stream == h->aux && h->aux_prio) ? 1 : 0;
  const int F = h->bt_F > 1 ? h->bt_F : 1;
  hipLaunchKernelGGL(ffgp_trsm128_kernel<0>, dim3((mrows + 15) / 16, F), dim3(256), 0, h->stream, A21, lda, mrows, Dj, prio,
                     F > 1 ? h->bt_sA : 0L, F > 1 ? h->bt_sD : 0L, TrsmSet());
  return hipGetLastError() == hipSuccess ? FFGP_OK : FFGP_ERR_HIP;
}

static int launch_trsm128_set(ffgp_handle* h, const TrsmSet& set, int cnt, int max_rows) {
  const int prio = (h->stream == h->aux && h->aux_prio) ? 1 : 0;
  hipLaunchKernelGGL(ffgp_trsm128_kernel<1>, dim3((max_rows + 15) / 16, cnt), dim3(256), 0, h->stream, (double*)nullptr, 0, 0,
                     (const double*)nullptr, prio, 0L, 0L, set);
  return hipGetLastError() == hipSuccess ? FFGP_OK : FFGP_ERR_HIP;
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
static inline void la_take_deferred(ffgp_handle* h, DiagRag& dr);
static int launch_diag(ffgp_handle* h, double* Ablk, int lda, int nb, double* Dinv_blk, int row_base, int do_factor) {
  if (h->bt_F > 1 && !(do_factor && h->diag_v2 == 4 && !h->diag_dbg)) return FFGP_ERR_ARG;   // only the round-4 kernel is batched
  if (!(h->diag_attr_set & 1)) {   // per handle = per device (the attribute lives in the device's context)
    FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_potrf_diag128),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, DIAG_LDS_BYTES));
    h->diag_attr_set |= 1;
  }
  if (do_factor && h->diag_v2 && !h->diag_dbg) {   // the pipelined kernel (the barrier version keeps the inverse-only entry)
#ifdef FFGP_DEV_OPTIONS
    if (!(h->diag_attr_set & 2)) {
      FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_potrf_diag128_v2<8, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, DIAG_LDS_BYTES));
      FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_potrf_diag128_v2<8, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, DIAG_LDS_BYTES));
      h->diag_attr_set |= 2;
    }
#endif
    if (h->diag_v2 == 4) {   // round 4: owner-computes helpers, wave 0's SIMD partner steps aside
      if (!(h->diag_attr_set & 4)) {
        FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_potrf_diag128_v3<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     DIAG_LDS_BYTES));
        FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_potrf_diag128_v3<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     DIAG_LDS_BYTES));
        h->diag_attr_set |= 4;
      }
      // (diag_excl: the panel's FIRST diagonal block of a chain-bound carry iteration asks for a whole CU's LDS, so the S_bz workgroups that
      //  start the moment it publishes cannot land beside it -- see ffgp_potrf_impl.  Only where a workgroup may have that much: the
      //  limit is asked for once, and a device or runtime that says less simply keeps the ordinary launch)
      if (!h->lds_cap_known) {
        int cap = 0;
        if (hipDeviceGetAttribute(&cap, hipDeviceAttributeMaxSharedMemoryPerBlock, h->device) != hipSuccess) {
          (void)hipGetLastError();
          cap = 64 * 1024;
        }
        h->lds_cap = cap;
        h->lds_cap_known = 1;
      }
      const bool excl = h->diag_excl_now && h->bt_F <= 1 && h->lds_cap >= 160 * 1024;
      const int lds_bytes = excl ? 160 * 1024 : DIAG_LDS_BYTES;
      if (excl && !h->diag_v4 && !(h->diag_attr_set & 8)) {
        FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_potrf_diag128_v3<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     160 * 1024));
        h->diag_attr_set |= 8;
      }
      if (h->diag_v4 && !(h->diag_attr_set & 16)) {
        const int want = h->lds_cap >= 160 * 1024 ? 160 * 1024 : DIAG4_LDS_BYTES;
        FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_potrf_diag128_v4<false>), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_potrf_diag128_v4<true>), hipFuncAttributeMaxDynamicSharedMemorySize, want));
        h->diag_attr_set |= 16;
      }
      // the pending "panel complete" publication is taken only now, when nothing can fail between here and the launch that carries it
      DiagRag dr = DiagRag();
      la_take_deferred(h, dr);
      if (h->diag_v4) {      // round 6: two barriers per stage (ffgp_potrf_diag128_v4)
        const int lds4 = excl ? lds_bytes : DIAG4_LDS_BYTES;
        auto kern = ffgp_potrf_diag128_v4<false>;
        if (h->bt_F > 1)
          hipLaunchKernelGGL(kern, dim3(h->bt_F), dim3(512), DIAG4_LDS_BYTES, h->stream, Ablk, lda, nb, Dinv_blk,
                             h->bt_info, row_base, h->aux_prio, h->bt_sA, h->bt_sD, 1, dr);
        else
          hipLaunchKernelGGL(kern, dim3(1), dim3(512), lds4, h->stream, Ablk, lda, nb, Dinv_blk, h->d_info,
                             row_base, h->aux_prio, 0L, 0L, 0, dr);
      } else if (h->bt_F > 1)
        hipLaunchKernelGGL(ffgp_potrf_diag128_v3<false>, dim3(h->bt_F), dim3(512), DIAG_LDS_BYTES, h->stream, Ablk, lda, nb, Dinv_blk,
                           h->bt_info, row_base, h->aux_prio, h->bt_sA, h->bt_sD, 1, dr);
      else
        hipLaunchKernelGGL(ffgp_potrf_diag128_v3<false>, dim3(1), dim3(512), lds_bytes, h->stream, Ablk, lda, nb, Dinv_blk, h->d_info,
                           row_base, h->aux_prio, 0L, 0L, 0, dr);
      if (hipGetLastError() != hipSuccess) {      // the launch that carried the publication did not happen: write it plainly, report
        if (dr.pub) (void)hipStreamWriteValue32(h->stream, dr.pub, dr.pub_val, 0);
        return FFGP_ERR_HIP;
      }
    }
#ifdef FFGP_DEV_OPTIONS
    else if (h->diag_v2 == 3)   // the round-3 pivot step (32-bit DPP moves), kept for A/B runs
      hipLaunchKernelGGL((ffgp_potrf_diag128_v2<8, false>), dim3(1), dim3(512), DIAG_LDS_BYTES, h->stream, Ablk, lda, nb, Dinv_blk,
                         h->d_info, row_base, h->aux_prio);
    else
      hipLaunchKernelGGL((ffgp_potrf_diag128_v2<8, true>), dim3(1), dim3(512), DIAG_LDS_BYTES, h->stream, Ablk, lda, nb, Dinv_blk,
                         h->d_info, row_base, h->aux_prio);
#endif
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

// ------------------------------------------------------------------------------------------------------------
// Cross-stream hand-offs of the look-ahead (round 5).  hipEventRecord + hipStreamWaitEvent between two hardware queues costs
// 10.7-11.1 us over the in-stream kernel boundary on this runtime; hipStreamWriteValue32 behind the producer + hipStreamWaitValue32
// in front of the consumer -- two command-processor packets on a word of device memory -- cost 2.9-4.7 us, with the producer's
// stores visible to a consumer on another XCD (tools/native/handoff_probe.hip: 200 alternating kernels, every value checked).
// Each of the handle's ten look-ahead events has a word and a host-side sequence number: "record" writes the next number behind the
// producer's work, "wait" waits for the number of the latest record (>=: the numbers only grow) -- the semantics the event pair had.
// A never-recorded hand-off waits for 0 and passes.  Events are kept while a stream is being captured into a graph (the value
// operations are not capturable), with option ho_values = 0, and for every event that is not one of the ten.
// ------------------------------------------------------------------------------------------------------------
static inline int la_slot(const ffgp_handle* h, hipEvent_t ev) {
  if (!h->ho_active) return -1;
  for (int i = 0; i < 10; ++i)
    if (h->la_ev[i] == ev) return i;
  return -1;
}
// The wait is the library's OWN one-wave kernel (round 6): hipStreamWaitValue32 is a polling kernel of the runtime with no timeout -- in a
// process whose dispatches are serialised by a tool that is not on default_ho_values()'s list (api.hip) it would spin for a producer
// that can never start, and the GPU hangs.  This gate polls the same word (measured at the same cost per hop: 3.5-4.0 us against the
// runtime's 2.9-4.7, tools/native/handoff_probe.hip `kgate` / `vgate`), but watches the 100 MHz wall clock: after `ticks` without the
// value it writes FFGP_HANDOFF_WATCHDOG into the status word and LEAVES -- the kernels behind it then run on incomplete data and the
// call returns FFGP_ERR_HANDOFF instead of never returning; the handle goes back to event pairs (ffgp_map_info).
__global__ void ffgp_handoff_gate(const unsigned* __restrict__ word, unsigned need, int* info, long ticks) {
  if (threadIdx.x != 0) return;
  const long t0 = wall_clock64();
  while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < need) {
    __builtin_amdgcn_s_sleep(1);
    if (wall_clock64() - t0 > ticks) {
      atomicCAS(info, 0, FFGP_HANDOFF_WATCHDOG);
      return;
    }
  }
  // (no fence of its own: the kernels behind the gate acquire at their dispatch, as after any kernel boundary)
}
// Create-time self-test across two streams (api.hip, create_resources): the gate is enqueued on the side stream FIRST, the write that
// satisfies it on the handle's own stream afterwards.  Where kernels of different queues can run side by side the gate sees the value
// within microseconds; in a process whose dispatches are serialised it gives up after 50 ms -- the handle then keeps the event pairs for
// its whole life (ho_selftest_failed: option "ho_values" = 1 is refused).  This is the DETECTION; the list of environment variables in
// default_ho_values() only spares such a process the 50 ms.
int ffgp_handoff_selftest(ffgp_handle* h) {
  unsigned* probe = h->ho_mem + 9 * 16 + 8;
  int* word = h->d_info + 8;
  FFGP_HIP(hipMemsetAsync(word, 0, sizeof(int), h->aux));
  hipLaunchKernelGGL(ffgp_handoff_gate, dim3(1), dim3(64), 0, h->aux, probe, 2u, word, 50L * 100000L);
  if (hipGetLastError() != hipSuccess) return FFGP_ERR_HIP;
  if (hipStreamWriteValue32(h->own, probe, 2u, 0) != hipSuccess) {
    (void)hipGetLastError();
    FFGP_HIP(hipMemsetAsync(probe, 0xff, sizeof(unsigned), h->own));      // (no value operations: release the gate another way)
    FFGP_HIP(hipStreamSynchronize(h->own));
    FFGP_HIP(hipStreamSynchronize(h->aux));
    return 1;
  }
  FFGP_HIP(hipStreamSynchronize(h->aux));
  FFGP_HIP(hipStreamSynchronize(h->own));
  int got = 0;
  FFGP_HIP(hipMemcpy(&got, word, sizeof(int), hipMemcpyDeviceToHost));
  FFGP_HIP(hipMemset(word, 0, sizeof(int)));
  FFGP_HIP(hipMemset(probe, 0, sizeof(unsigned)));
  return got != 0 ? 1 : 0;      // 1: the gate timed out -- kernels of two queues do not overlap here
}
// test hook (option "ho_withhold" = k > 0): the k-th publication from now on is never written -- the consumer's gate must time out
static inline bool la_withheld(ffgp_handle* h) {
  if (h->ho_withhold <= 0) return false;
  return --h->ho_withhold == 0;
}
// submission-order rule (see ffgp_potrf_impl): a wait is only ever ENQUEUED after the launch that satisfies it, so that polling gates in
// shared in-order hardware queues cannot form a cycle.  ho_launched[slot] = the newest sequence number whose producing operation has
// been enqueued; la_wait checks it (development build: an error; product: counted in ho_order_violations).
static inline void la_mark_launched(ffgp_handle* h, int slot) { h->ho_launched[slot] = h->ho_seq[slot]; }
static int la_record(ffgp_handle* h, hipEvent_t ev, hipStream_t s) {
  const int slot = la_slot(h, ev);
  if (slot < 0) {
    FFGP_HIP(hipEventRecord(ev, s));
    return FFGP_OK;
  }
  h->ho_seq[slot] += 1;
  la_mark_launched(h, slot);
  if (la_withheld(h)) return FFGP_OK;
  FFGP_HIP(hipStreamWriteValue32(s, h->ho_mem + slot * 16, h->ho_seq[slot], 0));
  return FFGP_OK;
}
static int la_wait(ffgp_handle* h, hipStream_t s, hipEvent_t ev) {
  const int slot = la_slot(h, ev);
  if (slot < 0) {
    FFGP_HIP(hipStreamWaitEvent(s, ev, 0));
    return FFGP_OK;
  }
  if (h->ho_launched[slot] != h->ho_seq[slot]) {      // the producer of this number has not been enqueued yet
    h->ho_order_violations += 1;
#ifdef FFGP_DEV_OPTIONS
    fprintf(stderr, "[ffgp] look-ahead: wait for hand-off %d #%u enqueued before its producer (#%u launched)\n", slot, h->ho_seq[slot],
            h->ho_launched[slot]);
    return FFGP_ERR_HIP;
#endif
  }
  if (h->ho_gate) {
    int* info = h->ho_info ? h->ho_info : h->d_info;
    hipLaunchKernelGGL(ffgp_handoff_gate, dim3(1), dim3(64), 0, s, h->ho_mem + slot * 16, h->ho_seq[slot], info,
                       (long)h->ho_timeout_ms * 100000L);
    return hipGetLastError() == hipSuccess ? FFGP_OK : FFGP_ERR_HIP;
  }
  FFGP_HIP(hipStreamWaitValue32(s, h->ho_mem + slot * 16, h->ho_seq[slot], hipStreamWaitValueGte, 0xffffffffu));
  return FFGP_OK;
}
// "record" whose word is written by the NEXT diagonal-block kernel launched on `s` as it starts (launch_diag / the ragged chain pick the
// pending publication up): on the chain's stream the hipStreamWriteValue32 is a 5 us kernel of its own between a panel's last update and
// the next panel's first diagonal block.  la_flush writes a publication nobody picked up (end of the factorisation, error paths: a gate
// already enqueued on another stream must never be left waiting).
static int la_flush_gemm(ffgp_handle* h) {
  if (h->ho_gdefer_slot < 0) return FFGP_OK;
  const int slot = h->ho_gdefer_slot;
  h->ho_gdefer_slot = -1;
  la_mark_launched(h, slot);
  FFGP_HIP(hipStreamWriteValue32(h->ho_gdefer_stream, h->ho_mem + slot * 16, h->ho_seq[slot], 0));
  return FFGP_OK;
}
static int la_flush(ffgp_handle* h) {
  const int grc = la_flush_gemm(h);
  if (h->ho_defer_slot < 0) return grc;
  const int slot = h->ho_defer_slot;
  h->ho_defer_slot = -1;
  la_mark_launched(h, slot);
  FFGP_HIP(hipStreamWriteValue32(h->ho_defer_stream, h->ho_mem + slot * 16, h->ho_seq[slot], 0));
  return grc;
}
// "record" on the update stream whose word is written by the next GEMM launched there as it starts (gemm_plan picks it up): S_bz's
// hand-off rides on the S_ii launch behind it instead of on a 5 us write kernel between the two
static int la_record_on_next_gemm(ffgp_handle* h, hipEvent_t ev, hipStream_t s) {
  const int slot = la_slot(h, ev);
  if (slot < 0 || h->ho_defer < 2) return la_record(h, ev, s);
  FFGP_CHECK(la_flush_gemm(h));
  h->ho_seq[slot] += 1;
  h->ho_gdefer_slot = slot;
  h->ho_gdefer_stream = s;
  return FFGP_OK;
}
static int la_record_deferred(ffgp_handle* h, hipEvent_t ev, hipStream_t s) {
  const int slot = la_slot(h, ev);
  // (only ffgp_potrf_diag128_v3 publishes: with another diagonal-block kernel selected -- option diag_v2 -- nobody would pick the word up, and
  //  the plain write at the end of the factorisation would sit behind kernels that wait for it; found by the suite's barrier-kernel case)
  if (slot < 0 || !h->ho_defer || h->diag_v2 != 4 || h->diag_dbg || h->use_naive) return la_record(h, ev, s);
  FFGP_CHECK(la_flush(h));
  h->ho_seq[slot] += 1;
  h->ho_defer_slot = slot;
  h->ho_defer_stream = s;
  return FFGP_OK;
}
static inline void la_take_deferred(ffgp_handle* h, DiagRag& dr) {
  dr.pub = nullptr;
  dr.pub_val = 0;
  if (h->ho_defer_slot >= 0 && h->ho_defer_stream == h->stream) {
    dr.pub = h->ho_mem + h->ho_defer_slot * 16;
    dr.pub_val = h->ho_seq[h->ho_defer_slot];
    la_mark_launched(h, h->ho_defer_slot);
    h->ho_defer_slot = -1;
    if (la_withheld(h)) dr.pub = nullptr;      // (test hook: this publication never happens)
  }
}
struct LaFlushGuard {      // whatever path leaves the factorisation, a pending publication is written
  ffgp_handle* h;
  ~LaFlushGuard() { (void)la_flush(h); }
};

// called once per factorisation, before its first hand-off: value hand-offs unless the stream is being captured; the sequence
// numbers start over (all of the handle's streams drained first) long before they could wrap
static int la_begin(ffgp_handle* h) {
  h->ho_active = 0;
  h->ho_defer_slot = -1;
  h->ho_gdefer_slot = -1;
  if (!h->ho_values || !h->ho_mem) return FFGP_OK;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(h->stream, &st) != hipSuccess) {
    (void)hipGetLastError();
    return FFGP_OK;
  }
  if (st != hipStreamCaptureStatusNone) return FFGP_OK;
  // A value wait is a polling kernel: it must never sit in the same hardware queue in front of its producer.  The runtime keeps queues per
  // priority, and the side stream is created at the greatest one -- a caller's stream of that same priority (ffgp_set_stream) could share
  // its queue, so such a binding keeps the event pairs.
  int pm = 0, pa = 0;
  if (hipStreamGetPriority(h->stream, &pm) != hipSuccess || hipStreamGetPriority(h->aux, &pa) != hipSuccess) {
    (void)hipGetLastError();
    return FFGP_OK;
  }
  if (pm == pa) return FFGP_OK;
  unsigned top = 0;
  for (int i = 0; i < 10; ++i) top = max(top, h->ho_seq[i]);
  if (top > 0x3fffffffu) {
    FFGP_HIP(hipStreamSynchronize(h->stream));
    FFGP_HIP(hipStreamSynchronize(h->aux));
    if (h->aux2) FFGP_HIP(hipStreamSynchronize(h->aux2));
    if (h->aux3) FFGP_HIP(hipStreamSynchronize(h->aux3));
    if (h->masked) FFGP_HIP(hipStreamSynchronize(h->masked));
    FFGP_HIP(hipMemsetAsync(h->ho_mem, 0, 10 * 16 * sizeof(unsigned), h->stream));
    FFGP_HIP(hipStreamSynchronize(h->stream));
    for (int i = 0; i < 10; ++i) h->ho_seq[i] = 0;
  }
  h->ho_active = 1;
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
// the factorisation's GEMMs: in a batched factorisation (h->bt_F blocks at fixed strides) every launch covers all blocks
static int potrf_gemm(ffgp_handle* h, int opa, int opb, int mode, int syrk_tag, const double* A, int lda, const double* B, int ldb,
                      double* C, int ldc, int m, int n, int k, double alpha, double beta, int tri = 0, int alias = ALIAS_NONE,
                      bool b_is_dinv = false) {
  if (h->bt_F > 1)
    return ffgp_gemm_launch(h, opa, opb, mode, syrk_tag, A, lda, B, ldb, C, ldc, m, n, k, alpha, beta, tri, alias, -h->bt_F, h->bt_sA,
                            b_is_dinv ? h->bt_sD : h->bt_sA, h->bt_sA);     // (-F: one member's tile shapes, see ffgp_gemm_launch)
  return ffgp_gemm_launch(h, opa, opb, mode, syrk_tag, A, lda, B, ldb, C, ldc, m, n, k, alpha, beta, tri, alias);
}

// `first_diag_done`: the caller has already launched the panel's first diagonal block on this stream (factor_panel_first_diag below)
static int factor_panel(ffgp_handle* h, double* A, int n, int mtot, int lda, int k0, int w1, hipEvent_t gate = nullptr, int carry = 0,
                        hipEvent_t gate2 = nullptr, bool first_diag_done = false) {
  const int pend = k0 + w1;
  for (int j0 = k0; j0 < pend; j0 += NB) {
    const int jb = min(NB, n - j0);
    double* Ajj = A + (size_t)j0 * lda + j0;
    double* Dj = h->dinv + (size_t)(j0 / NB) * NB * NB;
    if (!(first_diag_done && j0 == k0)) FFGP_CHECK(launch_diag(h, Ajj, lda, jb, Dj, j0, 1));
    const int mrows = mtot - (j0 + jb);
    if (mrows > 0) {
      double* A21 = A + (size_t)(j0 + jb) * lda + j0;
      // TRSM as GEMM: A21 <- A21 * Dj^T (in place: one column tile, each workgroup rewrites only rows it read)
      if (trsm128_fits(h, A21, lda, jb, mrows * max(1, h->bt_F)) && (h->bt_F <= 1 || !(h->bt_sA & 1)))
        FFGP_CHECK(launch_trsm128(h, A21, lda, mrows, Dj));
      else
        FFGP_CHECK(potrf_gemm(h, OP_KMAJOR, OP_KMAJOR, TILES_FULL, 0, A21, lda, Dj, NB, A21, lda, mrows, jb, jb, 1.0, 0.0, 0,
                              ALIAS_A, true));
      const int wrem = pend - (j0 + jb) + carry;
      if (wrem > 0) {
        if (gate && j0 == k0) FFGP_CHECK(la_wait(h, h->stream, gate));
        if (gate2 && j0 == k0) FFGP_CHECK(la_wait(h, h->stream, gate2));
        double* C = A + (size_t)(j0 + jb) * lda + (j0 + jb);
        FFGP_CHECK(potrf_gemm(h, OP_KMAJOR, OP_KMAJOR, TILES_LOWER, 0, A21, lda, A21, lda, C, lda, mrows, wrem, jb, -1.0,
                                    1.0));
      }
    }
  }
  return FFGP_OK;
}

// the first launch of factor_panel on its own: the look-ahead enqueues it before the update stream's wait for the previous panel, because
// that kernel is what publishes "the previous panel is complete" (la_record_deferred) -- see the ordering rule in ffgp_potrf_impl
static int factor_panel_first_diag(ffgp_handle* h, double* A, int n, int lda, int k0) {
  return launch_diag(h, A + (size_t)k0 * lda + k0, lda, min(NB, n - k0), h->dinv + (size_t)(k0 / NB) * NB * NB, k0, 1);
}

// the handle's CU-masked stream (tail_mask_m): every CU except the first tail_mask_cus of each XCD.  Mask bit i <-> XCD i % 8, CU
// i / 8 of that XCD (tools/native/cumask_probe.hip); a mask that leaves an XCD empty is ignored by the runtime.
int ffgp_ensure_masked(ffgp_handle* h) {
  if (h->masked) return FFGP_OK;
  if (h->masked_failed) return FFGP_ERR_HIP;
  uint32_t mask[8];
  for (int i = 0; i < 8; ++i) mask[i] = 0xffffffffu;
  const int cut = h->tail_mask_cus > 0 && h->tail_mask_cus < 32 ? h->tail_mask_cus : 8;
  for (int x = 0; x < 8; ++x)
    for (int c = 0; c < cut; ++c) {
      const int bit = c * 8 + x;
      mask[bit / 32] &= ~(1u << (bit % 32));
    }
  if (hipExtStreamCreateWithCUMask(&h->masked, 8, mask) != hipSuccess) {
    (void)hipGetLastError();
    h->masked = nullptr;
    h->masked_failed = 1;
    return FFGP_ERR_HIP;
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
  FFGP_CHECK(la_begin(h));
  h->ho_info = (h->bt_F > 1 && h->bt_info) ? h->bt_info : h->d_info;
  if (!h->fold_info) FFGP_CHECK(ffgp_zero_async(h, h->d_info, sizeof(int)));   // (ffgp_train_raw's loop: its Adam kernel clears the word)
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
    const bool in_order = !h->lookahead || n <= NB1 || n <= h->la_min_n;
    // Passenger rows off the chain (round 5).  Riding in the chain's own launches, the right-hand sides made every TRSM and
    // panel update of the dependency chain (n - j) + d rows tall: at N = 8192, d = 4096 a third of the chain's kernel time, on the
    // latency tiles, while most of the chip idled in the chain-bound tail.  With many of them (>= pass_split_min) and a look-ahead
    // form, the chain and the trailing updates cover the n x n matrix only, and the passenger rows follow ONE PANEL BEHIND on a
    // stream of their own: per panel the same operations as before (per 128-column block the product with the inverted diagonal
    // block and the update of the panel's remaining columns, then the K = panel-width update of the columns to the right), which
    // read the panel's finished columns of L and write passenger rows only -- they fill the CUs the chain leaves idle.
    const int npass = mtot - n;
    const bool split_pass = !in_order && h->pass_split_min > 0 && npass >= h->pass_split_min;
    const int mch = split_pass ? n : mtot;       // rows the chain's and the trailing updates' launches cover
    hipEvent_t pass_done = nullptr;
    if (split_pass) FFGP_CHECK(ffgp_ensure_aux2(h));
    auto pass_panel = [&](int k0, int w1, hipEvent_t ready) -> int {
      // `ready`: the panel's columns of L are final.  Runs on h->aux3; h->stream is restored by the caller's bookkeeping.
      hipStream_t keep = h->stream;
      FFGP_CHECK(la_wait(h, h->aux3, ready));
      h->stream = h->aux3;
      int rc = FFGP_OK;
      const int pend = k0 + w1;
      double* Ap0 = A + (size_t)n * lda;           // first passenger row
      for (int j0 = k0; j0 < pend && rc == FFGP_OK; j0 += NB) {
        const int jb = min(NB, n - j0);
        double* Ap = Ap0 + j0;
        double* Dj = h->dinv + (size_t)(j0 / NB) * NB * NB;
        rc = potrf_gemm(h, OP_KMAJOR, OP_KMAJOR, TILES_FULL, 0, Ap, lda, Dj, NB, Ap, lda, npass, jb, jb, 1.0, 0.0, 0, ALIAS_A, true);
        const int wrem = pend - (j0 + jb);
        if (rc == FFGP_OK && wrem > 0)
          rc = potrf_gemm(h, OP_KMAJOR, OP_KMAJOR, TILES_FULL, 0, Ap, lda, A + (size_t)(j0 + jb) * lda + j0, lda, Ap0 + (j0 + jb), lda, npass,
                          wrem, jb, -1.0, 1.0);
      }
      const int mt = n - pend;
      if (rc == FFGP_OK && mt > 0)
        rc = potrf_gemm(h, OP_KMAJOR, OP_KMAJOR, TILES_FULL, 0, Ap0 + k0, lda, A + (size_t)pend * lda + k0, lda, Ap0 + pend, lda, npass, mt, w1,
                        -1.0, 1.0);
      h->stream = keep;
      if (rc == FFGP_OK) {
        FFGP_CHECK(la_record(h, h->la_ev[9], h->aux3));
        pass_done = h->la_ev[9];
      }
      return rc;
    };
    if (in_order) {
      for (int k0 = 0; k0 < n; k0 += pw(k0)) {
        const int w1 = min(pw(k0), n - k0);
        const int pend = k0 + w1;  // end column of this outer panel
        FFGP_CHECK(factor_panel(h, A, n, mtot, lda, k0, w1));
        const int mt = n - pend;  // trailing columns; trailing rows include the passenger rows
        if (mt > 0) {
          double* P = A + (size_t)pend * lda + k0;
          double* C = A + (size_t)pend * lda + pend;
          FFGP_CHECK(potrf_gemm(h, OP_KMAJOR, OP_KMAJOR, TILES_LOWER, 1, P, lda, P, lda, C, lda, mtot - pend, mt, w1,
                                      -1.0, 1.0));
        }
      }
    } else {
      // Look-ahead.  A high-priority side stream runs the dependency chain (per 128-column block: diagonal factor, TRSM, update of the
      // panel's remaining columns), the main stream the trailing updates.  Two forms of an iteration (panel k complete, panel k+1 next):
      //
      // ROUND-1 form (blocks of more than la_carry_n rows, while their trailing matrix has more than la_carry_rows rows).  The trailing update of step k is cut into S_a (the
      // first 128 columns of panel k+1 -- all that its first diagonal factor and TRSM read), S_b (the rest of panel k+1's columns) and
      // S_ii (everything to the right).  The side stream runs  S_a(k) -> panel k+1  back to back; the main stream runs S_b(k), S_ii(k).
      // S_a sits on the chain's own stream: the only waits left on the chain are for hand-offs that fired long before (S_ii(k-1), S_b(k)).
      //
      // CARRY form (from there on; the whole factorisation for n <= la_carry_n).  Panel k's own update kernels (one per 128-column
      // block, K = 128) also cover Z(k+1) = the first 128 columns of panel k+1, so the chain goes from the last TRSM of panel k straight
      // into the first diagonal block of panel k+1 -- no K = 512 strip update and no wait between two panels.  The main stream's update of
      // step k is S_b(k) together with S_z(k) (Z(k+2): panel k's contribution to the strip that chain k+1 will carry into -- its
      // right-hand neighbour, so the two are one launch, S_bz) and S_ii(k) (everything right of Z(k+2)).  Writers of any one column range
      // are ordered: Z(k+2) <- S_ii(<= k-1), S_z(k) on the main stream, then chain k+1 (after S_bz's hand-off).
      //
      // The carry form wins where the chain is (nearly) the critical path, the round-1 form where the chain hides under the update: the
      // carry makes the panel's K = 128 launches 128 columns wider.  Whole factorisations: carry -1.25 % at N = 14336, equal at 16384 /
      // 20480, +0.6 ... +0.9 % at 24576 / 32768 (profiles/r05i_*).  So the form is chosen per ITERATION: the first carry iteration finds
      // a panel that did not carry Z(k+1) and runs S_a(k) once more; from then on every panel carries.
      hipStream_t main_s = h->stream;
      LaFlushGuard flush_guard{h};
      struct Polite64 {        // the carry iterations' 64-tile trailing updates leave half of every CU to the chain (gemm_plan); reset on every exit path
        int& f;
        explicit Polite64(int& f_) : f(f_) { f = 0; }
        ~Polite64() { f = 0; }
      } polite64(h->polite64_active);
      auto carry_of = [&](int pend_) { return max(0, min(NB, n - pend_)); };   // columns of the next panel's first block (0 at the end)
      const int w0 = min(pw(0), n);
      bool carried = (h->la_carry == 1) || (h->la_carry == 2 && n <= h->la_carry_n);      // does the complete panel carry Z(k+1)?  (blocks up to la_carry_n rows: carry form throughout)
      FFGP_CHECK(factor_panel(h, A, n, mch, lda, 0, w0, nullptr, carried ? carry_of(w0) : 0));
      FFGP_CHECK(la_record(h, h->la_ev[6], main_s));
      FFGP_CHECK(la_wait(h, h->aux, h->la_ev[6]));
      if (split_pass) FFGP_CHECK(pass_panel(0, w0, h->la_ev[6]));
      int it = 0;
      hipEvent_t eb_prev = nullptr, ei_prev = nullptr;
      // Chain-bound tail on a CU-masked stream (option tail_mask_m): once the trailing matrix has fewer rows than that, the trailing
      // updates are issued to a stream that may not use tail_mask_cus CUs of every XCD -- those CUs stay free of trailing-update
      // workgroups, so the chain's kernels (unmasked side stream) start at once and run undisturbed instead of waiting for slots and
      // sharing SIMDs with the update's MFMA stream; the update loses a quarter of the chip where it has slack anyway.
      hipStream_t syrk_s = main_s;
      for (int k0 = 0; k0 < n; k0 += pw(k0), ++it) {
        const int w1 = min(pw(k0), n - k0);
        const int pend = k0 + w1;
        const int mt = n - pend;
        if (mt <= 0) break;
        // carry iteration: panel k+1 is factored carrying Z(k+2) (every later iteration is one too: the trailing matrix only shrinks)
        const bool cm = (h->la_carry == 1) || (h->la_carry == 2 && (carried || mt <= h->la_carry_rows));
        h->polite64_active = cm ? 1 : 0;
        if (cm && h->tail_mask_m > 0 && mch - pend < h->tail_mask_m && syrk_s == main_s && ffgp_ensure_masked(h) == FFGP_OK) {
          FFGP_CHECK(la_record(h, h->la_ev[7], main_s));
          FFGP_CHECK(la_wait(h, h->masked, h->la_ev[7]));
          syrk_s = h->masked;
        }
        const int wn = min(pw(pend), mt);  // width of the next panel
        const int q = pend + wn;           // first column of panel k+2
        const int wz = cm ? carry_of(q) : 0;
        hipEvent_t eb = h->la_ev[(it & 1) * 3], eg = h->la_ev[(it & 1) * 3 + 1], ei = h->la_ev[(it & 1) * 3 + 2];
        const int wa = (cm || h->la_split) ? min(NB, wn) : wn;     // Z(k+1)
        // side stream: S_a(k) unless panel k carried it (after S_ii(k-1), which carried panel k-1 into these columns)
        if (!carried) {
          if (ei_prev) FFGP_CHECK(la_wait(h, h->aux, ei_prev));
          double* P = A + (size_t)pend * lda + k0;
          double* C = A + (size_t)pend * lda + pend;
          h->stream = h->aux;
          const int arc = potrf_gemm(h, OP_KMAJOR, OP_KMAJOR, TILES_LOWER, 1, P, lda, P, lda, C, lda, mch - pend, wa, w1, -1.0, 1.0);
          h->stream = main_s;
          FFGP_CHECK(arc);
        }
        // ORDER OF SUBMISSION.  Every wait below is enqueued AFTER the launch that will satisfy it.  A value wait is a kernel at the head of
        // its hardware queue, and streams of different handles (or the application's own) may share a queue: if a wait could be
        // enqueued before its producer, two handles could each sit in front of the other's producer and wait for ever (seen once the
        // suite ran beside its background load with the "panel complete" word left to the NEXT diagonal-block kernel: that kernel used
        // to be enqueued after the update stream's wait for it).  With producers first, dependencies only point backwards in submission
        // order -- the property event pairs have by construction -- and shared in-order queues cannot form a cycle.  Hence: the next
        // panel's first diagonal block (publishes "panel k complete"), then the update stream's wait for it, S_bz and S_ii (whose first
        // workgroup publishes S_bz's hand-off), and only then the rest of the panel with its gate.
        h->stream = h->aux;
        // The panel's first diagonal block runs while S_bz -- released by that very kernel's first instruction -- floods the chip, and the
        // S_bz workgroups that land beside it cost its pivot wave a third of its speed (39 us instead of 28 on the C2 timeline).  In the
        // chain-bound iterations the chip is idle when it is launched (the update of the step before is long done), so it can ask for
        // a whole CU's LDS and keep that CU to itself: N = 2048 / 4096 / 8192 -1.7 / -0.8 / -0.8 %.  Only for the process's only handle:
        // beside other handles' kernels an empty CU may be a long time coming.
        h->diag_excl_now = (cm && mt <= h->diag_excl_rows && ffgp_live_handles() == 1) ? 1 : 0;
        const int drc = factor_panel_first_diag(h, A, n, lda, pend);
        h->diag_excl_now = 0;
        h->stream = main_s;
        FFGP_CHECK(drc);
        // main stream, once panel k is complete: S_b(k) (and S_z(k), its right-hand neighbour: columns pend+wa .. q+wz, one launch)
        if (eb_prev) FFGP_CHECK(la_wait(h, syrk_s, eb_prev));
        hipEvent_t gate = nullptr;
        if (wn - wa + wz > 0) {
          double* Pb = A + (size_t)(pend + wa) * lda + k0;
          double* Cb = A + (size_t)(pend + wa) * lda + (pend + wa);
          h->stream = syrk_s;
          const int brc = potrf_gemm(h, OP_KMAJOR, OP_KMAJOR, TILES_LOWER, 1, Pb, lda, Pb, lda, Cb, lda, mch - pend - wa,
                                     wn - wa + wz, w1, -1.0, 1.0);
          h->stream = main_s;
          FFGP_CHECK(brc);
          FFGP_CHECK(la_record_on_next_gemm(h, eg, syrk_s));     // (published by S_ii(k), the next launch on that stream)
          gate = eg;
        }
        // main stream: S_ii(k), everything right of panel k+1 and of Z(k+2)
        const int mt2 = mt - wn - wz;
        ei_prev = nullptr;
        if (mt2 > 0) {
          double* P2 = A + (size_t)(q + wz) * lda + k0;
          double* C2 = A + (size_t)(q + wz) * lda + (q + wz);
          h->stream = syrk_s;
          const int irc = potrf_gemm(h, OP_KMAJOR, OP_KMAJOR, TILES_LOWER, 1, P2, lda, P2, lda, C2, lda, mch - q - wz, mt2, w1, -1.0, 1.0);
          h->stream = main_s;
          FFGP_CHECK(irc);
          if (!cm) {                                          // (the next iteration's S_a waits for it)
            FFGP_CHECK(la_record(h, ei, syrk_s));
            ei_prev = ei;
          }
        }
        FFGP_CHECK(la_flush_gemm(h));                         // (no S_ii: the hand-off of S_bz is written plainly)
        // side stream: the rest of panel k+1 (carrying Z(k+2) in a carry iteration)
        h->stream = h->aux;
        const int rc = factor_panel(h, A, n, mch, lda, pend, wn, gate, wz, nullptr, true);
        h->stream = main_s;
        FFGP_CHECK(rc);
        if (cm && !split_pass)
          FFGP_CHECK(la_record_deferred(h, eb, h->aux));     // (published by the next panel's first diagonal-block kernel)
        else
          FFGP_CHECK(la_record(h, eb, h->aux));              // (the next kernel on the side stream is S_a(k+1); pass_panel waits at once)
        eb_prev = eb;
        if (split_pass) FFGP_CHECK(pass_panel(pend, wn, eb));
        if (h->tri_hook_col > 0 && pend + wn == h->tri_hook_col) {   // the factor's columns < tri_hook_col are final from here on
          FFGP_CHECK(la_record(h, h->tri_ev[0], h->aux));
          h->tri_hook_fired = 1;
        }
        carried = cm;
      }
      FFGP_CHECK(la_flush(h));
      if (syrk_s != main_s) {
        FFGP_CHECK(la_record(h, h->la_ev[8], syrk_s));
        FFGP_CHECK(la_wait(h, main_s, h->la_ev[8]));
      }
      if (eb_prev) FFGP_CHECK(la_wait(h, main_s, eb_prev));
      if (pass_done) FFGP_CHECK(la_wait(h, main_s, pass_done));
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

// ------------------------------------------------------------------------------------------------------------
// ONE factorisation chain for R blocks of DIFFERENT sizes (ffgp_nlml_fused_batch with unequal members; the reference's fidelities are
// ragged by nature: 300 / 300 / 250 points in FidelityFusion_Models/ResGP.py:121-136, 100 low against 4..32 high in
// Experiments/GAR_Aligned/exp_aligned.py:66-74).  Every member follows the launch sequence of its OWN single call -- in order for
// n <= la_min_n, the carry form of the look-ahead above that -- and launches of the same kind at the same chain step are merged:
// the diagonal blocks of column j0 of all members that still have one are one launch (one workgroup each), their TRSMs one ragged
// GEMM launch, their panel updates another (ffgp_gemm_launch_rag: each member's own sizes, tile shape and form, so its values are the
// single call's bit for bit); a member drops out of the launches when its columns are used up.  The chain runs max(n) / 128 steps on
// the side stream; trailing updates of in-order members follow their panel on the chain's stream (as in their single call), those
// of look-ahead members run on the main stream behind the same two events per panel as in ffgp_potrf_impl's carry form.
// Not covered (FFGP_ERR_ARG, the caller evaluates such sets block by block): members above 12288 rows (their single call uses the
// round-1 look-ahead form), the naive / barrier-kernel modes, wide early panels (nb_big).
// ------------------------------------------------------------------------------------------------------------
int ffgp_potrf_ragged(ffgp_handle* h, int R, const ffgp_rag_block* mem) {
  if (R <= 0) return FFGP_OK;
  if (!mem || h->use_naive || h->diag_v2 != 4 || h->diag_dbg || h->nb_big > h->nb_outer) return FFGP_ERR_ARG;
  const int NB1 = h->nb_outer;
  std::vector<int> form(R);
  bool any_la = false;
  int nmax = 0;
  for (int f = 0; f < R; ++f) {
    const ffgp_rag_block& b = mem[f];
    if (!b.A || !b.dinv || b.n <= 0 || b.lda < b.n || b.mtot < b.n || (b.lda & 1) || (reinterpret_cast<uintptr_t>(b.A) & 15)) return FFGP_ERR_ARG;
    if (!h->lookahead || b.n <= NB1 || b.n <= h->la_min_n) form[f] = 0;
    else if (h->la_carry == 1 || (h->la_carry == 2 && b.n <= h->la_carry_n)) form[f] = 1;
    else return FFGP_ERR_ARG;
    any_la = any_la || form[f] == 1;
    nmax = max(nmax, b.n);
  }
  if (!(h->diag_attr_set & 4)) {
    FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_potrf_diag128_v3<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 DIAG_LDS_BYTES));
    FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_potrf_diag128_v3<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 DIAG_LDS_BYTES));
    h->diag_attr_set |= 4;
  }
  if (h->diag_v4 && !(h->diag_attr_set & 32)) {      // (the ragged launches never ask for more than the kernel's own image)
    if (!(h->diag_attr_set & 16)) {
      FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_potrf_diag128_v4<true>), hipFuncAttributeMaxDynamicSharedMemorySize, DIAG4_LDS_BYTES));
      FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_potrf_diag128_v4<false>), hipFuncAttributeMaxDynamicSharedMemorySize, DIAG4_LDS_BYTES));
    }
    h->diag_attr_set |= 32;
  }
  FFGP_CHECK(la_begin(h));
  h->ho_info = h->bt_info + mem[0].info_index;      // (a gate that gives up reports through the first member's status word)
  hipStream_t main_s = h->stream;
  hipStream_t chain_s = any_la ? h->aux : main_s;
  struct StreamGuard {      // whatever path leaves this function (the FFGP_HIP macros return at once), the handle gets its stream back
    ffgp_handle* h; hipStream_t s;
    ~StreamGuard() { h->stream = s; }
  } guard{h, main_s};
  LaFlushGuard flush_guard{h};
  if (any_la) {
    FFGP_CHECK(la_record(h, h->la_ev[6], main_s));
    FFGP_CHECK(la_wait(h, chain_s, h->la_ev[6]));
  }
  std::vector<GemmRagIn> in;
  in.reserve(R);
  hipEvent_t eg_prev = nullptr;
  int it = 0, rc = FFGP_OK;
  for (int k0 = 0; k0 < nmax && rc == FFGP_OK; k0 += NB1, ++it) {
    hipEvent_t eb = h->la_ev[(it & 1) * 3], eg = h->la_ev[(it & 1) * 3 + 1];
    // ---- the chain of this panel, every member that still has columns here
    h->stream = chain_s;
    bool gate_pending = eg_prev != nullptr;
    for (int j0 = k0; j0 < min(k0 + NB1, nmax) && rc == FFGP_OK; j0 += NB) {
      for (int f0 = 0; f0 < R && rc == FFGP_OK; f0 += FFGP_RAG_MAX) {      // diagonal blocks: one workgroup per member, 8 members per launch
        DiagRag dr;
        int cnt = 0;
        for (int f = f0; f < min(R, f0 + FFGP_RAG_MAX); ++f) {
          const ffgp_rag_block& b = mem[f];
          if (j0 >= b.n) continue;
          dr.A[cnt] = b.A + (size_t)j0 * b.lda + j0;
          dr.Dinv[cnt] = b.dinv + (size_t)(j0 / NB) * NB * NB;
          dr.lda[cnt] = b.lda;
          dr.nb[cnt] = min(NB, b.n - j0);
          dr.info[cnt] = b.info_index;
          ++cnt;
        }
        if (!cnt) continue;
        la_take_deferred(h, dr);
        for (int c = cnt; c < FFGP_RAG_MAX; ++c) { dr.A[c] = dr.A[0]; dr.Dinv[c] = dr.Dinv[0]; dr.lda[c] = dr.lda[0]; dr.nb[c] = dr.nb[0]; dr.info[c] = dr.info[0]; }
        if (h->diag_v4) {
          hipLaunchKernelGGL(ffgp_potrf_diag128_v4<true>, dim3(cnt), dim3(512), DIAG4_LDS_BYTES,
                             h->stream, (double*)nullptr, 0, 0, (double*)nullptr, h->bt_info, j0, h->aux_prio, 0L, 0L, 0, dr);
        } else
        hipLaunchKernelGGL(ffgp_potrf_diag128_v3<true>, dim3(cnt), dim3(512), DIAG_LDS_BYTES, h->stream, (double*)nullptr, 0, 0, (double*)nullptr,
                           h->bt_info, j0, h->aux_prio, 0L, 0L, 0, dr);
      }
      // TRSM of every row below: A21 <- A21 * Dj^T, in place
      in.clear();
      TrsmSet ts = TrsmSet();
      int tcnt = 0, tmax = 0;
      long rows_all = 0;
      for (int f = 0; f < R; ++f)
        if (j0 < mem[f].n) rows_all += max(0, mem[f].mtot - (j0 + min(NB, mem[f].n - j0)));
      const bool own_kernel = rows_all <= h->trsm128_max_m;
      for (int f = 0; f < R; ++f) {
        const ffgp_rag_block& b = mem[f];
        if (j0 >= b.n) continue;
        const int jb = min(NB, b.n - j0), mrows = b.mtot - (j0 + jb);
        if (mrows <= 0) continue;
        double* A21 = b.A + (size_t)(j0 + jb) * b.lda + j0;
        double* Dj = b.dinv + (size_t)(j0 / NB) * NB * NB;
        if (own_kernel && trsm128_fits(h, A21, b.lda, jb, mrows)) {      // (same values as the general GEMM: members may take either, one by one)
          ts.A[tcnt] = A21; ts.D[tcnt] = Dj; ts.lda[tcnt] = b.lda; ts.mrows[tcnt] = mrows;
          tmax = max(tmax, mrows);
          if (++tcnt == FFGP_RAG_MAX) {
            rc = launch_trsm128_set(h, ts, tcnt, tmax);
            tcnt = tmax = 0;
            if (rc != FFGP_OK) break;
          }
        } else {
          in.push_back(GemmRagIn{A21, b.lda, Dj, NB, A21, b.lda, mrows, jb, jb});
        }
      }
      if (rc == FFGP_OK && tcnt) rc = launch_trsm128_set(h, ts, tcnt, tmax);
      if (rc == FFGP_OK && !in.empty()) rc = ffgp_gemm_launch_rag(h, TILES_FULL, 0, (int)in.size(), in.data(), 1.0, 0.0, ALIAS_A);
      if (rc != FFGP_OK) break;
      // update of the panel's remaining columns (look-ahead members: and of the next panel's first block, the carry)
      in.clear();
      for (int f = 0; f < R; ++f) {
        const ffgp_rag_block& b = mem[f];
        if (j0 >= b.n) continue;
        const int jb = min(NB, b.n - j0), mrows = b.mtot - (j0 + jb);
        const int pend = min(k0 + NB1, b.n);
        const int carry = form[f] == 1 ? max(0, min(NB, b.n - pend)) : 0;
        const int wrem = pend - (j0 + jb) + carry;
        if (mrows <= 0 || wrem <= 0) continue;
        double* A21 = b.A + (size_t)(j0 + jb) * b.lda + j0;
        double* C = b.A + (size_t)(j0 + jb) * b.lda + (j0 + jb);
        in.push_back(GemmRagIn{A21, b.lda, A21, b.lda, C, b.lda, mrows, wrem, jb});
      }
      if (!in.empty()) {
        if (gate_pending) {      // the main stream's earlier contribution to these columns (S_bz of the previous panel) must have landed
          FFGP_CHECK(la_wait(h, h->stream, eg_prev));
          gate_pending = false;
        }
        rc = ffgp_gemm_launch_rag(h, TILES_LOWER, 0, (int)in.size(), in.data(), -1.0, 1.0, ALIAS_NONE);
      }
    }
    if (rc != FFGP_OK) break;
    // ---- in-order members: the panel's trailing update follows on the chain's stream
    in.clear();
    for (int f = 0; f < R; ++f) {
      const ffgp_rag_block& b = mem[f];
      if (form[f] != 0 || k0 >= b.n) continue;
      const int pend = min(k0 + NB1, b.n), mt = b.n - pend;
      if (mt <= 0) continue;
      double* P = b.A + (size_t)pend * b.lda + k0;
      double* C = b.A + (size_t)pend * b.lda + pend;
      in.push_back(GemmRagIn{P, b.lda, P, b.lda, C, b.lda, b.mtot - pend, mt, pend - k0});
    }
    if (!in.empty()) rc = ffgp_gemm_launch_rag(h, TILES_LOWER, 1, (int)in.size(), in.data(), -1.0, 1.0, ALIAS_NONE);
    h->stream = main_s;
    if (rc != FFGP_OK || !any_la) continue;
    FFGP_CHECK(la_record(h, eb, chain_s));     // (a plain write: the wait right below must not be enqueued before its producer)
    FFGP_CHECK(la_wait(h, main_s, eb));
    // ---- look-ahead members, main stream: S_bz (the rest of the next panel's columns and its carry strip), then S_ii
    in.clear();
    for (int f = 0; f < R; ++f) {
      const ffgp_rag_block& b = mem[f];
      if (form[f] != 1 || k0 >= b.n) continue;
      const int pend = min(k0 + NB1, b.n), mt = b.n - pend;
      if (mt <= 0) continue;
      const int wn = min(NB1, mt), q = pend + wn, wz = max(0, min(NB, b.n - q)), wa = min(NB, wn);
      if (wn - wa + wz <= 0) continue;
      double* Pb = b.A + (size_t)(pend + wa) * b.lda + k0;
      double* Cb = b.A + (size_t)(pend + wa) * b.lda + (pend + wa);
      in.push_back(GemmRagIn{Pb, b.lda, Pb, b.lda, Cb, b.lda, b.mtot - pend - wa, wn - wa + wz, pend - k0});
    }
    eg_prev = nullptr;
    if (!in.empty()) {
      rc = ffgp_gemm_launch_rag(h, TILES_LOWER, 1, (int)in.size(), in.data(), -1.0, 1.0, ALIAS_NONE);
      if (rc != FFGP_OK) break;
      FFGP_CHECK(la_record(h, eg, main_s));
      eg_prev = eg;
    }
    in.clear();
    for (int f = 0; f < R; ++f) {
      const ffgp_rag_block& b = mem[f];
      if (form[f] != 1 || k0 >= b.n) continue;
      const int pend = min(k0 + NB1, b.n), mt = b.n - pend;
      if (mt <= 0) continue;
      const int wn = min(NB1, mt), q = pend + wn, wz = max(0, min(NB, b.n - q));
      const int mt2 = mt - wn - wz;
      if (mt2 <= 0) continue;
      double* P2 = b.A + (size_t)(q + wz) * b.lda + k0;
      double* C2 = b.A + (size_t)(q + wz) * b.lda + (q + wz);
      in.push_back(GemmRagIn{P2, b.lda, P2, b.lda, C2, b.lda, b.mtot - q - wz, mt2, pend - k0});
    }
    if (!in.empty()) rc = ffgp_gemm_launch_rag(h, TILES_LOWER, 1, (int)in.size(), in.data(), -1.0, 1.0, ALIAS_NONE);
  }
  h->stream = main_s;
  if (la_flush(h) != FFGP_OK && rc == FFGP_OK) rc = FFGP_ERR_HIP;
  if (hipGetLastError() != hipSuccess && rc == FFGP_OK) rc = FFGP_ERR_HIP;
  return rc;
}

// status word -> return code: a pivot index passes through; the diagonal-block kernel's watchdog is a library error
int ffgp_map_info(int v) {
  if (v == FFGP_HANDOFF_WATCHDOG) {
    fprintf(stderr, "[ffgp] look-ahead: a gate waited for a cross-stream hand-off that never came and gave up (a tool that runs this "
                    "process's kernels one at a time?); the call's results are invalid -- FFGP_HANDOFF=events keeps the event pairs\n");
    return FFGP_ERR_HANDOFF;
  }
  if (v >= FFGP_DIAG_WATCHDOG) {
    fprintf(stderr, "[ffgp] potrf_diag128: a wave waited ~1 s for a hand-off inside the kernel and gave up (internal error; hand-off %d)\n",
            v - FFGP_DIAG_WATCHDOG);
    int dbg[32];
    if (hipMemcpyFromSymbol(dbg, HIP_SYMBOL(ffgp_d3_dbg), sizeof(dbg)) == hipSuccess) {
      fprintf(stderr, "[ffgp]   helper %d stage %d column %d block (%d, %d) have 0x%x task %d | seqF %d hs %d hd %d h2 %d | rows", dbg[0], dbg[1], dbg[2],
              dbg[3], dbg[4], dbg[5], dbg[26], dbg[6], dbg[7], dbg[8], dbg[9]);
      for (int z = 0; z < 8; ++z) fprintf(stderr, " %x", dbg[10 + z]);
      fprintf(stderr, " | cntA");
      for (int z = 0; z < 8; ++z) fprintf(stderr, " %d", dbg[18 + z]);
      fprintf(stderr, " | prog w0 %d h0..5 %d %d %d %d %d %d\n", dbg[27] & 0xffff, dbg[28] & 0xffff, dbg[29] & 0xffff, dbg[30] & 0xffff, dbg[31] & 0xffff,
              dbg[27] >> 16, dbg[28] >> 16);
    }
    return FFGP_ERR_HIP;
  }
  return v;
}
