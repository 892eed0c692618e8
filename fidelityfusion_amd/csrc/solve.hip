// Triangular solves, inverse (TRTRI + LAUUM) and likelihood reductions, all built on the fp64 MFMA GEMM with
// the pre-inverted 128x128 diagonal blocks the factorisation leaves in the handle (Dinv store).
#include "ffgp_internal.h"

#define NB FFGP_NB

// make sure h->dinv matches the factor L (it does right after ffgp_potrf on the same buffer)
static int ensure_dinv_for(ffgp_handle* h, const double* L, int n, int ldl) {
  if (h->dinv_L == L && h->dinv_n == n && h->dinv_ld == ldl) return FFGP_OK;
  return ffgp_refresh_dinv(h, L, n, ldl);
}

// ------------------------------------------------------------------------------------------------------------
// Super-block sweeps.  A 128-block sweep is a chain of 2 n/128 dependent launches with k = 128 each: for a thin
// right-hand side (posterior queries, appended points, the backward of a conditional Gaussian) that is pure launch
// latency -- 7 ms at n = 16384 for work that takes 1.4 ms.  The inverses of the S x S diagonal super-blocks (S = 1024:
// three doubling levels above the 128-block inverses the factorisation leaves behind, n S^2 / 3 flops, built once per
// factor and cached in the handle) turn it into 2 n/S launches with k = S.
// ------------------------------------------------------------------------------------------------------------
// The store is addressed like an n x n matrix with leading dimension S: element (r, c) of the block-diagonal inverse at
// Xc[r * S + c].  Rows overlap in memory, but only the band c in (r - S, r] is ever touched and band entries never collide.
// What it buys: stepping one pair down the diagonal is the SAME stride (2s * S + 2s) inside a super-block and across
// super-block boundaries, so every doubling level is one batched launch over all pairs of the matrix.
__global__ void ffgp_copy_dinv_super_kernel(const double* __restrict__ dinv, double* __restrict__ Xc, int S, int n, int b0) {
  const int b = b0 + blockIdx.y;         // 128-block index
  const int r0 = b * NB;
  const int nb = min(NB, n - r0);
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < NB * NB) {
    const int r = idx >> 7, c = idx & 127;
    if (r < nb && c <= r) Xc[(size_t)(r0 + r) * S + r0 + c] = dinv[(size_t)b * NB * NB + idx];
  }
}

static int ensure_sinv_for(ffgp_handle* h, const double* L, int n, int ldl) {
  const int S = h->super_block;
  // a factor that only GREW since the store was built (Posterior.append) keeps its complete leading super-blocks
  int sb_first = 0;
  if (h->sinv_L == L && h->sinv_ld == ldl && h->sinv_S == S && h->sinv_n > 0 && n > h->sinv_n && h->dinv_L == L) sb_first = h->sinv_n / S;
  FFGP_CHECK(ensure_dinv_for(h, L, n, ldl));   // (an incremental Dinv refresh leaves sinv_L in place, a full one clears it)
  if (h->sinv_L == L && h->sinv_n == n && h->sinv_ld == ldl && h->sinv_S == S) return FFGP_OK;
  if (h->sinv_L != L) sb_first = 0;
  // store: the band first, then the T scratch of the doubling levels
  // (one s x s product per pair: n * S / 4 doubles at the last level)
  const size_t band = (size_t)n * (S + 1) + S;   // element (r, c <= r) sits at r * S + c: the last one at (n - 1) * (S + 1)
  const size_t need = (band + (size_t)n * S / 4 + (size_t)S * S) * sizeof(double);
  if (need > h->sinv_bytes) {
    const size_t want = need + need / 8;   // head-room: appended points rarely re-allocate
    double* fresh = nullptr;
    if (hipMalloc(&fresh, want) != hipSuccess) return FFGP_ERR_ALLOC;
    if (h->sinv) {
      if (sb_first > 0)
        FFGP_HIP(hipMemcpyAsync(fresh, h->sinv, ((size_t)sb_first * S * S + (size_t)sb_first * S) * sizeof(double),
                                hipMemcpyDeviceToDevice, h->stream));
      hipStreamSynchronize(h->stream);
      hipFree(h->sinv);
    }
    h->sinv = fresh;
    h->sinv_bytes = want;
    ++h->alloc_epoch;
  }
  double* Xc = h->sinv;
  double* T = h->sinv + (h->sinv_bytes / sizeof(double) - ((size_t)n * S / 4 + (size_t)S * S));   // scratch at the end of the buffer
  const size_t o_first = (size_t)sb_first * S * S + (size_t)sb_first * S;     // first element of super-block sb_first
  FFGP_HIP(hipMemsetAsync(Xc + o_first, 0, (band - o_first) * sizeof(double), h->stream));
  const int nblk = (n + NB - 1) / NB;
  const int b0 = sb_first * (S / NB);
  hipLaunchKernelGGL(ffgp_copy_dinv_super_kernel, dim3(NB * NB / 256, nblk - b0), dim3(256), 0, h->stream, h->dinv, Xc, S, n, b0);
  const int r_first = sb_first * S;
  for (long s = NB; s < S && s < n - r_first; s *= 2) {   // the doubling of ffgp_trtri_impl, stopped below the super-block size
    const int p_first = (int)(r_first / (2 * s));
    const int full = (int)(n / (2 * s)) - p_first;  // pairs with both halves complete (pairs never straddle a super-block)
    const long strideL = 2 * s * (long)ldl + 2 * s, strideX = 2 * s * (long)S + 2 * s;
    const double* Lp = L + (size_t)p_first * strideL;
    double* Xp = Xc + (size_t)p_first * strideX;
    if (full > 0) {
      FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, Lp + (size_t)s * ldl, ldl, Xp, S, T, (int)s, (int)s, (int)s,
                                  (int)s, 1.0, 0.0, TRI_LO_J, ALIAS_NONE, full, strideL, strideX, s * s));
      FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, Xp + (size_t)s * S + s, S, T, (int)s, Xp + (size_t)s * S, S,
                                  (int)s, (int)s, (int)s, -1.0, 0.0, TRI_HI_I, ALIAS_NONE, full, strideX, s * s, strideX));
    }
    const long r0 = ((long)p_first + full) * 2 * s;   // ragged last pair: first half complete, second half partial
    const long n2 = n - r0 - s;
    if (n2 > 0) {
      const double* L21 = L + (size_t)(r0 + s) * ldl + r0;
      double* X11 = Xc + (size_t)r0 * S + r0;
      double* X22 = Xc + (size_t)(r0 + s) * S + (r0 + s);
      double* X21 = Xc + (size_t)(r0 + s) * S + r0;
      FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, L21, ldl, X11, S, T, (int)s, (int)n2, (int)s, (int)s, 1.0, 0.0,
                                  TRI_LO_J));
      FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, X22, S, T, (int)s, X21, S, (int)n2, (int)s, (int)n2, -1.0, 0.0,
                                  TRI_HI_I));
    }
  }
  h->sinv_L = L;
  h->sinv_n = n;
  h->sinv_ld = ldl;
  h->sinv_S = S;
  return FFGP_OK;
}

static int ensure_tsw(ffgp_handle* h, size_t bytes) {
  if (bytes <= h->tsw_bytes) return FFGP_OK;
  if (h->tsw) {
    hipStreamSynchronize(h->stream);
    hipFree(h->tsw);
  }
  h->tsw = nullptr;
  h->tsw_bytes = 0;
  const size_t gran = (size_t)16 << 20;
  const size_t want = (bytes + gran - 1) / gran * gran;
  if (hipMalloc(&h->tsw, want) != hipSuccess) return FFGP_ERR_ALLOC;
  h->tsw_bytes = want;
  ++h->alloc_epoch;
  return FFGP_OK;
}

static inline bool use_super(const ffgp_handle* h, int n) {
  return h->super_block > 0 && !h->use_naive && n >= h->super_min_n && n > h->super_block;
}

// transposed = 0: B <- L^-1 B (top-down);  1: B <- L^-T B (bottom-up).  V_b = Xinv_b(^T) B_b goes to the staging buffer (it
// cannot be formed in place: every row of B_b feeds every row of V_b), the update of the remaining rows reads it from there.
static int trsm_super(ffgp_handle* h, const double* L, int n, int ldl, double* B, int nrhs, int ldb, int transposed) {
  FFGP_CHECK(ensure_sinv_for(h, L, n, ldl));
  const int S = h->super_block;
  const int nsb = (n + S - 1) / S;
  const int ldv = (nrhs + 1) & ~1;   // even: 16-byte operand loads
  FFGP_CHECK(ensure_tsw(h, (size_t)n * ldv * sizeof(double)));
  double* V = h->tsw;
  // skinny right-hand sides go to the matrix-vector kernels, which take no k clipping: the store holds true zeros above
  // the diagonal of every super-block, so the unclipped product is the same number
  // (the same holds for right-hand sides narrow enough for the split-K path: <= 64 output tiles per super-block)
  const bool thin = (h->skinny_max_n > 0 && nrhs <= h->skinny_max_n && nrhs <= 8) || (h->splitk_min_k > 0 && S >= h->splitk_min_k && (S / 64) * ((nrhs + 63) / 64) <= 64);
  for (int i = 0; i < nsb; ++i) {
    const int sb = transposed ? nsb - 1 - i : i;
    const int r0 = sb * S;
    const int rb = min(S, n - r0);
    const double* Xs = h->sinv + (size_t)sb * S * S + (size_t)sb * S;   // band addressing, ld = S
    double* Bb = B + (size_t)r0 * ldb;
    double* Vb = V + (size_t)r0 * ldv;
    if (!transposed) {
      // V_b = Xinv_b B_b  (Xinv_b lower: k ends at the tile row)
      FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, Xs, S, Bb, ldb, Vb, ldv, rb, nrhs, rb, 1.0, 0.0, thin ? 0 : TRI_HI_I));
      const int below = n - (r0 + rb);
      if (below > 0)
        FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, L + (size_t)(r0 + rb) * ldl + r0, ldl, Vb, ldv,
                                    B + (size_t)(r0 + rb) * ldb, ldb, below, nrhs, rb, -1.0, 1.0));
    } else {
      // V_b = Xinv_b^T B_b  (stored k x m, m contiguous; upper triangular: k starts at the tile row)
      FFGP_CHECK(ffgp_gemm_launch(h, OP_MNMAJOR, OP_MNMAJOR, TILES_FULL, 0, Xs, S, Bb, ldb, Vb, ldv, rb, nrhs, rb, 1.0, 0.0, thin ? 0 : TRI_LO_I));
      if (r0 > 0)
        FFGP_CHECK(ffgp_gemm_launch(h, OP_MNMAJOR, OP_MNMAJOR, TILES_FULL, 0, L + (size_t)r0 * ldl, ldl, Vb, ldv, B, ldb, r0, nrhs,
                                    rb, -1.0, 1.0));
    }
  }
  FFGP_HIP(hipMemcpy2DAsync(B, (size_t)ldb * sizeof(double), V, (size_t)ldv * sizeof(double), (size_t)nrhs * sizeof(double), n,
                            hipMemcpyDeviceToDevice, h->stream));
  return FFGP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// B <- L^-1 B   (forward, blocked by 128):  B_b <- Dinv_b B_b ;  B[below] -= L[below, b] B_b
// ------------------------------------------------------------------------------------------------------------
int ffgp_trsm_lower_impl(ffgp_handle* h, const double* L, int n, int ldl, double* B, int nrhs, int ldb) {
  if (n <= 0 || nrhs <= 0) return FFGP_OK;
  if (!L || !B || ldl < n || ldb < nrhs) return FFGP_ERR_ARG;
  if (use_super(h, n)) return trsm_super(h, L, n, ldl, B, nrhs, ldb, 0);
  FFGP_CHECK(ensure_dinv_for(h, L, n, ldl));
  for (int r0 = 0; r0 < n; r0 += NB) {
    const int rb = min(NB, n - r0);
    double* Bb = B + (size_t)r0 * ldb;
    const double* Db = h->dinv + (size_t)(r0 / NB) * NB * NB;
    // in place: a single tile row (m = rb <= 128); every workgroup reads/writes only its own column tile
    FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, Db, NB, Bb, ldb, Bb, ldb, rb, nrhs, rb, 1.0, 0.0, 0,
                                ALIAS_B));
    const int below = n - (r0 + rb);
    if (below > 0)
      FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, L + (size_t)(r0 + rb) * ldl + r0, ldl, Bb, ldb,
                                  B + (size_t)(r0 + rb) * ldb, ldb, below, nrhs, rb, -1.0, 1.0));
  }
  return FFGP_OK;
}

// B <- L^-T B  (backward):  B_b <- Dinv_b^T B_b ;  B[above] -= L[b, above]^T B_b
int ffgp_trsm_lower_t_impl(ffgp_handle* h, const double* L, int n, int ldl, double* B, int nrhs, int ldb) {
  if (n <= 0 || nrhs <= 0) return FFGP_OK;
  if (!L || !B || ldl < n || ldb < nrhs) return FFGP_ERR_ARG;
  if (use_super(h, n)) return trsm_super(h, L, n, ldl, B, nrhs, ldb, 1);
  FFGP_CHECK(ensure_dinv_for(h, L, n, ldl));
  const int nblk = (n + NB - 1) / NB;
  for (int b = nblk - 1; b >= 0; --b) {
    const int r0 = b * NB;
    const int rb = min(NB, n - r0);
    double* Bb = B + (size_t)r0 * ldb;
    const double* Db = h->dinv + (size_t)b * NB * NB;
    // op(A) = Dinv_b^T: stored k x m with m contiguous = MN-major
    FFGP_CHECK(ffgp_gemm_launch(h, OP_MNMAJOR, OP_MNMAJOR, TILES_FULL, 0, Db, NB, Bb, ldb, Bb, ldb, rb, nrhs, rb, 1.0, 0.0, 0,
                                ALIAS_B));
    if (r0 > 0)  // op(A) = L[b-rows, 0:r0]^T : stored k(=rb) x m(=r0), MN-major
      FFGP_CHECK(ffgp_gemm_launch(h, OP_MNMAJOR, OP_MNMAJOR, TILES_FULL, 0, L + (size_t)r0 * ldl, ldl, Bb, ldb, B, ldb, r0,
                                  nrhs, rb, -1.0, 1.0));
  }
  return FFGP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// X = L^-1 (lower, zeros above the diagonal), recursive:  inv([[L11,0],[L21,L22]]) = [[X11,0],[-X22 L21 X11, X22]]
// ------------------------------------------------------------------------------------------------------------
__global__ void ffgp_copy_block_kernel(const double* __restrict__ src, int lds_, double* __restrict__ dst, int ldd,
                                       int rows, int cols) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < rows * cols) {
    const int r = idx / cols, c = idx % cols;
    dst[(size_t)r * ldd + c] = src[(size_t)r * lds_ + c];
  }
}

// batched copy of the inverted diagonal blocks into X's diagonal
// (full: also write zeros above the diagonal of every block -- for callers that do not zero X first)
__global__ void ffgp_copy_dinv_kernel(const double* __restrict__ dinv, double* __restrict__ X, int ldx, int n, int full) {
  const int b = blockIdx.y;
  const int r0 = b * NB;
  const int nb = min(NB, n - r0);
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < NB * NB) {
    const int r = idx >> 7, c = idx & 127;
    if (r < nb && c <= r) X[(size_t)(r0 + r) * ldx + r0 + c] = dinv[(size_t)b * NB * NB + idx];
    else if (full && r < nb && c < nb) X[(size_t)(r0 + r) * ldx + r0 + c] = 0.0;
  }
}

// X (n x n, ldx) <- L^-1; the strictly-upper triangle of X is zeroed.  Workspace T: >= n*n/4 + n*128 doubles.
// Bottom-up by doubling: at level s every aligned pair of inverted s x s diagonal blocks is merged,
//   inv([[L11,0],[L21,L22]]) = [[X11,0],[-X22 (L21 X11), X22]],
// all full pairs of a level in ONE batched launch per product (2 launches per level instead of 2 per pair).
int ffgp_lauum_impl(ffgp_handle* h, const double* X, int n, int ldx, double* S, int lds_);
static int trtri_levels(ffgp_handle* h, const double* L, int n, int ldl, double* X, int ldx, double* T) {
  for (long s = NB; s < n; s *= 2) {
    const int full = (int)(n / (2 * s));          // pairs with both halves complete
    const long strideL = 2 * s * (long)ldl + 2 * s, strideX = 2 * s * (long)ldx + 2 * s;
    if (full > 0) {
      const double* L21 = L + (size_t)s * ldl;
      double* X11 = X;
      double* X22 = X + (size_t)s * ldx + s;
      double* X21 = X + (size_t)s * ldx;
      // T = L21 * X11   (X11 lower: k starts at the tile column);   X21 = -X22 * T   (X22 lower: k ends at the tile row)
      FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, L21, ldl, X11, ldx, T, (int)s, (int)s, (int)s, (int)s,
                                  1.0, 0.0, TRI_LO_J, ALIAS_NONE, full, strideL, strideX, s * s));
      FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, X22, ldx, T, (int)s, X21, ldx, (int)s, (int)s, (int)s,
                                  -1.0, 0.0, TRI_HI_I, ALIAS_NONE, full, strideX, s * s, strideX));
    }
    const long r0 = (long)full * 2 * s;           // ragged last pair: first half complete, second half partial
    const long n2 = n - r0 - s;
    if (n2 > 0) {
      const double* L21 = L + (size_t)(r0 + s) * ldl + r0;
      double* X11 = X + (size_t)r0 * ldx + r0;
      double* X22 = X + (size_t)(r0 + s) * ldx + (r0 + s);
      double* X21 = X + (size_t)(r0 + s) * ldx + r0;
      FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, L21, ldl, X11, ldx, T, (int)s, (int)n2, (int)s, (int)s,
                                  1.0, 0.0, TRI_LO_J));
      FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, X22, ldx, T, (int)s, X21, ldx, (int)n2, (int)s, (int)n2,
                                  -1.0, 0.0, TRI_HI_I));
    }
  }
  return FFGP_OK;
}

int ffgp_trtri_impl(ffgp_handle* h, const double* L, int n, int ldl, double* X, int ldx, double* T) {
  FFGP_CHECK(ensure_dinv_for(h, L, n, ldl));
  const int nblk = (n + NB - 1) / NB;
  if (n <= NB) {      // one diagonal block: the copy writes its zeros above the diagonal itself (a launch less per small-N step)
    hipLaunchKernelGGL(ffgp_copy_dinv_kernel, dim3(NB * NB / 256, nblk), dim3(256), 0, h->stream, h->dinv, X, ldx, n, 1);
    return FFGP_OK;
  }
  FFGP_CHECK(ffgp_zero_async(h, X, (size_t)n * ldx * sizeof(double)));
  hipLaunchKernelGGL(ffgp_copy_dinv_kernel, dim3(NB * NB / 256, nblk), dim3(256), 0, h->stream, h->dinv, X, ldx, n, 0);
  return trtri_levels(h, L, n, ldl, X, ldx, T);
}

// X_f <- L_f^-1 and S_f <- X_f^T X_f for the F blocks of a shared chain in ONE sequence of launches (outer batch, gridDim.z = F;
// ffgp_handle::ob_*): every launch decides its tile shapes as the single block's launch does, so each block's bits are those of
// ffgp_trtri_impl + ffgp_lauum_impl on it alone.  dinv0 / sD: the F stores of inverted diagonal blocks the batched factorisation left.
int ffgp_trtri_lauum_ob(ffgp_handle* h, int F, const double* L0, long sL, int n, int ldl, double* X0, long sX, int ldx, double* T0, long sT,
                        double* S0, long sS, int lds_, const double* dinv0, long sD) {
  FFGP_CHECK(ffgp_zero_async(h, X0, (size_t)F * sX * sizeof(double)));
  const int nblk = (n + NB - 1) / NB;
  for (int f = 0; f < F; ++f)
    hipLaunchKernelGGL(ffgp_copy_dinv_kernel, dim3(NB * NB / 256, nblk), dim3(256), 0, h->stream, dinv0 + (size_t)f * sD, X0 + (size_t)f * sX,
                       ldx, n, 0);
  h->ob_n = 4;
  h->ob_rng[0] = {L0, L0 + (size_t)F * sL, sL};
  h->ob_rng[1] = {X0, X0 + (size_t)F * sX, sX};
  h->ob_rng[2] = {T0, T0 + (size_t)F * sT, sT};
  h->ob_rng[3] = {S0, S0 + (size_t)F * sS, sS};
  h->ob_F = F;
  int rc = trtri_levels(h, L0, n, ldl, X0, ldx, T0);
  if (rc == FFGP_OK) rc = ffgp_lauum_impl(h, X0, n, ldx, S0, lds_);
  h->ob_F = 0;
  return rc;
}

// The same inverse in two parts, split at column n1 (a power-of-two multiple of 128, n1 < n <= 2 n1 -- the top level's own split):
//   head: everything that only needs the factor's first n1 columns -- X11 = L11^-1 and Ttop = L21 X11 (3/4 of the flops when
//         n = 2 n1).  Those columns are final long before the factorisation ends, so the head runs on a third stream UNDER the
//         factorisation's chain-bound tail (nlml_fused_enqueue);
//   tail: X22 = L22^-1 and X21 = -X22 Ttop, after the factorisation.
// The head reads the store of inverted diagonal blocks while the factorisation is still appending to it: blocks < n1 / 128 only.
int ffgp_trtri_head(ffgp_handle* h, const double* L, int n, int ldl, double* X, int ldx, double* T, double* Ttop, int n1) {
  // Every consumer of this X (the levels below, the tail, LAUUM, the gradient tiles) reads it tile-wise in the lower triangle with
  // triangular hints: tiles wholly above the diagonal are never touched, so only the diagonal blocks' own upper parts need zeros --
  // written by the copy kernel -- instead of a 2 GB fill per step at N = 16384 (option "trtri_fill" = 1 restores the fill; = 2 fills with
  // NaN instead, the test that nothing reads up there: loss and gradients stay bit-identical).  C3 with gradients 78.03 -> 77.82 ms.
  const int fill = h->trtri_fill;
  if (fill == 1) FFGP_CHECK(ffgp_zero_async(h, X, (size_t)n * ldx * sizeof(double)));
  if (fill == 2) FFGP_HIP(hipMemsetAsync(X, 0xFF, (size_t)n * ldx * sizeof(double), h->stream));   // test mode: NaN everywhere -- a consumer that read above the diagonal would show it
  hipLaunchKernelGGL(ffgp_copy_dinv_kernel, dim3(NB * NB / 256, n1 / NB), dim3(256), 0, h->stream, h->dinv, X, ldx, n1, fill == 1 ? 0 : 1);
  FFGP_CHECK(trtri_levels(h, L, n1, ldl, X, ldx, T));
  const int n2 = n - n1;
  return ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, L + (size_t)n1 * ldl, ldl, X, ldx, Ttop, n1, n2, n1, n1, 1.0, 0.0, TRI_LO_J);
}

int ffgp_trtri_tail(ffgp_handle* h, const double* L, int n, int ldl, double* X, int ldx, double* T, const double* Ttop, int n1) {
  FFGP_CHECK(ensure_dinv_for(h, L, n, ldl));
  const int n2 = n - n1;
  double* X22 = X + (size_t)n1 * ldx + n1;
  hipLaunchKernelGGL(ffgp_copy_dinv_kernel, dim3(NB * NB / 256, (n2 + NB - 1) / NB), dim3(256), 0, h->stream,
                     h->dinv + (size_t)(n1 / NB) * NB * NB, X22, ldx, n2, h->trtri_fill == 1 ? 0 : 1);
  FFGP_CHECK(trtri_levels(h, L + (size_t)n1 * ldl + n1, n2, ldl, X22, ldx, T));
  return ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, X22, ldx, Ttop, n1, X + (size_t)n1 * ldx, ldx, n2, n1, n2, -1.0, 0.0,
                          TRI_HI_I);
}

// S (lower triangle, n x n, lds) <- X^T X  for lower-triangular X  (= Sigma^-1 when X = L^-1)
int ffgp_lauum_impl(ffgp_handle* h, const double* X, int n, int ldx, double* S, int lds_) {
  // op(A) = X^T, op(B) = X: both stored k x (m|n) = MN-major; X[k][i] = 0 for k < i -> k starts at the tile row
  return ffgp_gemm_launch(h, OP_MNMAJOR, OP_MNMAJOR, TILES_LOWER, 0, X, ldx, X, ldx, S, lds_, n, n, n, 1.0, 0.0, TRI_LO_I);
}

// ------------------------------------------------------------------------------------------------------------
// reductions:  ss = sum over the (rows x cols, ld) region of M^2 ; ld = sum_i log L_ii ; then the NLL formula
// ------------------------------------------------------------------------------------------------------------
#define RED_BLOCKS 512

// sum of squares of the entries e0, e0 + stride, e0 + 2 stride, ... (row-major rows x cols view of M, leading dimension ldm), added in
// that order (one multiply-add chain per thread: the value does not depend on how the loads are issued).  Eight loads in flight and no
// division inside the loop: with one load and a 64-bit division per entry the stage ran at 1.9 TB/s (140 us for the 8192 x 4096
// right-hand sides of one config-5 block, at the end of its forward pass).
__device__ __forceinline__ double red_sumsq(const double* __restrict__ M, const long total, const int cols, const int ldm, const long e0,
                                            const long stride) {
  double ss = 0.0;
  if (e0 >= total) return ss;
  long r = e0 / cols, c = e0 - r * cols;
  const long dr = stride / cols, dc = stride - dr * cols;
  long e = e0;
  for (; e + 7 * stride < total; e += 8 * stride) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      v[u] = M[r * ldm + c];
      r += dr;
      c += dc;
      if (c >= cols) {
        c -= cols;
        r += 1;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) ss = __builtin_fma(v[u], v[u], ss);
  }
  for (; e < total; e += stride) {
    const double v = M[r * ldm + c];
    r += dr;
    c += dc;
    if (c >= cols) {
      c -= cols;
      r += 1;
    }
    ss = __builtin_fma(v, v, ss);
  }
  return ss;
}

// (fin_out != null and a ONE-block launch: the block finishes the value itself -- stage 2 of a single partial sum is that sum, so the
//  bits are those of the two-launch form; a launch saved on every small problem, d * n <= 256)
__global__ __launch_bounds__(256) void ffgp_reduce_stage1(const double* __restrict__ M, int rows, int cols, int ldm,
                                                          const double* __restrict__ L, int n, int ldl,
                                                          double* __restrict__ partial, double* __restrict__ fin_out, int fin_d,
                                                          double fin_pi, double* __restrict__ fin_aux) {
  __shared__ double r1[4], r2[4];
  double lg = 0.0;
  const long total = (long)rows * cols;
  double ss = red_sumsq(M, total, cols, ldm, (long)blockIdx.x * 256 + threadIdx.x, (long)gridDim.x * 256);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) lg += log(L[(size_t)i * ldl + i]);
  for (int o = 32; o > 0; o >>= 1) {
    ss += __shfl_down(ss, o);
    lg += __shfl_down(lg, o);
  }
  if ((threadIdx.x & 63) == 0) {
    r1[threadIdx.x >> 6] = ss;
    r2[threadIdx.x >> 6] = lg;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double ssb = r1[0] + r1[1] + r1[2] + r1[3], lgb = r2[0] + r2[1] + r2[2] + r2[3];
    partial[blockIdx.x] = ssb;
    partial[RED_BLOCKS + blockIdx.x] = lgb;
    if (fin_out && gridDim.x == 1) {
      // what ffgp_reduce_stage2 computes from one partial: the wave sums of (ssb, 0, 0, ...) and the 4-term sum of (ssb, 0, 0, 0)
      const double ss2 = ((ssb + 0.0) + 0.0) + 0.0, lg2 = ((lgb + 0.0) + 0.0) + 0.0;
      fin_out[0] = 0.5 * ss2 + (double)fin_d * lg2 + 0.5 * (double)n * (double)fin_d * log(2.0 * fin_pi);
      if (fin_aux) {
        fin_aux[0] = ss2;
        fin_aux[1] = lg2;
      }
    }
  }
}

__global__ __launch_bounds__(256) void ffgp_reduce_stage2(const double* __restrict__ partial, int nblocks, int variant,
                                                          int n, int d, double pi_const, double* __restrict__ out,
                                                          double* __restrict__ aux) {
  __shared__ double r1[4], r2[4];
  double ss = 0.0, lg = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) {
    ss += partial[i];
    lg += partial[RED_BLOCKS + i];
  }
  for (int o = 32; o > 0; o >>= 1) {
    ss += __shfl_down(ss, o);
    lg += __shfl_down(lg, o);
  }
  if ((threadIdx.x & 63) == 0) {
    r1[threadIdx.x >> 6] = ss;
    r2[threadIdx.x >> 6] = lg;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    ss = r1[0] + r1[1] + r1[2] + r1[3];
    lg = r2[0] + r2[1] + r2[2] + r2[3];
    // V1: 1/2 ss + d*lg + 1/2 n d log(2 pi~)      V2 (-LL): 1/2 (ss + 2 d lg + n d log(2 pi))  -- same expression
    out[0] = 0.5 * ss + (double)d * lg + 0.5 * (double)n * (double)d * log(2.0 * pi_const);
    if (aux) {
      aux[0] = ss;
      aux[1] = lg;
    }
    (void)variant;
  }
}

// the same two stages for up to FFGP_MULTI_MAX independent problems per launch (the members of a shared-chain batch: F x 2 tiny
// launches otherwise); every member keeps its own block count, so its sums are grouped exactly as in its single call
__global__ __launch_bounds__(256) void ffgp_reduce_stage1_multi(ffgp_multi_red q, double* __restrict__ partial_all) {
  const int z = blockIdx.y, nb = q.blocks[z];
  if ((int)blockIdx.x >= nb) return;
  const double* __restrict__ M = q.M[z];
  const double* __restrict__ L = q.L[z];
  const int rows = q.rows[z], cols = q.cols[z], ldm = q.ldm[z], n = q.n[z], ldl = q.ldl[z];
  double* __restrict__ partial = partial_all + (size_t)(q.first + z) * 2 * RED_BLOCKS;
  __shared__ double r1[4], r2[4];
  double lg = 0.0;
  const long total = (long)rows * cols;
  double ss = red_sumsq(M, total, cols, ldm, (long)blockIdx.x * 256 + threadIdx.x, (long)nb * 256);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += nb * 256) lg += log(L[(size_t)i * ldl + i]);
  for (int o = 32; o > 0; o >>= 1) {
    ss += __shfl_down(ss, o);
    lg += __shfl_down(lg, o);
  }
  if ((threadIdx.x & 63) == 0) {
    r1[threadIdx.x >> 6] = ss;
    r2[threadIdx.x >> 6] = lg;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = r1[0] + r1[1] + r1[2] + r1[3];
    partial[RED_BLOCKS + blockIdx.x] = r2[0] + r2[1] + r2[2] + r2[3];
  }
}

__global__ __launch_bounds__(256) void ffgp_reduce_stage2_multi(ffgp_multi_red q, const double* __restrict__ partial_all) {
  const int z = blockIdx.x;
  const double* __restrict__ partial = partial_all + (size_t)(q.first + z) * 2 * RED_BLOCKS;
  const int nblocks = q.blocks[z];
  __shared__ double r1[4], r2[4];
  double ss = 0.0, lg = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) {
    ss += partial[i];
    lg += partial[RED_BLOCKS + i];
  }
  for (int o = 32; o > 0; o >>= 1) {
    ss += __shfl_down(ss, o);
    lg += __shfl_down(lg, o);
  }
  if ((threadIdx.x & 63) == 0) {
    r1[threadIdx.x >> 6] = ss;
    r2[threadIdx.x >> 6] = lg;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    ss = r1[0] + r1[1] + r1[2] + r1[3];
    lg = r2[0] + r2[1] + r2[2] + r2[3];
    double v = 0.5 * ss + (double)q.d[z] * lg + 0.5 * (double)q.n[z] * (double)q.d[z] * log(2.0 * q.pi_const[z]);
    if (q.scale[z] != 1.0) v *= q.scale[z];      // (the output scale of the raw-parameter calls, a separate launch in the single call)
    q.out[z][0] = v;
  }
}

// partial_ws: F x 2 x RED_BLOCKS doubles; scale[f] multiplies member f's value (1.0: none)
int ffgp_nll_reduce_multi(ffgp_handle* h, int F, const double* const* L, const int* n, const int* ldl, const double* const* M,
                          const int* rows, const int* cols, const int* ldm, const int* d, const double* pi_const, const double* scale,
                          double* const* out, double* partial_ws) {
  for (int f0 = 0; f0 < F; f0 += FFGP_MULTI_MAX) {
    const int cnt = F - f0 < FFGP_MULTI_MAX ? F - f0 : FFGP_MULTI_MAX;
    ffgp_multi_red q;
    q.first = f0;
    int maxb = 1;
    for (int z = 0; z < FFGP_MULTI_MAX; ++z) {
      const int f = f0 + (z < cnt ? z : 0);
      long total = (long)rows[f] * cols[f];
      int blocks = (int)((total + 255) / 256);
      if (blocks > RED_BLOCKS) blocks = RED_BLOCKS;
      if (blocks < 1) blocks = 1;
      q.M[z] = M[f]; q.L[z] = L[f]; q.rows[z] = rows[f]; q.cols[z] = cols[f]; q.ldm[z] = ldm[f]; q.n[z] = n[f]; q.ldl[z] = ldl[f];
      q.d[z] = d[f]; q.pi_const[z] = pi_const[f]; q.scale[z] = scale ? scale[f] : 1.0; q.out[z] = out[f];
      q.blocks[z] = blocks;
      if (z < cnt && blocks > maxb) maxb = blocks;
    }
    hipLaunchKernelGGL(ffgp_reduce_stage1_multi, dim3(maxb, cnt), dim3(256), 0, h->stream, q, partial_ws);
    hipLaunchKernelGGL(ffgp_reduce_stage2_multi, dim3(cnt), dim3(256), 0, h->stream, q, partial_ws);
  }
  if (hipGetLastError() != hipSuccess) return FFGP_ERR_HIP;
  return FFGP_OK;
}

int ffgp_nll_reduce_impl(ffgp_handle* h, int variant, const double* L, int n, int ldl, const double* M, int rows, int cols,
                         int ldm, int d, double pi_const, double* out_dev) {
  if (!L || !M || !out_dev) return FFGP_ERR_ARG;
  double* partial = h->d_scal + 64;  // 2*RED_BLOCKS doubles reserved behind the scalar scratch
  long total = (long)rows * cols;
  int blocks = (int)((total + 255) / 256);
  if (blocks > RED_BLOCKS) blocks = RED_BLOCKS;
  if (blocks < 1) blocks = 1;
  if (blocks == 1) {
    hipLaunchKernelGGL(ffgp_reduce_stage1, dim3(1), dim3(256), 0, h->stream, M, rows, cols, ldm, L, n, ldl, partial, out_dev, d, pi_const,
                       h->d_scal + 8);
  } else {
    hipLaunchKernelGGL(ffgp_reduce_stage1, dim3(blocks), dim3(256), 0, h->stream, M, rows, cols, ldm, L, n, ldl, partial, (double*)nullptr, 0,
                       0.0, (double*)nullptr);
    hipLaunchKernelGGL(ffgp_reduce_stage2, dim3(1), dim3(256), 0, h->stream, partial, blocks, variant, n, d, pi_const, out_dev,
                       h->d_scal + 8);
  }
  if (hipGetLastError() != hipSuccess) return FFGP_ERR_HIP;
  return FFGP_OK;
}
