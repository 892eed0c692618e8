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
#define DLD 130                 // LDS leading dimension of the 128x128 image (conflict-free MFMA operand reads)
#define DIAG_LDS_DOUBLES (NB * DLD + 8 * 256 + 128)
#define DIAG_LDS_BYTES (DIAG_LDS_DOUBLES * 8)

__device__ __forceinline__ double readlane_d(double x, int l) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
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

// ------------------------------------------------------------------------------------------------------------
// potrf_diag128: factor one diagonal block (nb <= 128 valid rows/cols, identity-padded) and invert it.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ffgp_potrf_diag128(double* __restrict__ A, int lda, int nb,
                                                          double* __restrict__ Dinv, int* info, int row_base,
                                                          int do_factor) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* S = lds;                    // [128][DLD]
  double* Dg = lds + NB * DLD;        // [8][16][16] inverses of the 16x16 diagonal blocks
  double* rd = Dg + 8 * 256;          // [128] reciprocals of the diagonal of L
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- phase 0: load (lower part; identity padding beyond nb; zeros above the diagonal)
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int r = idx >> 7, c = idx & 127;
    double v = (r == c) ? 1.0 : 0.0;
    if (r < nb && c <= r) v = A[(size_t)r * lda + c];
    S[r * DLD + c] = v;
  }
  __syncthreads();

  if (do_factor) {
    // ---- phase 1: left-looking factorisation over 16-column blocks
    for (int jj = 0; jj < 8; ++jj) {
      // (a) S[i][jj] -= sum_{p<jj} S[i][p] * S[jj][p]^T   for block rows i = jj..7 (MFMA)
      if (jj > 0) {
        for (int i = jj + wave; i < 8; i += 4) {
          d4_t acc = {0.0, 0.0, 0.0, 0.0};
          for (int p = 0; p < jj; ++p)
            mma16<true>(acc, S + (i * 16) * DLD + p * 16, DLD, S + (jj * 16) * DLD + p * 16, DLD, lane);
#pragma unroll
          for (int r = 0; r < 4; ++r) S[(i * 16 + (lane >> 4) + 4 * r) * DLD + jj * 16 + (lane & 15)] -= acc[r];
        }
      }
      __syncthreads();
      // (b) wave 0: 16x16 Cholesky, one lane per row (lanes >= 16 shadow lanes & 15), column broadcasts by readlane
      if (wave == 0) {
        const int i = lane & 15;
        double v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = S[(jj * 16 + i) * DLD + jj * 16 + c];
        int bad = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          double d = readlane_d(v[j], j);
          if (!(d > 0.0)) {
            if (!bad) bad = j + 1;
            d = 1.0;
          }
          const double rs = rsqrt_nr(d);
          const double lij = (i == j) ? d * rs : v[j] * rs;
          v[j] = lij;
          if (lane == j) rd[jj * 16 + j] = rs;
#pragma unroll
          for (int c = j + 1; c < 16; ++c) {
            const double lcj = readlane_d(lij, c);
            v[c] -= lij * lcj;
          }
        }
        if (lane < 16) {
#pragma unroll
          for (int c = 0; c < 16; ++c) S[(jj * 16 + i) * DLD + jj * 16 + c] = (c <= i) ? v[c] : 0.0;
        }
        if (bad && lane == 0 && (jj * 16 + bad) <= nb) atomicCAS(info, 0, row_base + jj * 16 + bad);
      }
      __syncthreads();
      // (c) rows below: x * L_jj^T = b by substitution, one lane per row; L_jj entries are LDS broadcasts
      {
        const int nrows = NB - (jj + 1) * 16;
        if (tid < nrows) {
          const int row = (jj + 1) * 16 + tid;
          double* pr = S + row * DLD + jj * 16;
          const double* Lj = S + (jj * 16) * DLD + jj * 16;
          double x[16];
#pragma unroll
          for (int c = 0; c < 16; ++c) x[c] = pr[c];
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            double s = x[c];
#pragma unroll
            for (int k = 0; k < c; ++k) s -= x[k] * Lj[c * DLD + k];
            x[c] = s * rd[jj * 16 + c];
          }
#pragma unroll
          for (int c = 0; c < 16; ++c) pr[c] = x[c];
        }
      }
      __syncthreads();
    }
    // ---- phase 2: write L (lower part of the valid block)
    for (int idx = tid; idx < NB * NB; idx += 256) {
      const int r = idx >> 7, c = idx & 127;
      if (r < nb && c <= r) A[(size_t)r * lda + c] = S[r * DLD + c];
    }
  } else {
    // inverse-only entry (Dinv refresh for a factor produced elsewhere): reciprocals of the diagonal
    if (tid < NB) rd[tid] = 1.0 / S[tid * DLD + tid];
    __syncthreads();
  }

  // ---- phase 3: inverses of the eight 16x16 diagonal blocks; 16 lanes per block (one per column), 4 blocks per wave
  if (wave < 2) {
    const int jj = wave * 4 + (lane >> 4), c = lane & 15;
    const double* Lj = S + (jj * 16) * DLD + jj * 16;
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      double s = (i == c) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) s -= Lj[i * DLD + k] * x[k];
      x[i] = s * rd[jj * 16 + i];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) Dg[jj * 256 + i * 16 + c] = x[i];
  }
  __syncthreads();
  // diagonal blocks of the image now hold X_jj
  for (int idx = tid; idx < 8 * 256; idx += 256) {
    const int jj = idx >> 8, i = (idx >> 4) & 15, c = idx & 15;
    S[(jj * 16 + i) * DLD + jj * 16 + c] = Dg[idx];
  }
  __syncthreads();

  // ---- phase 4: in-place blocked inversion, block columns right to left:
  //      X[i>j, j] = -X[i>j, i>j] * L[i>j, j] * X_jj
  for (int j = 6; j >= 0; --j) {
    // T_i = L_ij * X_jj  (each wave overwrites only the blocks it read)
    for (int i = j + 1 + wave; i < 8; i += 4) {
      d4_t acc = {0.0, 0.0, 0.0, 0.0};
      mma16<false>(acc, S + (i * 16) * DLD + j * 16, DLD, Dg + j * 256, 16, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) S[(i * 16 + (lane >> 4) + 4 * r) * DLD + j * 16 + (lane & 15)] = acc[r];
    }
    __syncthreads();
    // X_ij = -sum_{k=j+1..i} X_ik * T_k ; results parked in registers until every wave has read T
    d4_t res[2];
    int cnt = 0;
    for (int i = j + 1 + wave; i < 8; i += 4) {
      d4_t acc = {0.0, 0.0, 0.0, 0.0};
      for (int k = j + 1; k <= i; ++k)
        mma16<false>(acc, S + (i * 16) * DLD + k * 16, DLD, S + (k * 16) * DLD + j * 16, DLD, lane);
      if (cnt == 0) res[0] = acc; else res[1] = acc;
      ++cnt;
    }
    __syncthreads();
    cnt = 0;
    for (int i = j + 1 + wave; i < 8; i += 4) {
      const d4_t acc = (cnt == 0) ? res[0] : res[1];
#pragma unroll
      for (int r = 0; r < 4; ++r) S[(i * 16 + (lane >> 4) + 4 * r) * DLD + j * 16 + (lane & 15)] = -acc[r];
      ++cnt;
    }
    __syncthreads();
  }
  // ---- phase 5: write the inverse (dense 128x128, zeros above the diagonal)
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int r = idx >> 7, c = idx & 127;
    Dinv[idx] = (c <= r) ? S[r * DLD + c] : 0.0;
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
static bool g_diag_attr_set = false;

static int launch_diag(ffgp_handle* h, double* Ablk, int lda, int nb, double* Dinv_blk, int row_base, int do_factor) {
  if (!g_diag_attr_set) {
    FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ffgp_potrf_diag128),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, DIAG_LDS_BYTES));
    g_diag_attr_set = true;
  }
  hipLaunchKernelGGL(ffgp_potrf_diag128, dim3(1), dim3(256), DIAG_LDS_BYTES, h->stream, Ablk, lda, nb, Dinv_blk,
                     h->d_info, row_base, do_factor);
  return FFGP_OK;
}

int ffgp_ensure_dinv(ffgp_handle* h, int n) {
  const int nblk = (n + NB - 1) / NB;
  const size_t need = (size_t)nblk * NB * NB * sizeof(double);
  if (need > h->dinv_bytes) {
    if (h->dinv) hipFree(h->dinv);
    h->dinv = nullptr;
    h->dinv_bytes = 0;
    if (hipMalloc(&h->dinv, need) != hipSuccess) return FFGP_ERR_ALLOC;
    h->dinv_bytes = need;
  }
  return FFGP_OK;
}

// (re)build the inverted diagonal blocks for a factor that is already in L (used when a caller hands us a
// factor this handle did not just produce)
int ffgp_refresh_dinv(ffgp_handle* h, const double* L, int n, int ldl) {
  FFGP_CHECK(ffgp_ensure_dinv(h, n));
  const int nblk = (n + NB - 1) / NB;
  if (h->use_naive) {
    hipLaunchKernelGGL(ffgp_dinv_naive, dim3(nblk), dim3(128), 0, h->stream, L, ldl, n, h->dinv);
  } else {
    for (int b = 0; b < nblk; ++b) {
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
  FFGP_HIP(hipMemsetAsync(h->d_info, 0, sizeof(int), h->stream));
  h->dinv_L = nullptr;

  if (h->use_naive) {
    hipLaunchKernelGGL(ffgp_potrf_naive, dim3(1), dim3(256), 0, h->stream, A, lda, n, h->d_info);
    FFGP_CHECK(ffgp_refresh_dinv(h, A, n, lda));
    if (mtot > n) hipLaunchKernelGGL(ffgp_trsm_rows_naive, dim3(mtot - n), dim3(64), 0, h->stream, A, lda, n);
  } else {
    const int NB1 = h->nb_outer;
    for (int k0 = 0; k0 < n; k0 += NB1) {
      const int w1 = min(NB1, n - k0);
      const int pend = k0 + w1;  // end column of this outer panel
      for (int j0 = k0; j0 < pend; j0 += NB) {
        const int jb = min(NB, n - j0);
        double* Ajj = A + (size_t)j0 * lda + j0;
        double* Dj = h->dinv + (size_t)(j0 / NB) * NB * NB;
        FFGP_CHECK(launch_diag(h, Ajj, lda, jb, Dj, j0, 1));
        const int mrows = mtot - (j0 + jb);
        if (mrows > 0) {
          double* A21 = A + (size_t)(j0 + jb) * lda + j0;
          // TRSM as GEMM: A21 <- A21 * Dj^T (in place: one column tile, each workgroup rewrites only rows it read)
          FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_KMAJOR, TILES_FULL, 0, A21, lda, Dj, NB, A21, lda, mrows, jb, jb,
                                      1.0, 0.0));
          const int wrem = pend - (j0 + jb);
          if (wrem > 0) {
            double* C = A + (size_t)(j0 + jb) * lda + (j0 + jb);
            FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_KMAJOR, TILES_LOWER, 0, A21, lda, A21, lda, C, lda, mrows, wrem,
                                        jb, -1.0, 1.0));
          }
        }
      }
      const int mt = n - pend;  // trailing columns; trailing rows include the passenger rows
      if (mt > 0) {
        double* P = A + (size_t)pend * lda + k0;
        double* C = A + (size_t)pend * lda + pend;
        FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_KMAJOR, TILES_LOWER, 1, P, lda, P, lda, C, lda, mtot - pend, mt, w1,
                                    -1.0, 1.0));
      }
    }
    h->dinv_L = A;
    h->dinv_n = n;
    h->dinv_ld = lda;
  }
  if (!sync_info) return FFGP_OK;
  FFGP_HIP(hipMemcpyAsync(h->h_info, h->d_info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  FFGP_HIP(hipStreamSynchronize(h->stream));
  return h->h_info[0];
}
