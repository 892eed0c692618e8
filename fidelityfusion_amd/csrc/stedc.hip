// Symmetric eigensolver, stage 3: eigen-decomposition of the tridiagonal matrix by divide and conquer (Cuppen), entirely on
// the device, no host round trip: every shape is fixed by n alone.  (LAPACK's dstedc underneath `torch.linalg.eigh(K)` in the
// HOGP block: FidelityFusion_Models/two_fidelity_models/hogp_simple.py:15-19,97-100.)  CPU restatement: the tests' numpy model eigh_twostage.py
// (stedc, merge_S, secular_root).
//
//   leaves   64 x 64 blocks of the tridiagonal matrix (rank-one corrections |e_b| taken off both sides of every leaf boundary at
//            once) -> the batched LDS Jacobi kernel of eig.hip
//   level l  pairs of solved blocks [off, off + n1), [off + n1, off + n1 + n2) are merged:  diag(d1, d2) + rho z z^T
//     dc_setup    z from the last / first rows of the two eigenvector blocks, rho, tolerances
//     dc_rank     sort d by counting (no assumption that the halves arrive sorted)
//     dc_deflate  one lane per merge walks the sorted list: negligible z components and close pairs (one Givens rotation
//                 each) are deflated exactly as LAPACK's dlaed2 does; the rotations are recorded
//     dc_rotate_s ... and folded into the rows of S (so the old eigenvector block stays block diagonal)
//     dc_secular  one thread per root: the secular equation in the variable shifted to the nearer pole, two-pole rational
//                 ("middle way") steps safeguarded by a bracket -- the root is returned as (origin, offset), so every
//                 difference d_j - lambda_i is formed without cancellation
//     dc_zhat     Gu-Eisenstat: z recomputed from the computed roots, which is what makes the eigenvectors orthogonal to working
//                 precision however clustered the spectrum is
//     dc_norm, dc_rank2, dc_build_s   the dense matrix S = P [V | I] that maps the old basis to the new one, columns already in
//                 ascending order of the merged eigenvalues
//     GEMM        Q_new = Q_old S on the fp64 matrix cores (one batched launch for the level's equal merges, one for a ragged last)
// Kernel matrices deflate massively (their spectrum clusters at 0); nothing here depends on that.
#include "ffgp_internal.h"
#include "syevd_internal.h"

int ffgp_syevj_small_impl(ffgp_handle* h, const double* M, int n, int ldm, int batch, long strideM, double* Q, int ldq, long strideQ,
                          double* evals, long strideE, int descending);

#define DC_EPS 2.220446049250313e-16

struct DcLevel {
  int n;          // order of the whole problem
  int bs;         // size of a full merged block at this level; its first half has bs / 2
  const double* e;
  double* Zc; int ldz;       // eigenvector blocks of the previous level (block diagonal)
  const double* lam_in;      // eigenvalues of the previous level, ascending inside each block
  double* lam_out;
  double* z; double* ds; double* zs;   // [n] each: z (unsorted), sorted d and z (modified by the deflation)
  int* perm;      // sorted position -> index inside the merge
  int* defl;      // per sorted position
  int* nd;        // non-deflated sorted positions first (k of them), deflated ones after, per merge at [off, off + nn)
  int* kcnt;      // [merges] number of non-deflated
  int* nrot;      // [merges]
  int* rotp; int* rotj; double* rotc; double* rots;   // rotations per merge at [off, ...)
  double* rho;    // [merges]  (already doubled: the factor of the normalised z)
  double* tol;    // [merges]
  double* dk; double* zk;    // compacted non-deflated d and z
  int* org; double* mu;      // roots: lambda_i = dk[org_i] + mu_i
  double* zhat; double* vnorm;
  double* lamn;   // merged eigenvalues before the final sort: roots first, deflated after
  int* rank2;     // final column of each
  double* S; int lds;
};

__device__ __forceinline__ void dc_shape(const DcLevel& p, int m, int& off, int& n1, int& n2) {
  off = m * p.bs;
  n1 = min(p.bs / 2, p.n - off);
  n2 = min(p.bs / 2, p.n - off - n1);
  if (n2 < 0) n2 = 0;
}

__global__ __launch_bounds__(256) void dc_setup(DcLevel p) {
  __shared__ double red[256];
  int off, n1, n2;
  dc_shape(p, blockIdx.x, off, n1, n2);
  const int nn = n1 + n2, tid = threadIdx.x;
  double beta = 0.0;
  if (n2 > 0) beta = p.e[off + n1 - 1];
  const double sgn = (beta >= 0.0) ? 1.0 : -1.0;
  const double r2 = 0.70710678118654752440;
  double zmax = 0.0, dmax = 0.0;
  for (int i = tid; i < nn; i += 256) {
    double zv = 0.0;
    if (n2 > 0) zv = (i < n1) ? p.Zc[(size_t)(off + n1 - 1) * p.ldz + off + i] : sgn * p.Zc[(size_t)(off + n1) * p.ldz + off + i];
    zv *= r2;
    p.z[off + i] = zv;
    zmax = fmax(zmax, fabs(zv));
    dmax = fmax(dmax, fabs(p.lam_in[off + i]));
  }
  red[tid] = zmax;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
    __syncthreads();
  }
  zmax = red[0];
  __syncthreads();
  red[tid] = dmax;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
    __syncthreads();
  }
  dmax = red[0];
  if (tid == 0) {
    p.rho[blockIdx.x] = 2.0 * fabs(beta);
    p.tol[blockIdx.x] = 8.0 * DC_EPS * fmax(dmax, zmax);
  }
}

// sorted position of every entry of the merge (ties by index): ds, zs, perm
__global__ __launch_bounds__(256) void dc_rank(DcLevel p) {
  int off, n1, n2;
  dc_shape(p, blockIdx.x, off, n1, n2);
  const int nn = n1 + n2;
  const int i = blockIdx.y * 256 + threadIdx.x;
  if (i >= nn) return;
  const double di = p.lam_in[off + i];
  int r = 0;
  for (int j = 0; j < nn; ++j) {
    const double dj = p.lam_in[off + j];
    r += (dj < di || (dj == di && j < i)) ? 1 : 0;
  }
  p.ds[off + r] = di;
  p.zs[off + r] = p.z[off + i];
  p.perm[off + r] = i;
}

// the sequential part of dlaed2, one lane per merge; chunks of the sorted arrays are staged through LDS by the whole wave
__global__ __launch_bounds__(64) void dc_deflate(DcLevel p) {
  __shared__ double cd[64], cz[64];
  int off, n1, n2;
  dc_shape(p, blockIdx.x, off, n1, n2);
  const int nn = n1 + n2, lane = threadIdx.x;
  const double rho = p.rho[blockIdx.x], tol = p.tol[blockIdx.x];
  // rho * max|z| <= tol: everything deflates
  double zmax = 0.0;
  for (int i = lane; i < nn; i += 64) zmax = fmax(zmax, fabs(p.zs[off + i]));
  for (int o = 32; o > 0; o >>= 1) zmax = fmax(zmax, __shfl_xor(zmax, o));
  const bool all_defl = (rho * zmax <= tol);
  int prev = -1, nr = 0;
  double dprev = 0.0, zprev = 0.0;
  for (int base = 0; base < nn; base += 64) {
    __syncthreads();
    if (base + lane < nn) {
      cd[lane] = p.ds[off + base + lane];
      cz[lane] = p.zs[off + base + lane];
    }
    __syncthreads();
    if (lane == 0) {
      const int cnt = min(64, nn - base);
      for (int t = 0; t < cnt; ++t) {
        const int j = base + t;
        double dj = cd[t], zj = cz[t];
        if (all_defl || rho * fabs(zj) <= tol) {
          p.defl[off + j] = 1;
          continue;
        }
        if (prev >= 0) {
          // close pair?  |t c s| <= tol with c = z_j / tau, s = -z_prev / tau, tau = hypot(z_j, z_prev)  <=>  |t z_j z_prev| <= tol tau^2:
          // the test without the square root and the two divisions -- this lane walks the whole list alone, and nearly every pair fails it
          const double tt = dj - dprev;
          if (fabs(tt * zj * zprev) <= tol * __builtin_fma(zj, zj, zprev * zprev)) {   // rotate z[prev] into z[j], position prev deflates
            const double tau = hypot(zj, zprev);
            const double cc = zj / tau, ss = -zprev / tau;
            zj = tau;
            const double dp = dprev * cc * cc + dj * ss * ss;
            dj = dprev * ss * ss + dj * cc * cc;
            p.ds[off + prev] = dp;
            p.zs[off + prev] = 0.0;
            p.defl[off + prev] = 1;
            p.rotp[off + nr] = prev;
            p.rotj[off + nr] = j;
            p.rotc[off + nr] = cc;
            p.rots[off + nr] = ss;
            ++nr;
          } else {
            p.defl[off + prev] = 0;
          }
        }
        // position j stays a candidate; its final values are written when it is superseded or at the end
        p.ds[off + j] = dj;
        p.zs[off + j] = zj;
        prev = j;
        dprev = dj;
        zprev = zj;
      }
    }
  }
  if (lane == 0) {
    if (prev >= 0) p.defl[off + prev] = 0;
    p.nrot[blockIdx.x] = nr;
  }
  __syncthreads();
  // compaction (stable, the whole wave): non-deflated positions first, deflated after
  int kb = 0;
  for (int pass = 0; pass < 2; ++pass) {
    for (int base = 0; base < nn; base += 64) {
      const int j = base + lane;
      const bool f = (j < nn) && ((p.defl[off + j] != 0) == (pass == 1));
      const unsigned long long mask = __ballot(f);
      const int pos = kb + __popcll(mask & ((1ull << lane) - 1ull));
      if (f) {
        p.nd[off + pos] = j;
        if (pass == 0) {
          p.dk[off + pos] = p.ds[off + j];
          p.zk[off + pos] = p.zs[off + j];
        } else {
          p.lamn[off + pos] = p.ds[off + j];
        }
      }
      kb += __popcll(mask);
    }
    if (pass == 0 && lane == 0) p.kcnt[blockIdx.x] = kb;
  }
}

// The deflation's rotations act on columns of the old eigenvector block: Q_rot = Q_old R_1 R_2 ... R_r.  They are folded into S
// instead (Q_new = Q_old (R_1 (R_2 ... (R_r S)))), last rotation first, on pairs of ROWS of S: one thread per column of S, coalesced,
// and Q_old keeps its block-diagonal form, which halves the merge GEMMs.
__global__ __launch_bounds__(256) void dc_rotate_s(DcLevel p) {
  int off, n1, n2;
  dc_shape(p, blockIdx.x, off, n1, n2);
  const int nn = n1 + n2;
  const int c = blockIdx.y * 256 + threadIdx.x;
  const int nr = p.nrot[blockIdx.x];
  if (c >= nn || nr == 0) return;
  double* col = p.S + (size_t)off * p.lds + off + c;
  for (int t = nr - 1; t >= 0; --t) {
    const size_t ra = (size_t)p.perm[off + p.rotp[off + t]] * p.lds, rb = (size_t)p.perm[off + p.rotj[off + t]] * p.lds;
    const double cs = p.rotc[off + t], sn = p.rots[off + t];
    const double a = col[ra], b = col[rb];
    col[ra] = cs * a - sn * b;
    col[rb] = sn * a + cs * b;
  }
}

__device__ __forceinline__ double dc_small_root(double a, double b, double c) {   // root of a x^2 + b x + c of smaller magnitude
  if (a == 0.0) return (b != 0.0) ? -c / b : HUGE_VAL;
  double disc = b * b - 4.0 * a * c;
  if (disc < 0.0) disc = 0.0;
  const double sq = sqrt(disc);
  const double q = -0.5 * (b + ((b >= 0.0) ? sq : -sq));
  if (q == 0.0) return 0.0;
  const double r1 = c / q, r2 = q / a;
  return (fabs(r1) <= fabs(r2)) ? r1 : r2;
}

template <int CTRL>
__device__ __forceinline__ double dc_dpp_add(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return x + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dc_sum8(double x) {   // sum over the 8 lanes of a root's group, in all of them
  x = dc_dpp_add<0xB1>(x);
  x = dc_dpp_add<0x4E>(x);
  return dc_dpp_add<0x141>(x);
}
__device__ __forceinline__ double dc_rcp(double d) {
  double y = __builtin_amdgcn_rcp(d);
  y = __builtin_fma(y, __builtin_fma(-d, y, 1.0), y);
  y = __builtin_fma(y, __builtin_fma(-d, y, 1.0), y);
  return y;
}

// one root per group of 8 lanes: the poles are dealt round-robin to the lanes, the four sums of an iteration are combined inside
// the group (DPP), every lane of the group runs the same scalar iteration
__global__ __launch_bounds__(256) void dc_secular(DcLevel p) {
  int off, n1, n2;
  dc_shape(p, blockIdx.x, off, n1, n2);
  const int k = p.kcnt[blockIdx.x];
  const int i = blockIdx.y * 32 + (threadIdx.x >> 3);
  const int part = threadIdx.x & 7;
  if (blockIdx.y * 32 >= k) return;          // whole workgroup idle
  const bool live = i < k;                   // idle groups of a live workgroup run along (DPP wants every lane) on root k - 1
  const int ii = live ? i : k - 1;
  const double rho = p.rho[blockIdx.x];
  const double* __restrict__ dk = p.dk + off;
  const double* __restrict__ zk = p.zk + off;
  int o;
  double lo, hi;
  if (k == 1) {
    if (live && part == 0) {
      p.org[off] = 0;
      p.mu[off] = rho * zk[0] * zk[0];
      p.lamn[off] = dk[0] + p.mu[off];
    }
    return;
  }
  if (ii < k - 1) {
    const double di = dk[ii];
    const double mid = 0.5 * (dk[ii + 1] - di);
    double f = 0.0;
    for (int j = part; j < k; j += 8) f = __builtin_fma(zk[j] * zk[j], dc_rcp((dk[j] - di) - mid), f);
    f = 1.0 + rho * dc_sum8(f);
    if (f >= 0.0) {
      o = ii; lo = 0.0; hi = mid;
    } else {
      o = ii + 1; lo = -mid; hi = 0.0;
    }
  } else {
    double sacc = 0.0;
    for (int j = part; j < k; j += 8) sacc = __builtin_fma(zk[j], zk[j], sacc);
    o = k - 1; lo = 0.0; hi = rho * dc_sum8(sacc);
  }
  const double dorg = dk[o];
  double mu = 0.5 * (lo + hi);
  const double sk = sqrt((double)k);
  bool done = false;
  for (int it = 0; it < 100; ++it) {
    double psi = 0.0, phi = 0.0, dpsi = 0.0, dphi = 0.0;
    for (int j = part; j < k; j += 8) {
      const double r = dc_rcp((dk[j] - dorg) - mu);
      const double t = zk[j] * zk[j] * r;
      if (j <= ii) {
        psi += t;
        dpsi = __builtin_fma(t, r, dpsi);
      } else {
        phi += t;
        dphi = __builtin_fma(t, r, dphi);
      }
    }
    psi = dc_sum8(psi); phi = dc_sum8(phi); dpsi = dc_sum8(dpsi); dphi = dc_sum8(dphi);
    if (!done) {
      const double f = 1.0 + rho * (psi + phi);
      const double err = 8.0 * DC_EPS * (1.0 + rho * (fabs(psi) + fabs(phi))) * sk;
      if (fabs(f) <= err) {
        done = true;
      } else {
        if (f > 0.0) hi = mu; else lo = mu;
        double eta;
        if (ii < k - 1) {
          const double a1 = (dk[ii] - dorg) - mu, a2 = (dk[ii + 1] - dorg) - mu;   // a1 < 0 < a2
          const double a = rho * dpsi * a1 * a1, b = rho * dphi * a2 * a2;
          const double c = f - rho * dpsi * a1 - rho * dphi * a2;
          eta = dc_small_root(c, -(c * (a1 + a2) + a + b), c * a1 * a2 + a * a2 + b * a1);
        } else {
          const double a1 = (dk[k - 1] - dorg) - mu;
          const double a = rho * dpsi * a1 * a1;
          const double c = f - rho * dpsi * a1;
          eta = (c != 0.0) ? a1 + a / c : HUGE_VAL;
        }
        double nw = mu + eta;
        if (!(nw > lo && nw < hi)) nw = 0.5 * (lo + hi);   // also catches NaN / inf
        const bool stop = (nw == mu) || (hi - lo <= 2.0 * DC_EPS * fmax(fabs(lo), fabs(hi)));
        mu = nw;
        if (stop) done = true;
      }
    }
    if (__all(done)) break;   // the wave leaves together (the DPP sums need all of its lanes)
  }
  if (live && part == 0) {
    p.org[off + i] = o;
    p.mu[off + i] = mu;
    p.lamn[off + i] = dorg + mu;
  }
}

// zhat_j = sign(z_j) sqrt(| prod_i (lam_i - d_j) / prod_{i != j} (d_i - d_j) | / rho)
__global__ __launch_bounds__(256) void dc_zhat(DcLevel p) {
  int off, n1, n2;
  dc_shape(p, blockIdx.x, off, n1, n2);
  const int k = p.kcnt[blockIdx.x];
  const int j = blockIdx.y * 256 + threadIdx.x;
  if (j >= k) return;
  const double* __restrict__ dk = p.dk + off;
  const double dj = dk[j];
  double prod = 1.0;
  for (int i = 0; i < k; ++i) {
    const double num = (dk[p.org[off + i]] - dj) + p.mu[off + i];
    prod *= (i == j) ? num : num / (dk[i] - dj);
  }
  const double v = sqrt(fabs(prod / p.rho[blockIdx.x]));
  p.zhat[off + j] = (p.zk[off + j] >= 0.0) ? v : -v;
}

__global__ __launch_bounds__(256) void dc_norm(DcLevel p) {
  int off, n1, n2;
  dc_shape(p, blockIdx.x, off, n1, n2);
  const int k = p.kcnt[blockIdx.x];
  const int i = blockIdx.y * 256 + threadIdx.x;
  if (i >= k) return;
  const double* __restrict__ dk = p.dk + off;
  const double dorg = dk[p.org[off + i]], mu = p.mu[off + i];
  double s = 0.0;
  for (int j = 0; j < k; ++j) {
    const double t = p.zhat[off + j] / ((dk[j] - dorg) - mu);
    s = __builtin_fma(t, t, s);
  }
  p.vnorm[off + i] = 1.0 / sqrt(s);
}

// ascending order of the merged eigenvalues (roots first, deflated after, in lamn): final column of each, lam_out
__global__ __launch_bounds__(256) void dc_rank2(DcLevel p) {
  int off, n1, n2;
  dc_shape(p, blockIdx.x, off, n1, n2);
  const int nn = n1 + n2;
  const int i = blockIdx.y * 256 + threadIdx.x;
  if (i >= nn) return;
  const double li = p.lamn[off + i];
  int r = 0;
  for (int j = 0; j < nn; ++j) {
    const double lj = p.lamn[off + j];
    r += (lj < li || (lj == li && j < i)) ? 1 : 0;
  }
  p.rank2[off + i] = r;
  p.lam_out[off + r] = li;
}

// S[row of the old basis][new column]: for a root i and a non-deflated position j: zhat_j / (d_j - lam_i) / |v_i|; a deflated
// position keeps its (rotated) old vector.  One thread per (t, i): t = index in the nd list, i = index in lamn.
__global__ __launch_bounds__(256) void dc_build_s(DcLevel p) {
  int off, n1, n2;
  dc_shape(p, blockIdx.x, off, n1, n2);
  const int nn = n1 + n2;
  const int k = p.kcnt[blockIdx.x];
  const int i = blockIdx.y * 256 + threadIdx.x;   // column (fast index: coalesced writes after the rank permutation, mostly)
  const int t = blockIdx.z;
  if (i >= nn) return;
  for (int tt = t; tt < nn; tt += gridDim.z) {
    const int row = off + p.perm[off + p.nd[off + tt]];
    const int col = off + p.rank2[off + i];
    double v = 0.0;
    if (tt < k) {
      if (i < k) {
        const double* __restrict__ dk = p.dk + off;
        v = p.zhat[off + tt] / ((dk[tt] - dk[p.org[off + i]]) - p.mu[off + i]) * p.vnorm[off + i];
      }
    } else if (i == tt) {
      v = 1.0;
    }
    p.S[(size_t)row * p.lds + col] = v;
  }
}

// leaves: dense 64 x 64 images of the tridiagonal blocks with the rank-one corrections taken off both ends
__global__ __launch_bounds__(256) void dc_leaf_fill(const double* __restrict__ d, const double* __restrict__ e, int n, double* __restrict__ M) {
  const int b = blockIdx.x, base = b * 64;
  double* Mb = M + (size_t)b * 4096;
  for (int idx = threadIdx.x; idx < 4096; idx += 256) {
    const int i = idx >> 6, j = idx & 63;
    double v = 0.0;
    if (i == j) {
      v = d[base + i];
      if (i == 0 && base > 0) v -= fabs(e[base - 1]);
      if (i == 63 && base + 64 < n) v -= fabs(e[base + 63]);
    } else if (j == i + 1) {
      v = e[base + i];
    } else if (i == j + 1) {
      v = e[base + j];
    }
    Mb[idx] = v;
  }
}

size_t ffgp_stedc_ws_doubles(int n) {
  // Z ping-pong partner + S: 2 n^2;  leaf images n * 64;  ~20 vectors of n (ints counted as doubles)
  return (size_t)2 * n * n + (size_t)n * 64 + (size_t)24 * n + 1024;
}

// d, e [n] (e[n-1] ignored) -> lam [n] ascending, Z [n, ldz] eigenvectors in columns.  n a multiple of 64.
int ffgp_stedc_impl(ffgp_handle* h, const double* d, const double* e, int n, double* lam, double* Z, int ldz, double* ws) {
  if (n < 64 || n % 64) return FFGP_ERR_ARG;
  hipStream_t st = h->stream;
  double* Zb = ws;                          // [n][n]
  double* S = Zb + (size_t)n * n;           // [n][n]
  double* M = S + (size_t)n * n;            // leaf images
  double* v = M + (size_t)n * 64;
  DcLevel p;
  p.n = n; p.e = e;
  double* lamA = v; v += n;
  double* lamB = v; v += n;
  p.z = v; v += n;
  p.ds = v; v += n;
  p.zs = v; v += n;
  p.rotc = v; v += n;
  p.rots = v; v += n;
  p.rho = v; v += n / 64 + 8;
  p.tol = v; v += n / 64 + 8;
  p.dk = v; v += n;
  p.zk = v; v += n;
  p.mu = v; v += n;
  p.zhat = v; v += n;
  p.vnorm = v; v += n;
  p.lamn = v; v += n;
  int* iv = reinterpret_cast<int*>(v);
  p.perm = iv; iv += n;
  p.defl = iv; iv += n;
  p.nd = iv; iv += n;
  p.kcnt = iv; iv += n / 64 + 8;
  p.nrot = iv; iv += n / 64 + 8;
  p.rotp = iv; iv += n;
  p.rotj = iv; iv += n;
  p.org = iv; iv += n;
  p.rank2 = iv; iv += n;
  p.S = S; p.lds = n;
  // leaves
  const int nl = n / 64;
  hipLaunchKernelGGL(dc_leaf_fill, dim3(nl), dim3(256), 0, st, d, e, n, M);
  // ping-pong so that the last level writes into the caller's Z
  int levels = 0;
  for (int bs = 128; bs / 2 < n; bs *= 2) ++levels;
  double* Zcur = (levels % 2 == 0) ? Z : Zb;
  int ldc = (levels % 2 == 0) ? ldz : n;
  double* Zoth = (levels % 2 == 0) ? Zb : Z;
  int ldo = (levels % 2 == 0) ? n : ldz;
  double* lcur = (levels % 2 == 0) ? lam : lamA;
  double* loth = (levels % 2 == 0) ? lamA : lam;
  (void)lamB;
  FFGP_HIP(hipMemsetAsync(Zb, 0, (size_t)n * n * sizeof(double), st));
  FFGP_HIP(hipMemset2DAsync(Z, (size_t)ldz * sizeof(double), 0, (size_t)n * sizeof(double), n, st));
  FFGP_CHECK(ffgp_syevj_small_impl(h, M, 64, 64, nl, 4096, Zcur, ldc, (long)64 * ldc + 64, lcur, 64, 0));
  for (int bs = 128; bs / 2 < n; bs *= 2) {
    const int nm = (n + bs - 1) / bs;
    p.bs = bs;
    p.Zc = Zcur; p.ldz = ldc;
    p.lam_in = lcur; p.lam_out = loth;
    const int chunks = (bs + 255) / 256;
    FFGP_HIP(hipMemsetAsync(p.defl, 0, (size_t)n * sizeof(int), st));
    hipLaunchKernelGGL(dc_setup, dim3(nm), dim3(256), 0, st, p);
    hipLaunchKernelGGL(dc_rank, dim3(nm, chunks), dim3(256), 0, st, p);
    hipLaunchKernelGGL(dc_deflate, dim3(nm), dim3(64), 0, st, p);
    hipLaunchKernelGGL(dc_secular, dim3(nm, (bs + 31) / 32), dim3(256), 0, st, p);
    hipLaunchKernelGGL(dc_zhat, dim3(nm, chunks), dim3(256), 0, st, p);
    hipLaunchKernelGGL(dc_norm, dim3(nm, chunks), dim3(256), 0, st, p);
    hipLaunchKernelGGL(dc_rank2, dim3(nm, chunks), dim3(256), 0, st, p);
    hipLaunchKernelGGL(dc_build_s, dim3(nm, chunks, min(bs, 256)), dim3(256), 0, st, p);
    hipLaunchKernelGGL(dc_rotate_s, dim3(nm, chunks), dim3(256), 0, st, p);
    if (hipGetLastError() != hipSuccess) return FFGP_ERR_HIP;
    // Q_new = blockdiag(Q1, Q2) S per merge: the top n1 rows from Q1 and S's first n1 rows, the bottom n2 from Q2 and the rest
    const int nfull = n / bs, hb = bs / 2;
    if (nfull > 0) {
      const long sa = (long)bs * ldc + bs, sb = (long)bs * n + bs, sc = (long)bs * ldo + bs;
      FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, Zcur, ldc, S, n, Zoth, ldo, hb, bs, hb, 1.0, 0.0, 0, ALIAS_NONE, nfull,
                                  sa, sb, sc));
      FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, Zcur + (size_t)hb * ldc + hb, ldc, S + (size_t)hb * n, n,
                                  Zoth + (size_t)hb * ldo, ldo, hb, bs, hb, 1.0, 0.0, 0, ALIAS_NONE, nfull, sa, sb, sc));
    }
    const int rem = n - nfull * bs;
    if (rem > 0) {
      const size_t o = (size_t)nfull * bs;
      const int r1 = min(hb, rem), r2 = rem - r1;
      FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, Zcur + o * ldc + o, ldc, S + o * n + o, n, Zoth + o * ldo + o, ldo,
                                  r1, rem, r1, 1.0, 0.0));
      if (r2 > 0)
        FFGP_CHECK(ffgp_gemm_launch(h, OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, Zcur + (o + r1) * ldc + o + r1, ldc, S + (o + r1) * n + o, n,
                                    Zoth + (o + r1) * ldo + o, ldo, r2, rem, r2, 1.0, 0.0));
    }
    double* tz = Zcur; Zcur = Zoth; Zoth = tz;
    int tl = ldc; ldc = ldo; ldo = tl;
    double* tlam = lcur; lcur = loth; loth = tlam;
  }
  return FFGP_OK;
}
