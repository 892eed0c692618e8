// Covariance assembly for gfx950:  K_ij = amp * exp(-1/2 * max(sum_k ((x_ik - x_jk) w_k)^2, clamp))  (+ Sigma extras).
//
// One pass, write-only over K: the reference's torch path makes >= 6 N x N passes (scaled copies, norms, N x N
// matmul, sqrt, square, exp, mul, plus 2-3 torch.eye temporaries for the diagonal adds -- GaussianProcess/kernel.py:
// 100-105, cigp_v10.py:57-60).  64x64 output tile per 256-thread workgroup, 4x4 outputs per thread; the two
// scaled input panels are staged through LDS in 16-dimension chunks (x1 as [row][17], x2 transposed [dim][64], both
// conflict-free), every output row is written as 128-byte segments.  The diagonal / full-matrix / all-entries
// adds of the four Sigma conventions (S1-S4) are fused into the epilogue; the mean(K) jitter of
// gp_computation_pack.negative_log_likelihood is a tile-sum atomics + one tiny follow-up kernel.
#include "ffgp_internal.h"

#define AT 64   // tile edge
#define DC 16   // dimension chunk

struct AsmArgs {
  const double* X1; int n1;
  const double* X2; int n2;
  int D;
  const double* w; const double* amp; double clamp;
  const double* diag_add; const double* diag_vec; long diag_stride;
  const double* add_mat; int ld_add; double add_all;
  double* K; int ldk; int lower_only; int symmetric;
  double* ksum;   // nullable: accumulates sum(K) over the full n1 x n2 matrix
  int tiles_n;
  int kfun; double rinv;   // radial profile and 1/rho
};

// one 64 x 64 tile in the difference form  sum_k ((x_ik - x_jk) w_k)^2  (every rounding error relative to the distance itself)
__device__ __forceinline__ void asm_tile_diff(const AsmArgs& a, int ti, int tj, int tid, double (*x1s)[DC + 1], double (*x2t)[AT + 1],
                                              const ExpCoef& ec, double& tsum_out) {
  const int tx = tid & 15, ty = tid >> 4;
  const int r0 = ti * AT, c0 = tj * AT;

  double sq[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) sq[i][j] = 0.0;

  for (int d0 = 0; d0 < a.D; d0 += DC) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 256 * q;
      const int row = idx >> 4, dd = idx & 15;
      const int gd = d0 + dd;
      // (unconditional loads from clamped addresses, masked afterwards: guarded loads serialise on memory latency)
      const int gdc = min(gd, a.D - 1);
      const double wk = a.w[gdc];
      const int g1 = r0 + row, g2 = c0 + row;
      const double l1 = a.X1[(size_t)min(g1, a.n1 - 1) * a.D + gdc], l2 = a.X2[(size_t)min(g2, a.n2 - 1) * a.D + gdc];
      x1s[row][dd] = (gd < a.D && g1 < a.n1) ? l1 * wk : 0.0;
      x2t[dd][row] = (gd < a.D && g2 < a.n2) ? l2 * wk : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int dd = 0; dd < DC; ++dd) {
      double p[4], q2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) p[i] = x1s[ty + 16 * i][dd];
#pragma unroll
      for (int j = 0; j < 4; ++j) q2[j] = x2t[dd][tx + 16 * j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const double df = p[i] - q2[j];
          sq[i][j] = __builtin_fma(df, df, sq[i][j]);
        }
    }
    __syncthreads();
  }

  const double amp = a.amp[0];
  const double dadd = a.diag_add ? a.diag_add[0] : 0.0;
  const bool se = (a.kfun == FFGP_KFUN_SE);
  double tsum = 0.0;
  // interior tiles of the squared-exponential profile (all 64 x 64 entries exist, none on the diagonal, no matrix add): no
  // per-entry bounds / diagonal / triangle selects -- the kernel is bound by vector-instruction issue, every one counts
  if (se && r0 + AT <= a.n1 && c0 + AT <= a.n2 && !(a.symmetric && ti == tj) && !a.add_mat) {
    const double addall = a.symmetric ? a.add_all : 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double* __restrict__ dst = a.K + (size_t)(r0 + ty + 16 * i) * a.ldk + c0 + tx;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double k = amp * ffgp_exp_fast(-0.5 * fmax(sq[i][j], a.clamp), ec);
        tsum += k;
        dst[16 * j] = k + addall;
      }
    }
  } else
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = r0 + ty + 16 * i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = c0 + tx + 16 * j;
      if (row < a.n1 && col < a.n2) {
        const double s = fmax(sq[i][j], a.clamp);
        double k = amp * (se ? ffgp_exp_fast(-0.5 * s, ec) : ffgp_kfun_val(a.kfun, a.rinv, s));
        tsum += k;
        if (a.symmetric) {
          if (row == col) {
            k += dadd;
            if (a.diag_vec) k += a.diag_vec[(size_t)row * a.diag_stride];
          }
          if (a.add_mat) k += (row >= col) ? a.add_mat[(size_t)row * a.ld_add + col] : a.add_mat[(size_t)col * a.ld_add + row];
          k += a.add_all;
        }
        if (!a.lower_only || col <= row) a.K[(size_t)row * a.ldk + col] = k;
      }
    }
  }
  // full-matrix sum: an off-diagonal tile of the lower-only sweep stands for its mirror image too
  tsum_out += (a.lower_only && ti != tj) ? 2.0 * tsum : tsum;
}

// block sum of the tile sums -> one atomic (the mean(K) jitter of gp_computation_pack.negative_log_likelihood)
__device__ __forceinline__ void asm_ksum(const AsmArgs& a, double tsum, double* red, int tid) {
  for (int o = 32; o > 0; o >>= 1) tsum += __shfl_down(tsum, o);
  if ((tid & 63) == 0) red[tid >> 6] = tsum;
  __syncthreads();
  if (tid == 0) atomicAdd(a.ksum, red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void ffgp_assemble_kernel(AsmArgs a) {
  __shared__ double x1s[AT][DC + 1];
  __shared__ double x2t[DC][AT + 1];
  __shared__ double red[4];
  const int tid = threadIdx.x;
  int ti, tj;
  if (a.lower_only) {
    const int t = blockIdx.x;
    int r = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= t) ++r;
    while (r * (r + 1) / 2 > t) --r;
    ti = r;
    tj = t - r * (r + 1) / 2;
  } else {
    ti = blockIdx.x / a.tiles_n;
    tj = blockIdx.x % a.tiles_n;
  }
  ExpCoef ec;
  ffgp_exp_load(ec);
  double tsum = 0.0;
  asm_tile_diff(a, ti, tj, tid, x1s, x2t, ec, tsum);
  if (a.ksum) asm_ksum(a, tsum, red, tid);
}

// up to FFGP_MULTI_MAX independent assemblies per launch (the members of a shared-chain batch of small blocks: one launch of a few
// tiles each otherwise); gridDim.y = member, a member's surplus workgroups leave at once
struct AsmMulti {
  AsmArgs a[FFGP_MULTI_MAX];
  int tiles[FFGP_MULTI_MAX];
};
__global__ __launch_bounds__(256) void ffgp_assemble_kernel_multi(AsmMulti q) {
  __shared__ double x1s[AT][DC + 1];
  __shared__ double x2t[DC][AT + 1];
  const int z = blockIdx.y;
  if ((int)blockIdx.x >= q.tiles[z]) return;
  const AsmArgs& a = q.a[z];
  const int tid = threadIdx.x;
  int ti, tj;
  if (a.lower_only) {
    const int t = blockIdx.x;
    int r = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= t) ++r;
    while (r * (r + 1) / 2 > t) --r;
    ti = r;
    tj = t - r * (r + 1) / 2;
  } else {
    ti = blockIdx.x / a.tiles_n;
    tj = blockIdx.x % a.tiles_n;
  }
  ExpCoef ec;
  ffgp_exp_load(ec);
  double tsum = 0.0;
  asm_tile_diff(a, ti, tj, tid, x1s, x2t, ec, tsum);
}

struct AsmCollector {
  std::vector<AsmArgs> a;
  std::vector<int> tiles;
};

// launch what ffgp_assemble_impl parked in the handle's collector (ffgp_handle::asm_collect), eight members per launch
int ffgp_assemble_flush(ffgp_handle* h) {
  AsmCollector* c = static_cast<AsmCollector*>(h->asm_collect);
  if (!c) return FFGP_OK;
  const int F = (int)c->a.size();
  for (int f0 = 0; f0 < F; f0 += FFGP_MULTI_MAX) {
    const int cnt = F - f0 < FFGP_MULTI_MAX ? F - f0 : FFGP_MULTI_MAX;
    AsmMulti q;
    int gx = 1;
    for (int z = 0; z < FFGP_MULTI_MAX; ++z) {
      const int f = f0 + (z < cnt ? z : 0);
      q.a[z] = c->a[f];
      q.tiles[z] = c->tiles[f];
      if (z < cnt) gx = max(gx, c->tiles[f]);
    }
    hipLaunchKernelGGL(ffgp_assemble_kernel_multi, dim3(gx, cnt), dim3(256), 0, h->stream, q);
  }
  c->a.clear();
  c->tiles.clear();
  if (hipGetLastError() != hipSuccess) return FFGP_ERR_HIP;
  return FFGP_OK;
}
int ffgp_assemble_collect_begin(ffgp_handle* h) {
  if (!h->asm_collect) h->asm_collect = new AsmCollector();
  static_cast<AsmCollector*>(h->asm_collect)->a.clear();
  static_cast<AsmCollector*>(h->asm_collect)->tiles.clear();
  h->asm_collecting = 1;
  return FFGP_OK;
}
void ffgp_assemble_collect_free(ffgp_handle* h) {
  delete static_cast<AsmCollector*>(h->asm_collect);
  h->asm_collect = nullptr;
}
int ffgp_assemble_collect_end(ffgp_handle* h) {
  h->asm_collecting = 0;
  return ffgp_assemble_flush(h);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Interior tiles of the squared-exponential profile on the matrix cores (round 3).  The vector pipe is what bounds the
// difference form (32 fp64 instructions per entry at D = 16 before the exp even starts), so for 64 x 64 tiles that lie wholly
// inside the matrix and off the diagonal the squared distance comes out of ONE MFMA chain over k = D + 2:
//     [ x_i o w , |x_i o w|^2 , 1 ] . [ -2 x_j o w , 1 , |x_j o w|^2 ]  =  |x_i o w|^2 + |x_j o w|^2 - 2 (x_i o w).(x_j o w)
// with the inputs shifted by the first point and scaled once by a small pre-kernel (xs, nr: N (D + 1) doubles, L2-resident), the
// operands loaded from there straight into the MFMA lane layout (no LDS, no barrier: every wave is on its own), and the
// epilogue -- clamp, exp, amplitude, 16-byte stores: the two 16-column blocks of a wave hold interleaved columns, so a lane owns
// two adjacent entries of a row -- running on the vector pipe of one wave while other waves' chains occupy the matrix pipe.
// Workgroups are persistent (a wave walks 32 x 32 quarters of tiles t, t + grid, ...) and load the next tile's operands before
// the epilogue of the current one.
// The expansion's error is eps (|x_i|^2 + |x_j|^2) in the distance, the difference form's eps |x_i - x_j|^2: wherever an entry
// of a wave's block has  distance < ASM_TAU (|x_i|^2 + |x_j|^2)  (near-coincident points: the entries that decide how
// well-conditioned Sigma is) the wave writes nothing and flags the tile; a second, small launch recomputes flagged tiles in the
// difference form together with the diagonal and edge tiles.  Every other profile runs the difference kernel alone.
// torch.cdist -- the reference, kernel.py:100-105 -- uses the expansion for every entry.
// ------------------------------------------------------------------------------------------------------------------------------
#define ASM_TAU 1e-6
typedef double asm_v4d __attribute__((ext_vector_type(4)));
typedef double asm_v2d __attribute__((ext_vector_type(2)));

// Operand arrays in the MFMA lane layout (16 k-columns per chunk, zero-padded; lane = 16 g + c owns row c of its block and the
// k-columns 4 g .. 4 g + 3 of the chunk, split in two 16-byte halves h so that a wave's load of one half is 1 KB contiguous):
//   xa[((blkA * nchunk + kc) * 2 + h) * 64 + lane][2]         blkA = row / 16,                   c = row % 16        value  v
//   xb[(((blkB * 2 + q) * nchunk + kc) * 2 + h) * 64 + lane][2]   blkB = row / 32, q = row % 2,   c = (row % 32) / 2   value -2 v
// (the B side holds the interleaved columns 2 c + q of a 32-column block, so a lane's two accumulators are adjacent entries of a
// row), v = (x - x0) o w; nr = |v|^2.  Rows n .. npad - 1 are zero.
__global__ __launch_bounds__(256) void ffgp_asm_prep_kernel(const double* __restrict__ X, int n, int npad, int D, int nchunk,
                                                            const double* __restrict__ w, const double* __restrict__ X0,
                                                            double* __restrict__ xa, double* __restrict__ xb, double* __restrict__ nr,
                                                            int* __restrict__ flags, int nflags) {
  // 16 lanes per row: lane kl owns the k-columns kl, kl + 16, ...; 16 rows per 256 threads
  const int gt = blockIdx.x * 256 + threadIdx.x;
  for (int f = gt; f < nflags; f += gridDim.x * 256) flags[f] = 0;
  const int i = gt >> 4, kl = gt & 15;
  if (i >= npad) return;
  double s = 0.0;
  const int ca = i & 15, blkA = i >> 4, cb = (i & 31) >> 1, blkB = i >> 5, q = i & 1;
  const int g = kl >> 2, hh = (kl >> 1) & 1, e = kl & 1;
  for (int kc = 0; kc < nchunk; ++kc) {
    const int k = 16 * kc + kl;
    const double v = (i < n && k < D) ? (X[(size_t)i * D + k] - X0[k]) * w[k] : 0.0;
    if (xa) xa[((((size_t)blkA * nchunk + kc) * 2 + hh) * 64 + 16 * g + ca) * 2 + e] = v;
    if (xb) xb[(((((size_t)blkB * 2 + q) * nchunk + kc) * 2 + hh) * 64 + 16 * g + cb) * 2 + e] = -2.0 * v;
    s = __builtin_fma(v, v, s);
  }
  for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
  if (kl == 0) nr[i] = s;
}

struct AsmMM {
  const double* xa; const double* nr1;     // rows of X1 (A layout)
  const double* xb; const double* nr2;     // rows of X2 (B layout, times -2)
  int nchunk;
  int* flags;      // [tiles_m][tiles_n]: 1 = recompute this 64 x 64 tile in the difference form
  int tiles_m, ntiles;
};

__device__ __forceinline__ void asm_tile_decode(const AsmArgs& a, int t, int& ti, int& tj) {
  if (a.lower_only) {
    int r = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((r + 1) * (r + 2) / 2 <= t) ++r;
    while (r * (r + 1) / 2 > t) --r;
    ti = r;
    tj = t - r * (r + 1) / 2;
  } else {
    ti = t / a.tiles_n;
    tj = t % a.tiles_n;
  }
}

__device__ __forceinline__ bool asm_tile_interior(const AsmArgs& a, int ti, int tj) {
  return (ti + 1) * AT <= a.n1 && (tj + 1) * AT <= a.n2 && !(a.symmetric && ti == tj);
}

// operands of one 32 x 32 block, first 16 coordinates: lane group g owns coordinates 4 g .. 4 g + 3 of its row -- four consecutive
// doubles, so a chunk touches every 128-byte line of the operand rows once (the MFMA sums over k in any order, as long as both
// operands use the same one); step t of the chunk multiplies coordinate kb + 4 g + t
struct AsmOps {
  asm_v2d av[2][2], bv[2][2];   // [row block | q][half]
  double an[2], bn[2];
};

__device__ __forceinline__ void asm_load_ops(const AsmMM& m, int R0, int C0, int lane, int kc, bool norms, AsmOps& o) {
  const asm_v2d* __restrict__ xa = reinterpret_cast<const asm_v2d*>(m.xa);
  const asm_v2d* __restrict__ xb = reinterpret_cast<const asm_v2d*>(m.xb);
  const int c = lane & 15;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const unsigned ba = (unsigned)(((R0 >> 4) + rb) * m.nchunk + kc) * 128u + (unsigned)lane;
    const unsigned bb = (unsigned)(((C0 >> 5) * 2 + rb) * m.nchunk + kc) * 128u + (unsigned)lane;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      o.av[rb][hh] = xa[ba + 64u * hh];
      o.bv[rb][hh] = xb[bb + 64u * hh];
    }
    if (norms) {
      o.an[rb] = m.nr1[R0 + 16 * rb + c];
      o.bn[rb] = m.nr2[C0 + 2 * c + rb];
    }
  }
}

// next interior tile at or after (ti, tj) in row-major order of the sweep; false when the range [.., t_end) is exhausted
__device__ __forceinline__ bool asm_next_interior(const AsmArgs& a, int& t, int t_end, int& ti, int& tj) {
  while (t < t_end) {
    if (asm_tile_interior(a, ti, tj)) return true;
    ++t;
    ++tj;
    if (tj >= (a.lower_only ? ti + 1 : a.tiles_n)) { tj = 0; ++ti; }
  }
  return false;
}

// 16 rows x 64 columns of a diagonal tile in the difference form (the diagonal tiles of the symmetric sweep are cut in four
// slices and dealt over the persistent workgroups of the matrix-core kernel, one slice each at N = 16384)
__device__ __forceinline__ void asm_diag_slice(const AsmArgs& a, int td, int sl, int tid, double (*x1s)[DC + 1], double (*x2t)[AT + 1],
                                               const ExpCoef& ec, double& tsum) {
  const int r0 = td * AT + 16 * sl, c0 = td * AT;
  const int ty = tid >> 4, tx = tid & 15;
  double sq[4] = {0.0, 0.0, 0.0, 0.0};
  for (int d0 = 0; d0 < a.D; d0 += DC) {
    {
      const int dd = tid & 15, gd = d0 + dd, gdc = min(gd, a.D - 1);
      const double wk = a.w[gdc];
      const double l1 = a.X1[(size_t)min(r0 + ty, a.n1 - 1) * a.D + gdc];
      x1s[ty][dd] = (gd < a.D && r0 + ty < a.n1) ? l1 * wk : 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = ty + 16 * q;
        const double l2 = a.X2[(size_t)min(c0 + row, a.n2 - 1) * a.D + gdc];
        x2t[dd][row] = (gd < a.D && c0 + row < a.n2) ? l2 * wk : 0.0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int dd = 0; dd < DC; ++dd) {
      const double p = x1s[ty][dd];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double df = p - x2t[dd][tx + 16 * j];
        sq[j] = __builtin_fma(df, df, sq[j]);
      }
    }
    __syncthreads();
  }
  const double amp = a.amp[0];
  const double dadd = a.diag_add ? a.diag_add[0] : 0.0;
  const int row = r0 + ty;
  double ts = 0.0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int col = c0 + tx + 16 * j;
    if (row < a.n1 && col < a.n2) {
      double k = amp * ffgp_exp_fast(-0.5 * fmax(sq[j], a.clamp), ec);
      ts += k;
      if (row == col) {
        k += dadd;
        if (a.diag_vec) k += a.diag_vec[(size_t)row * a.diag_stride];
      }
      k += a.add_all;
      if (!a.lower_only || col <= row) a.K[(size_t)row * a.ldk + col] = k;
    }
  }
  tsum += ts;
}

struct AsmPend {       // a finished block whose stores are issued under the next block's products
  double* dst;
  bool valid;
};

__device__ __forceinline__ void asm_store_block(const AsmArgs& a, const AsmPend& pd, const asm_v4d (&kv)[2][2]) {
  if (!pd.valid) return;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<asm_v2d*>(pd.dst + (size_t)(16 * rb + 4 * r) * a.ldk) = asm_v2d{kv[rb][0][r], kv[rb][1][r]};
}

struct AsmWalk {
  int t, t_end, ti, tj;
  bool have;
};

// one 32 x 32 block: products into `acc`, the previous block's stores from `prev`, the next block's operand loads, then the
// epilogue of this block in place (acc -> K values; stored by the next step)
template <bool KS>
__device__ __forceinline__ void asm_mm_step(const AsmArgs& a, const AsmMM& m, const ExpCoef& ec, double amp, double addall, int lane,
                                            int roff, int coff, AsmOps& o, asm_v4d (&acc)[2][2], const asm_v4d (&prev)[2][2],
                                            AsmPend& pd, AsmWalk& wk, double& tsum) {
  const int g = lane >> 4, c = lane & 15;
  const int R0 = wk.ti * AT + roff, C0 = wk.tj * AT + coff;
  const int tflag = wk.ti * a.tiles_n + wk.tj;
  // the smallest distance this wave accepts from the expansion (a bound over its whole block)
  double na = fmax(o.an[0], o.an[1]), nb = fmax(o.bn[0], o.bn[1]);
  for (int sh = 1; sh < 16; sh <<= 1) {
    na = fmax(na, __shfl_xor(na, sh));
    nb = fmax(nb, __shfl_xor(nb, sh));
  }
  const double thr = ASM_TAU * (na + nb);
  {   // the two norm columns:  |a|^2 . 1  +  1 . |b|^2
    double ca[2], cb[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      ca[rb] = (g == 0) ? o.an[rb] : (g == 1 ? 1.0 : 0.0);
      cb[rb] = (g == 0) ? 1.0 : (g == 1 ? o.bn[rb] : 0.0);
    }
    const asm_v4d zero = asm_v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int q = 0; q < 2; ++q) acc[rb][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[rb], cb[q], zero, 0, 0, 0);
  }
  for (int kc = 0; kc < m.nchunk; ++kc) {
    if (kc) asm_load_ops(m, R0, C0, lane, kc, false, o);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          acc[rb][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.av[rb][s >> 1][s & 1], o.bv[q][s >> 1][s & 1], acc[rb][q], 0, 0, 0);
  }
  asm_store_block(a, pd, prev);
  // next interior tile: its operands travel under this block's epilogue
  ++wk.t;
  ++wk.tj;
  if (wk.tj >= (a.lower_only ? wk.ti + 1 : a.tiles_n)) { wk.tj = 0; ++wk.ti; }
  wk.have = asm_next_interior(a, wk.t, wk.t_end, wk.ti, wk.tj);
  if (wk.have) asm_load_ops(m, wk.ti * AT + roff, wk.tj * AT + coff, lane, 0, true, o);
  // epilogue on registers: lane holds rows g + 4 r of row block rb, columns C0 + 2 c + {0, 1}
  double smin = 1e300;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) smin = fmin(smin, acc[rb][q][r]);
  if (__any(smin < thr)) {
    if (lane == 0) {
      m.flags[tflag] = 1;       // (benign race between the four waves of a tile: every writer stores 1)
      m.flags[-1] = 1;          // "something to redo" for the fix-up launch
    }
    pd.valid = false;
  } else {
    double ts = 0.0;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double e = ffgp_exp_fast(-0.5 * fmax(acc[rb][q][r], a.clamp), ec);
          if (KS) {
            const double k0 = amp * e;
            ts += k0;
            acc[rb][q][r] = k0 + addall;
          } else {
            acc[rb][q][r] = __builtin_fma(amp, e, addall);
          }
        }
    if (KS) tsum += a.lower_only ? 2.0 * ts : ts;   // (an interior tile of the lower sweep stands for its mirror image too)
    pd.valid = true;
    pd.dst = a.K + (size_t)(R0 + g) * a.ldk + C0 + 2 * c;
  }
}

template <bool KS>
__global__ __launch_bounds__(256, 3) void ffgp_assemble_mm_kernel(AsmArgs a, AsmMM m) {
  __shared__ double x1s[16][DC + 1];
  __shared__ double x2t[DC][AT + 1];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  ExpCoef ec;
  ffgp_exp_load(ec);
  const double amp = a.amp[0];
  const double addall = a.symmetric ? a.add_all : 0.0;
  double tsum = 0.0;
  // a workgroup owns a contiguous range of the tile sweep: one decode, then increments
  const int per = (m.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  AsmWalk wk;
  wk.t = __builtin_amdgcn_readfirstlane((int)blockIdx.x * per);
  wk.t_end = min(wk.t + per, m.ntiles);
  wk.ti = 0;
  wk.tj = 0;
  if (wk.t < wk.t_end) asm_tile_decode(a, wk.t, wk.ti, wk.tj);
  wk.ti = __builtin_amdgcn_readfirstlane(wk.ti);
  wk.tj = __builtin_amdgcn_readfirstlane(wk.tj);
  wk.have = asm_next_interior(a, wk.t, wk.t_end, wk.ti, wk.tj);
  const int roff = 32 * (wv >> 1), coff = 32 * (wv & 1);
  AsmOps o;
  if (wk.have) asm_load_ops(m, wk.ti * AT + roff, wk.tj * AT + coff, lane, 0, true, o);
  // this workgroup's share of the diagonal tiles (difference form), under the first operand loads
  if (a.symmetric)
    for (int sl = blockIdx.x; sl < 4 * m.tiles_m; sl += gridDim.x) asm_diag_slice(a, sl >> 2, sl & 3, tid, x1s, x2t, ec, tsum);
  asm_v4d accA[2][2], accB[2][2];
  AsmPend pd;
  pd.valid = false;
  pd.dst = a.K;
  bool lastA = true;
  while (wk.have) {
    asm_mm_step<KS>(a, m, ec, amp, addall, lane, roff, coff, o, accA, accB, pd, wk, tsum);
    lastA = true;
    if (!wk.have) break;
    asm_mm_step<KS>(a, m, ec, amp, addall, lane, roff, coff, o, accB, accA, pd, wk, tsum);
    lastA = false;
  }
  if (lastA) asm_store_block(a, pd, accA); else asm_store_block(a, pd, accB);
  if (KS) {
    for (int sh = 32; sh > 0; sh >>= 1) tsum += __shfl_down(tsum, sh);
    if (lane == 0) atomicAdd(a.ksum, tsum);
  }
}

// edge tiles, and the tiles the matrix-core pass flagged, in the difference form: workgroup (ti, s) walks the column tiles
// s, s + 4, ... of its row.  (The diagonal tiles of a symmetric sweep were done by the matrix-core launch.)
#define ASM_FIX_S 4
__global__ __launch_bounds__(256) void ffgp_assemble_fix_kernel(AsmArgs a, const int* __restrict__ flags, int edges) {
  __shared__ double x1s[AT][DC + 1];
  __shared__ double x2t[DC][AT + 1];
  __shared__ double red[4];
  const int tid = threadIdx.x;
  const int ti = blockIdx.x / ASM_FIX_S, s0 = blockIdx.x % ASM_FIX_S;
  if (!edges && !flags[-1]) return;     // nothing flagged, no ragged edge: the usual case
  ExpCoef ec;
  ffgp_exp_load(ec);
  double tsum = 0.0;
  const int jend = a.lower_only ? ti + 1 : a.tiles_n;
  for (int tj = s0; tj < jend; tj += ASM_FIX_S)
    if ((!asm_tile_interior(a, ti, tj) && !(a.symmetric && ti == tj)) || flags[ti * a.tiles_n + tj])
      asm_tile_diff(a, ti, tj, tid, x1s, x2t, ec, tsum);
  if (a.ksum) asm_ksum(a, tsum, red, tid);
}

__global__ void ffgp_zero_scalar(double* p) { p[0] = 0.0; }

__global__ void ffgp_mean_jitter_kernel(double* K, int ldk, int n, const double* ksum, double coef) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) K[(size_t)i * ldk + i] += coef * ksum[0] / ((double)n * (double)n);
}

// dst[c, r] = src[r, c]  (src rows x cols, row-major) -- used to park Y^T / K_*^T as passenger rows and to
// transpose results back
__global__ __launch_bounds__(256) void ffgp_transpose_kernel(const double* __restrict__ src, int rows, int cols, int lds_,
                                                             double* __restrict__ dst, int ldd, double scale) {
  __shared__ double t[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;  // bx over cols, by over rows
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int r = by + i, c = bx + tx;
    t[i][tx] = (r < rows && c < cols) ? src[(size_t)r * lds_ + c] : 0.0;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = bx + i, r = by + tx;
    if (c < cols && r < rows) dst[(size_t)c * ldd + r] = scale * t[tx][i];
  }
}

int ffgp_transpose(ffgp_handle* h, const double* src, int rows, int cols, int ld_src, double* dst, int ld_dst, double scale) {
  if (rows <= 0 || cols <= 0) return FFGP_OK;
  dim3 grid((cols + 31) / 32, (rows + 31) / 32);
  hipLaunchKernelGGL(ffgp_transpose_kernel, grid, dim3(256), 0, h->stream, src, rows, cols, ld_src, dst, ld_dst, scale);
  return FFGP_OK;
}

// up to FFGP_MULTI_MAX independent transposes per launch (the targets of a shared-chain batch's members)
__global__ __launch_bounds__(256) void ffgp_transpose_multi_kernel(ffgp_multi_tr q) {
  const int z = blockIdx.z;
  const int rows = q.rows[z], cols = q.cols[z];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  if (bx >= cols || by >= rows) return;
  const double* __restrict__ src = q.src[z];
  double* __restrict__ dst = q.dst[z];
  const int lds_ = q.lds[z], ldd = q.ldd[z];
  __shared__ double t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = by + i, c = bx + tx;
    t[i][tx] = (r < rows && c < cols) ? src[(size_t)r * lds_ + c] : 0.0;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = bx + i, r = by + tx;
    if (c < cols && r < rows) dst[(size_t)c * ldd + r] = t[tx][i];
  }
}

int ffgp_transpose_multi(ffgp_handle* h, int F, const double* const* src, const int* rows, const int* cols, const int* ld_src,
                         double* const* dst, const int* ld_dst) {
  for (int f0 = 0; f0 < F; f0 += FFGP_MULTI_MAX) {
    const int cnt = F - f0 < FFGP_MULTI_MAX ? F - f0 : FFGP_MULTI_MAX;
    ffgp_multi_tr q;
    int gx = 1, gy = 1;
    for (int z = 0; z < FFGP_MULTI_MAX; ++z) {
      const int f = f0 + (z < cnt ? z : 0);
      q.src[z] = src[f]; q.dst[z] = dst[f]; q.rows[z] = rows[f]; q.cols[z] = cols[f]; q.lds[z] = ld_src[f]; q.ldd[z] = ld_dst[f];
      if (z < cnt) {
        gx = max(gx, (cols[f] + 31) / 32);
        gy = max(gy, (rows[f] + 31) / 32);
      }
    }
    hipLaunchKernelGGL(ffgp_transpose_multi_kernel, dim3(gx, gy, cnt), dim3(256), 0, h->stream, q);
  }
  return FFGP_OK;
}

int ffgp_assemble_impl(ffgp_handle* h, const double* X1, int n1, const double* X2, int n2, int D, const double* w,
                       const double* amp, double clamp_min, const double* diag_add, const double* diag_vec,
                       long diag_stride, const double* add_mat, int ld_add, double add_all, double mean_jitter, double* K,
                       int ldk, int lower_only, int kfun, double kparam) {
  if (n1 <= 0 || n2 <= 0) return FFGP_OK;
  if (!X1 || !X2 || !w || !amp || !K || D <= 0 || ldk < n2) return FFGP_ERR_ARG;
  const bool symmetric = (X1 == X2 && n1 == n2);
  const bool extras = diag_add || diag_vec || add_mat || add_all != 0.0 || mean_jitter != 0.0;
  if (extras && !symmetric) return FFGP_ERR_ARG;
  if (lower_only && !symmetric) return FFGP_ERR_ARG;
  AsmArgs a;
  a.X1 = X1; a.n1 = n1; a.X2 = X2; a.n2 = n2; a.D = D;
  a.w = w; a.amp = amp; a.clamp = clamp_min;
  a.diag_add = diag_add; a.diag_vec = diag_vec; a.diag_stride = diag_stride;
  a.add_mat = add_mat; a.ld_add = ld_add; a.add_all = add_all;
  a.K = K; a.ldk = ldk; a.lower_only = lower_only ? 1 : 0; a.symmetric = symmetric ? 1 : 0;
  a.ksum = nullptr;
  a.kfun = kfun;
  a.rinv = (kparam != 0.0) ? 1.0 / kparam : 1.0;
  if (mean_jitter != 0.0) {
    a.ksum = h->d_scal + 32;
    hipLaunchKernelGGL(ffgp_zero_scalar, dim3(1), dim3(1), 0, h->stream, a.ksum);
  }
  const int tm = (n1 + AT - 1) / AT;
  a.tiles_n = (n2 + AT - 1) / AT;
  const int tiles = lower_only ? tm * (tm + 1) / 2 : tm * a.tiles_n;
  // squared-exponential profile, big enough, 16-byte stores possible: interior tiles on the matrix cores
  const bool mm = kfun == FFGP_KFUN_SE && !add_mat && h->asm_mm && (long)n1 * n2 >= (long)h->asm_mm_min * h->asm_mm_min && n1 >= 256 &&
                  n2 >= 128 && D <= 128 && (ldk & 1) == 0 && ((uintptr_t)K & 15) == 0;
  if (mm) {
    const int nchunk = (D + 15) / 16, np1 = tm * AT, np2 = a.tiles_n * AT;
    const size_t dA = (size_t)np1 * nchunk * 16, dB = (size_t)np2 * nchunk * 16;
    const size_t need = (dA + dB + (size_t)np1 + (size_t)np2) * sizeof(double) + (size_t)tm * (size_t)a.tiles_n * sizeof(int) + 128;
    if (h->asm_bytes < need) {
      if (h->d_asm) FFGP_HIP(hipFree(h->d_asm));
      h->d_asm = nullptr;
      h->asm_bytes = 0;
      FFGP_HIP(hipMalloc(&h->d_asm, need));
      h->asm_bytes = need;
      ++h->alloc_epoch;
    }
    AsmMM m;
    const int nflags = tm * a.tiles_n;
    double* xa = h->d_asm;
    double* xb = xa + dA;
    double* nr1 = xb + dB;
    double* nr2 = nr1 + np1;
    m.flags = (int*)(nr2 + np2) + 2;      // flags[-1]: "any tile flagged"
    m.xa = xa; m.xb = xb; m.nr1 = nr1; m.nr2 = symmetric ? nr1 : nr2; m.nchunk = nchunk;
    if (symmetric) {
      hipLaunchKernelGGL(ffgp_asm_prep_kernel, dim3((np1 + 15) / 16), dim3(256), 0, h->stream, X1, n1, np1, D, nchunk, w, X1, xa, xb, nr1,
                         m.flags - 1, nflags + 1);
    } else {
      hipLaunchKernelGGL(ffgp_asm_prep_kernel, dim3((np1 + 15) / 16), dim3(256), 0, h->stream, X1, n1, np1, D, nchunk, w, X1, xa,
                         (double*)nullptr, nr1, m.flags - 1, nflags + 1);
      hipLaunchKernelGGL(ffgp_asm_prep_kernel, dim3((np2 + 15) / 16), dim3(256), 0, h->stream, X2, n2, np2, D, nchunk, w, X1,
                         (double*)nullptr, xb, nr2, (int*)nullptr, 0);
    }
    m.tiles_m = tm;
    m.ntiles = tiles;
    const int grid = tiles < h->asm_mm_grid ? tiles : h->asm_mm_grid;
    if (a.ksum) hipLaunchKernelGGL(ffgp_assemble_mm_kernel<true>, dim3(grid), dim3(256), 0, h->stream, a, m);
    else hipLaunchKernelGGL(ffgp_assemble_mm_kernel<false>, dim3(grid), dim3(256), 0, h->stream, a, m);
    hipLaunchKernelGGL(ffgp_assemble_fix_kernel, dim3(tm * ASM_FIX_S), dim3(256), 0, h->stream, a, m.flags,
                       (n1 % AT != 0 || n2 % AT != 0) ? 1 : 0);
  } else if (h->asm_collecting && !a.ksum && tiles <= 1024) {
    // a member of a batch of small blocks: parked, launched together with the other members' (ffgp_assemble_flush)
    AsmCollector* c = static_cast<AsmCollector*>(h->asm_collect);
    c->a.push_back(a);
    c->tiles.push_back(tiles);
  } else {
    hipLaunchKernelGGL(ffgp_assemble_kernel, dim3(tiles), dim3(256), 0, h->stream, a);
  }
  if (mean_jitter != 0.0)
    hipLaunchKernelGGL(ffgp_mean_jitter_kernel, dim3((n1 + 255) / 256), dim3(256), 0, h->stream, K, ldk, n1, a.ksum,
                       mean_jitter);
  if (hipGetLastError() != hipSuccess) return FFGP_ERR_HIP;
  return FFGP_OK;
}
