// fp64 MFMA GEMM for gfx950: C = alpha * op(A) op(B) + beta * C on v_mfma_f64_16x16x4_f64.
//
// This one kernel carries every O(N^3) stage of the GP hot path: the trailing SYRK update of the blocked
// Cholesky (the roofline kernel), panel updates, TRSM-as-GEMM with pre-inverted diagonal blocks, TRTRI, LAUUM
// and the dense products of the gradient / prediction paths.
//
// Geometry: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each wave 64x64 = 4x4 MFMA tiles,
// 16 accumulators x 4 fp64 = 128 VGPRs), BK = 16 (4 MFMA k-steps), double-buffered LDS, one barrier per
// k-tile, global->register->LDS staging issued a full compute phase ahead (the 64-cycle fp64 MFMA leaves
// the load/LDS pipes almost idle, so a deeper pipeline buys nothing).  2 workgroups per CU (2 x 72 KiB LDS,
// <= 256 VGPRs) so one workgroup's C epilogue overlaps the other's MFMA stream.
//
// LDS images (both bank-conflict-free for the ds_read_b64 operand fetch, see MI355X guide "LDS"):
//   K-major operand (global rows have k contiguous):  [row][16] doubles, 16-byte chunks XOR-swizzled with
//       (row>>1)&7, so the 16 rows x 2 k-values a 32-lane half reads cover all 64 banks exactly once;
//   MN-major operand (global rows have m/n contiguous): [k][128+16] doubles; the 16-double pad shifts
//       consecutive k rows by 128 B = half the bank row.
// Workgroup -> tile map: XCD-aware (blocks b, b+8, ... share an XCD/L2, so each XCD gets a contiguous chunk
// of the tile order) over 8-tile-row bands walked column-major, so the ~64 tiles resident on one XCD share
// 8 A panels and ~8 B panels through its L2.
#include <algorithm>

#include "ffgp_internal.h"

#define BK 16
#ifndef FFGP_PD_SMALL
#define FFGP_PD_SMALL 4   // register prefetch depth (k-tiles) of the latency-shape tiles
#endif

// tools/trace_gemm.py builds a second library with -DFFGP_GEMM_TRACE: every workgroup stamps s_memtime at the
// phase boundaries of each tile (never compiled into libffgp.so)
#ifdef FFGP_GEMM_TRACE
__device__ unsigned long long* ffgp_trace_buf = nullptr;
extern "C" int ffgp_debug_set_trace(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(ffgp_trace_buf), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#define FFGP_TRACE(slot)                                                                            \
  do {                                                                                              \
    if (tid == 0 && ffgp_trace_buf) {                                                               \
      ffgp_trace_buf[(size_t)bid * 8 + (slot)] = __builtin_readcyclecounter();                      \
      if ((slot) == 0) {                                                                            \
        unsigned hwid = 0, xcc = 0;                                                                 \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));                          \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                          \
        ffgp_trace_buf[(size_t)bid * 8 + 6] = ((unsigned long long)xcc << 32) | hwid;               \
        ffgp_trace_buf[(size_t)bid * 8 + 7] = wall_clock64();                                       \
      }                                                                                             \
    }                                                                                               \
  } while (0)
#else
#define FFGP_TRACE(slot)
#endif

// Tile geometry: TS x TS output tile per 256-thread workgroup, 4 waves as 2 x 2, each wave (TS/2) x (TS/2)
//   TS = 128: 4 x 4 MFMA tiles per wave (128 accumulator VGPRs), 72 KiB LDS, 2 workgroups per CU -- the throughput shape
//   TS =  64: 2 x 2 MFMA tiles per wave, 20 KiB LDS, up to 4 workgroups per CU -- the latency shape: a quarter of the
//             per-tile MFMA chain and 4x the tiles, used when a launch has fewer 128-tiles than the chip has slots
//             (panel TRSM/updates of the look-ahead chain, the tail of the factorisation)
//   TM x TN = 64 x 128 (FULL tiles only): the in-place TRSM shape -- one column tile spans the whole 128-wide panel
//             block, so every workgroup re-writes only rows nobody else reads
template <int TS>
struct Geo {                                    // per-operand geometry (TS = tile extent of that operand)
  static constexpr int WT = TS / 32;            // MFMA tiles per wave along this dimension
  static constexpr int NLD = TS / 32;           // 16-byte loads per thread per operand tile
  static constexpr int MNLD = TS + 16;          // leading dimension of the MN-major LDS image
  static constexpr int OPBUF = 16 * (TS + 16);  // doubles per operand buffer (>= TS*16)
};

__device__ __forceinline__ d2_t ld2_guard(const double* p, int rem, bool vec) {
  d2_t v = {0.0, 0.0};
  if (rem >= 2) {
    if (vec) {
      v = *reinterpret_cast<const d2_t*>(p);
    } else {  // odd leading dimension / unaligned base: two 8-byte loads
      v.x = p[0];
      v.y = p[1];
    }
  } else if (rem == 1) {
    v.x = p[0];
  }
  return v;
}

// GUARD = false: interior tile (all rows/cols in range, full k-tiles, 16-byte aligned operands): straight-line
// vector loads, no per-element branches.
template <int OP, int TS, bool GUARD>
__device__ __forceinline__ void gload(const double* __restrict__ P, int ld, int r0, int R, int k0, int K, int tid,
                                      bool vec, d2_t (&v)[Geo<TS>::NLD]) {
#pragma unroll
  for (int i = 0; i < Geo<TS>::NLD; ++i) {
    const int idx = tid + 256 * i;
    d2_t z = {0.0, 0.0};
    if (OP == OP_KMAJOR) {
      const int row = idx >> 3, ch = idx & 7;
      const int gr = r0 + row, gk = k0 + ch * 2;
      if (GUARD)
        v[i] = (gr < R) ? ld2_guard(P + (size_t)gr * ld + gk, K - gk, vec) : z;
      else
        v[i] = *reinterpret_cast<const d2_t*>(P + (size_t)gr * ld + gk);
    } else {
      const int kk = idx / (TS / 2), c2 = idx % (TS / 2);
      const int gk = k0 + kk, gr = r0 + c2 * 2;
      if (GUARD)
        v[i] = (gk < K) ? ld2_guard(P + (size_t)gk * ld + gr, R - gr, vec) : z;
      else
        v[i] = *reinterpret_cast<const d2_t*>(P + (size_t)gk * ld + gr);
    }
  }
}

template <int OP, int TS>
__device__ __forceinline__ void sstore(double* s, int tid, const d2_t (&v)[Geo<TS>::NLD]) {
#pragma unroll
  for (int i = 0; i < Geo<TS>::NLD; ++i) {
    const int idx = tid + 256 * i;
    if (OP == OP_KMAJOR) {
      const int row = idx >> 3, ch = idx & 7;
      *reinterpret_cast<d2_t*>(s + row * 16 + ((ch ^ ((row >> 1) & 7)) << 1)) = v[i];
    } else {
      const int kk = idx / (TS / 2), c2 = idx % (TS / 2);
      *reinterpret_cast<d2_t*>(s + kk * Geo<TS>::MNLD + c2 * 2) = v[i];
    }
  }
}

// offset (doubles) of this lane's operand element for MFMA k-step kq, sub-tile 0 of the wave's rows
template <int OP, int TS>
__device__ __forceinline__ void frag_offsets(int lane, int wbase, int (&off)[4]) {
#pragma unroll
  for (int kq = 0; kq < 4; ++kq) {
    if (OP == OP_KMAJOR) {
      const int row = wbase + (lane & 15);
      const int c0 = (lane >> 5) ^ ((lane & 15) >> 1);
      off[kq] = row * 16 + ((((kq << 1) ^ c0)) << 1) + ((lane >> 4) & 1);
    } else {
      off[kq] = (kq * 4 + (lane >> 4)) * Geo<TS>::MNLD + wbase + (lane & 15);
    }
  }
}

// Tile order -> (ti, tj).  Everything here is wave-uniform; it is written without integer divisions on the common
// path (full 8-row bands) so it stays on the scalar unit: a workgroup's prologue shares its SIMDs with the MFMA
// stream of the other resident workgroup, and every vector instruction it needs waits behind a 64-cycle MFMA.
__device__ __forceinline__ void decode_tile(int t, int mode, int tiles_m, int tiles_n, int& ti, int& tj, const int GL = 3) {
  const int G = 1 << GL;
  if (mode == TILES_FULL) {
    const int band_sz = G * tiles_n;
    int r0 = 0;
    while (t >= band_sz) {
      t -= band_sz;
      r0 += G;
    }
    const int hgt = min(G, tiles_m - r0);
    if (hgt == G) {
      tj = t >> GL;
      ti = r0 + (t & (G - 1));
    } else {
      tj = t / hgt;
      ti = r0 + t % hgt;
    }
  } else {
    // lower trapezoid (m >= n, origin on the diagonal): tile row ti owns columns 0..min(ti, tiles_n-1)
    int r0 = 0, hgt = 0, fc = 0;
    while (true) {
      hgt = min(G, tiles_m - r0);
      fc = min(r0, tiles_n);                      // columns every row of the band owns
      const int nc = max(0, min(hgt, tiles_n - r0));  // rows of the band that reach the diagonal
      const int cnt = hgt * fc + nc * hgt - nc * (nc - 1) / 2;
      if (t < cnt) break;
      t -= cnt;
      r0 += hgt;
    }
    if (t < hgt * fc) {
      if (hgt == G) {
        tj = t >> GL;
        ti = r0 + (t & (G - 1));
      } else {
        tj = t / hgt;
        ti = r0 + t % hgt;
      }
    } else {
      t -= hgt * fc;
      int c = 0;
      while (t >= hgt - c) {
        t -= hgt - c;
        ++c;
      }
      tj = r0 + c;
      ti = r0 + c + t;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Fast form: interior tile, 16-byte aligned operands, alpha = +-1, beta in {0, 1}, no partial store.
// Every global address is  (uniform 64-bit base in SGPRs) + (loop-invariant 32-bit per-lane offset) + immediate,
// so a tile's prologue and epilogue contain almost no vector-ALU instructions: the C tile is loaded straight into
// the accumulators while the first operand tiles are in flight, alpha = -1 is the MFMA's own negate-A modifier,
// and the epilogue is 64 plain stores.
// ------------------------------------------------------------------------------------------------------------
template <int OP, int TS>
__device__ __forceinline__ void lane_byte_offsets(int ld, int tid, unsigned (&voff)[Geo<TS>::NLD]) {
#pragma unroll
  for (int i = 0; i < Geo<TS>::NLD; ++i) {
    const int idx = tid + 256 * i;
    if (OP == OP_KMAJOR) {
      const int row = idx >> 3, ch = idx & 7;
      voff[i] = (unsigned)(row * ld + ch * 2) * 8u;
    } else {
      const int kk = idx / (TS / 2), c2 = idx % (TS / 2);
      voff[i] = (unsigned)(kk * ld + c2 * 2) * 8u;
    }
  }
}

template <int TS>
__device__ __forceinline__ void gload_fast(const char* __restrict__ base, const unsigned (&voff)[Geo<TS>::NLD],
                                           d2_t (&v)[Geo<TS>::NLD]) {
#pragma unroll
  for (int i = 0; i < Geo<TS>::NLD; ++i) v[i] = *reinterpret_cast<const d2_t*>(base + voff[i]);
}

// doubles per operand buffer: the K-major image is exactly [TS][16]; the MN-major one is [16][TS + 16]
template <int OP, int TS>
constexpr int opbuf() { return OP == OP_KMAJOR ? TS * 16 : Geo<TS>::OPBUF; }

// PD = register prefetch depth in k-tiles.  The 128x128 tile keeps one k-tile in flight (64 MFMAs per wave per k-tile
// cover any memory latency, and its registers are spoken for); the small latency-shape tiles have only 4-16 MFMAs per
// k-tile, far less than an L2/HBM round trip, so they keep PD k-tiles of operands in flight in (cheap) registers --
// otherwise every one of their K/16 steps costs a full memory latency and a K = 512 launch of 32x32 tiles takes 30 us.
template <int OPA, int OPB, int TM, int TN, bool NEG, int PD>
__device__ __forceinline__ void gemm_tile_fast(const char* __restrict__ baseA, const char* __restrict__ baseB, size_t stepA,
                                               size_t stepB, char* __restrict__ baseC, size_t row4_bytes, unsigned voffC,
                                               bool load_c, double* smem, int nkt, int tid,
                                               const unsigned (&voffA)[Geo<TM>::NLD], const unsigned (&voffB)[Geo<TN>::NLD],
                                               const int (&offA)[4], const int (&offB)[4], int trace_bid) {
  constexpr int WM = Geo<TM>::WT, WN = Geo<TN>::WT;
  constexpr int BUFA = opbuf<OPA, TM>(), BUFB = opbuf<OPB, TN>(), STAGE = BUFA + BUFB;
  constexpr int subA = (OPA == OP_KMAJOR) ? 256 : 16;
  constexpr int subB = (OPB == OP_KMAJOR) ? 256 : 16;
  static_assert(PD == 1 || (PD & 1) == 0, "prefetch depth must be 1 or even (LDS stage parity is compile-time)");
  d2_t ra[PD][Geo<TM>::NLD], rb[PD][Geo<TN>::NLD];
#pragma unroll
  for (int s = 0; s < PD; ++s) {
    if (s < nkt) {
      gload_fast<TM>(baseA + s * stepA, voffA, ra[s]);
      gload_fast<TN>(baseB + s * stepB, voffB, rb[s]);
    }
  }
  d4_t acc[WM][WN];
  if (load_c) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const char* rowb = baseC + (size_t)(i * 4 + r) * row4_bytes;
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j][r] = *reinterpret_cast<const double*>(rowb + voffC + j * 128);
      }
  } else {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};
  }
  sstore<OPA, TM>(smem, tid, ra[0]);
  sstore<OPB, TN>(smem + BUFA, tid, rb[0]);
  __syncthreads();
#ifdef FFGP_GEMM_TRACE
  if (tid == 0 && ffgp_trace_buf) ffgp_trace_buf[(size_t)trace_bid * 8 + 1] = __builtin_readcyclecounter();
#endif
  baseA += (size_t)PD * stepA;   // next tile to fetch
  baseB += (size_t)PD * stepB;
  for (int kt0 = 0; kt0 < nkt; kt0 += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      const int kt = kt0 + u;
      if (kt >= nkt) break;
      const int par = (PD == 1) ? (kt & 1) : (u & 1);
      const double* sA = smem + par * STAGE;
      const double* sB = sA + BUFA;
      if (kt + PD < nkt) {   // slot u was drained into LDS one step ago
        gload_fast<TM>(baseA, voffA, ra[u]);
        gload_fast<TN>(baseB, voffB, rb[u]);
        baseA += stepA;
        baseB += stepB;
      }
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {
        double a[WM], b[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) a[i] = sA[offA[kq] + i * subA];
#pragma unroll
        for (int j = 0; j < WN; ++j) b[j] = sB[offB[kq] + j * subB];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int j = 0; j < WN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, NEG ? 1 : 0);  // blgp bit 0: -A
      }
      if (kt + 1 < nkt) {
        double* dA = smem + (par ^ 1) * STAGE;
        sstore<OPA, TM>(dA, tid, ra[(u + 1) % PD]);
        sstore<OPB, TN>(dA + BUFA, tid, rb[(u + 1) % PD]);
      }
      __syncthreads();
    }
  }
#ifdef FFGP_GEMM_TRACE
  if (tid == 0 && ffgp_trace_buf) ffgp_trace_buf[(size_t)trace_bid * 8 + 2] = __builtin_readcyclecounter();
#endif
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      char* rowb = baseC + (size_t)(i * 4 + r) * row4_bytes;
#pragma unroll
      for (int j = 0; j < WN; ++j) *reinterpret_cast<double*>(rowb + voffC + j * 128) = acc[i][j][r];
    }
}

// ------------------------------------------------------------------------------------------------------------
// Direct form (round 5; option syrk_direct; measured 11 % SLOWER at C3 -- twice the operand traffic from L2 / Infinity Cache per MFMA,
// docs/experiments.md -- development build only): the trailing update's interior 128 x 128 tiles WITHOUT LDS and WITHOUT barriers.
// The panel P (rows x K, K-major) is first re-written in the MFMA operand layout -- per 16-row block and per 8 k-columns one
// 1 KiB record: lane l = 16 kk + r holds { P[16 rb + r][8 kp + kk], P[16 rb + r][8 kp + 4 + kk] } -- which serves as the A operand
// of one MFMA pair AND, the update being P_r P_c^T, as the B operand.  Every wave then streams its four row-block records and four
// column-block records per k-pair straight from L2 into registers (one global_load_dwordx4 of 1 KiB contiguous per record, two
// k-pairs in flight) and is on its own: the LDS form's barrier per k-tile, its LDS stores and reads, and the lock-step of a
// workgroup's four waves are gone.  Same MFMAs in the same k order on the same accumulators (C loaded into them first, -A by the
// instruction's modifier): the values are the fast form's bit for bit.  Diagonal and edge tiles keep the LDS forms.
// ------------------------------------------------------------------------------------------------------------
#ifdef FFGP_DEV_OPTIONS
__global__ __launch_bounds__(256) void ffgp_pack_panel_kernel(const double* __restrict__ P, int ld, int rows, int K, d2_t* __restrict__ out) {
  // workgroup = (row block rb, chunk of 128 k-columns): thread t reads 8 consecutive k of row t >> 4 (a row's 16 threads read 1 KiB)
  const int rb = blockIdx.x, kc = blockIdx.y, t = threadIdx.x;
  const int r = t >> 4, c8 = t & 15;
  const int row = rb * 16 + r, k0 = kc * 128 + c8 * 8;
  if (k0 >= K) return;
  d2_t v[4];
  const bool ok = row < rows;
  const double* src = P + (size_t)(ok ? row : 0) * ld + k0;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = ok ? *reinterpret_cast<const d2_t*>(src + 2 * i) : (d2_t){0.0, 0.0};
  const double e[8] = {v[0].x, v[0].y, v[1].x, v[1].y, v[2].x, v[2].y, v[3].x, v[3].y};
  const int KP = K >> 3, kp = k0 >> 3;
  d2_t* dst = out + ((size_t)rb * KP + kp) * 64 + r;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) dst[kk * 16] = (d2_t){e[kk], e[4 + kk]};
}

template <bool NEG>
__device__ __forceinline__ void gemm_tile_direct(const GemmArgs& p, double* __restrict__ Cg, int ti, int tj, int tid) {
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;   // (uniform: scalar bases)
  const int KP = p.k >> 3;
  const size_t rec = (size_t)KP * 1024;                        // bytes of one row block's records
  const char* __restrict__ pk = reinterpret_cast<const char*>(p.pack);
  const char* baseA = pk + (size_t)(ti * 8 + wm * 4) * rec;    // this wave's four row blocks (A operand) ...
  const char* baseB = pk + (size_t)(tj * 8 + wn * 4) * rec;    // ... and four column blocks (B operand): wave-uniform
  const unsigned vo = (unsigned)lane * 16u;
  // C quadrant straight into the accumulators (layout as in gemm_tile_fast)
  char* bC = reinterpret_cast<char*>(Cg) + ((size_t)(ti * 128) * p.ldc + tj * 128) * 8;
  const size_t row4 = (size_t)p.ldc * 32;
  const unsigned voffC = (unsigned)((wm * 64 + (lane >> 4)) * p.ldc + wn * 64 + (lane & 15)) * 8u;
  d4_t acc[4][4];
  d2_t a0[4], b0[4], a1[4], b1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a0[i] = *reinterpret_cast<const d2_t*>(baseA + i * rec + vo);
    b0[i] = *reinterpret_cast<const d2_t*>(baseB + i * rec + vo);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a1[i] = *reinterpret_cast<const d2_t*>(baseA + i * rec + 1024 + vo);
    b1[i] = *reinterpret_cast<const d2_t*>(baseB + i * rec + 1024 + vo);
  }
  if (p.beta != 0.0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const char* rowb = bC + (size_t)(i * 4 + r) * row4;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j][r] = *reinterpret_cast<const double*>(rowb + voffC + j * 128);
      }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};
  }
  baseA += 2048;
  baseB += 2048;
  for (int kp = 0; kp < KP; kp += 2) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(h ? a0[i].y : a0[i].x, h ? b0[j].y : b0[j].x, acc[i][j], 0, 0, NEG ? 1 : 0);
    if (kp + 2 < KP) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a0[i] = *reinterpret_cast<const d2_t*>(baseA + i * rec + vo);
        b0[i] = *reinterpret_cast<const d2_t*>(baseB + i * rec + vo);
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(h ? a1[i].y : a1[i].x, h ? b1[j].y : b1[j].x, acc[i][j], 0, 0, NEG ? 1 : 0);
    if (kp + 3 < KP) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a1[i] = *reinterpret_cast<const d2_t*>(baseA + i * rec + 1024 + vo);
        b1[i] = *reinterpret_cast<const d2_t*>(baseB + i * rec + 1024 + vo);
      }
    }
    baseA += 2048;
    baseB += 2048;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      char* rowb = bC + (size_t)(i * 4 + r) * row4;
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<double*>(rowb + voffC + j * 128) = acc[i][j][r];
    }
}

#endif   // FFGP_DEV_OPTIONS (direct form)

template <int OPA, int OPB, int TM, int TN, bool GUARD>
__device__ __forceinline__ void gemm_mainloop(const GemmArgs& p, const double* __restrict__ Ag, const double* __restrict__ Bg,
                                              double* smem, int m0, int n0, int kt0, int kt1, int tid,
                                              const int (&offA)[4], const int (&offB)[4],
                                              d4_t (&acc)[Geo<TM>::WT][Geo<TN>::WT]) {
  constexpr int WM = Geo<TM>::WT, WN = Geo<TN>::WT;
  constexpr int BUFA = opbuf<OPA, TM>(), BUFB = opbuf<OPB, TN>(), STAGE = BUFA + BUFB;
  constexpr int subA = (OPA == OP_KMAJOR) ? 256 : 16;
  constexpr int subB = (OPB == OP_KMAJOR) ? 256 : 16;
  d2_t ra[Geo<TM>::NLD], rb[Geo<TN>::NLD];
  gload<OPA, TM, GUARD>(Ag, p.lda, m0, p.m, kt0 * BK, p.k, tid, p.avec != 0, ra);
  gload<OPB, TN, GUARD>(Bg, p.ldb, n0, p.n, kt0 * BK, p.k, tid, p.bvec != 0, rb);
  sstore<OPA, TM>(smem, tid, ra);
  sstore<OPB, TN>(smem + BUFA, tid, rb);
  __syncthreads();
  for (int kt = kt0; kt < kt1; ++kt) {
    const int par = (kt - kt0) & 1;
    const double* sA = smem + par * STAGE;
    const double* sB = sA + BUFA;
    const bool more = (kt + 1 < kt1);
    if (more) {
      gload<OPA, TM, GUARD>(Ag, p.lda, m0, p.m, (kt + 1) * BK, p.k, tid, p.avec != 0, ra);
      gload<OPB, TN, GUARD>(Bg, p.ldb, n0, p.n, (kt + 1) * BK, p.k, tid, p.bvec != 0, rb);
    }
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      double a[WM], b[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) a[i] = sA[offA[kq] + i * subA];
#pragma unroll
      for (int j = 0; j < WN; ++j) b[j] = sB[offB[kq] + j * subB];
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      double* dA = smem + (par ^ 1) * STAGE;
      sstore<OPA, TM>(dA, tid, ra);
      sstore<OPB, TN>(dA + BUFA, tid, rb);
    }
    __syncthreads();
  }
}

// One output tile (ti, tj) of shape TM x TN: k loop + epilogue.  Everything that depends on the tile shape lives here so
// that a launch can mix shapes (see the split tail in ffgp_gemm_f64).
template <int OPA, int OPB, int MODE, int TM, int TN>
__device__ __forceinline__ void gemm_one_tile(const GemmArgs& p, const double* __restrict__ Ag, const double* __restrict__ Bg,
                                              double* __restrict__ Cg, double* smem, int ti, int tj, const int tid, const int bid) {
  constexpr int WM = Geo<TM>::WT, WN = Geo<TN>::WT;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int offA[4], offB[4];
  frag_offsets<OPA, TM>(lane, wm * (TM / 2), offA);
  frag_offsets<OPB, TN>(lane, wn * (TN / 2), offB);
  const bool fast_ab = p.fast && p.avec && p.bvec;
  const int m0 = ti * TM, n0 = tj * TN;

  // k range of this tile (triangular operands skip the k-tiles that are structurally zero)
  int kbeg = 0, kend = p.k;
  if (p.lo_i) kbeg = max(kbeg, ti * TM);
  if (p.lo_j) kbeg = max(kbeg, tj * TN);
  if (p.hi_i) kend = min(kend, (ti + 1) * TM);
  if (p.hi_j) kend = min(kend, (tj + 1) * TN);
  const int kt0 = kbeg / BK;
  const int kt1 = (kend + BK - 1) / BK;

  const bool interior = (m0 + TM <= p.m) && (n0 + TN <= p.n) && (kt1 * BK <= p.k) && p.avec && p.bvec;
  if constexpr (TM == 128 && TN == 128 && OPA == OP_KMAJOR && OPB == OP_KMAJOR) {
    // (direct form active: the interior off-diagonal tiles belong to ffgp_gemm_f64_direct's launch)
    if (p.pack && fast_ab && interior && kt0 == 0 && kt1 * BK == p.k && !(MODE == TILES_LOWER && m0 < n0 + TN)) return;
  }
  if (fast_ab && interior && kt0 < kt1 && !(MODE == TILES_LOWER && m0 < n0 + TN)) {
    const size_t k0 = (size_t)kt0 * BK;
    const char* bA = reinterpret_cast<const char*>(Ag) +
                     ((OPA == OP_KMAJOR) ? ((size_t)m0 * p.lda + k0) : (k0 * p.lda + m0)) * 8;
    const char* bB = reinterpret_cast<const char*>(Bg) +
                     ((OPB == OP_KMAJOR) ? ((size_t)n0 * p.ldb + k0) : (k0 * p.ldb + n0)) * 8;
    const size_t stepA = (OPA == OP_KMAJOR) ? (size_t)BK * 8 : (size_t)BK * p.lda * 8;
    const size_t stepB = (OPB == OP_KMAJOR) ? (size_t)BK * 8 : (size_t)BK * p.ldb * 8;
    char* bC = reinterpret_cast<char*>(Cg) + ((size_t)m0 * p.ldc + n0) * 8;
    const size_t row4 = (size_t)p.ldc * 32;   // 4 rows of C
    constexpr int PD = (TM == 128) ? 1 : FFGP_PD_SMALL;      // (128-row tiles are throughput shapes: 128 x 128, and 128 x 64 below)
    unsigned voffA[Geo<TM>::NLD], voffB[Geo<TN>::NLD];   // per-lane byte offsets (a dozen VALU ops per tile)
    lane_byte_offsets<OPA, TM>(p.lda, tid, voffA);
    lane_byte_offsets<OPB, TN>(p.ldb, tid, voffB);
    const unsigned voffC = (unsigned)((wm * (TM / 2) + (lane >> 4)) * p.ldc + wn * (TN / 2) + (lane & 15)) * 8u;
    if (p.alpha < 0.0)
      gemm_tile_fast<OPA, OPB, TM, TN, true, PD>(bA, bB, stepA, stepB, bC, row4, voffC, p.beta != 0.0, smem, kt1 - kt0, tid, voffA,
                                             voffB, offA, offB, bid);
    else
      gemm_tile_fast<OPA, OPB, TM, TN, false, PD>(bA, bB, stepA, stepB, bC, row4, voffC, p.beta != 0.0, smem, kt1 - kt0, tid, voffA,
                                              voffB, offA, offB, bid);
    FFGP_TRACE(3);
    return;
  }

  d4_t acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};

  if (kt0 < kt1) {
    if (interior)
      gemm_mainloop<OPA, OPB, TM, TN, false>(p, Ag, Bg, smem, m0, n0, kt0, kt1, tid, offA, offB, acc);
    else
      gemm_mainloop<OPA, OPB, TM, TN, true>(p, Ag, Bg, smem, m0, n0, kt0, kt1, tid, offA, offB, acc);
  }

  FFGP_TRACE(2);
  // epilogue: lane holds rows (lane>>4)+4r, column lane&15 of each 16x16 accumulator tile
  // (loads batched per 16-row group: addresses are clamped in-bounds so the loads are unconditional and
  //  the compiler batches them instead of one vmcnt(0) round trip per element)
  const double alpha = p.alpha, beta = p.beta;
  const bool use_c = (beta != 0.0);
  const int rbase0 = m0 + wm * (TM / 2) + (lane >> 4);
  const int cbase = n0 + wn * (TN / 2) + (lane & 15);
  // software-pipelined over the WM 16-row groups: the C loads of group i+1 are in flight while group i is
  // combined and stored
  double cv[2][4][WN];
  auto load_group = [&](int i, double (&dst)[4][WN]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int rowc = min(rbase0 + i * 16 + 4 * r, p.m - 1);
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int colc = min(cbase + j * 16, p.n - 1);
        dst[r][j] = Cg[(size_t)rowc * p.ldc + colc];
      }
    }
  };
  if (use_c) {
    load_group(0, cv[0]);
  } else {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < WN; ++j) cv[b][r][j] = 0.0;
  }
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    if (use_c && i + 1 < WM) load_group(i + 1, cv[(i + 1) & 1]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rbase0 + i * 16 + 4 * r;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int col = cbase + j * 16;
        if (row < p.m && col < p.n && (MODE != TILES_LOWER || col <= row)) {
          Cg[(size_t)row * p.ldc + col] = alpha * acc[i][j][r] + beta * cv[i & 1][r][j];
        }
      }
    }
  }
  FFGP_TRACE(3);
}

// the work of one workgroup of a GEMM launch: block id -> tile (or quarter tile of the split tail) -> gemm_one_tile
template <int OPA, int OPB, int MODE, int TM, int TN>
__device__ __forceinline__ void gemm_block(const GemmArgs& p, const double* __restrict__ Ag, const double* __restrict__ Bg,
                                           double* __restrict__ Cg, double* smem, const int tid, const int bid) {
  // One workgroup per tile.  (A persistent 2-per-CU grid was measured and dropped: it keeps the two workgroups of
  // a CU in lock-step, so their prologues and epilogues coincide instead of hiding under each other's k loop.)
  FFGP_TRACE(0);
  if constexpr (TM == 128 && TN == 128) {
    // Split tail: equal-sized tiles finish in rounds of (resident workgroups) and the last round is mostly idle CUs.
    // The launcher may therefore hand the last `split` 128-tiles of the tile order out as quarter tiles (64 x 64), placed
    // at the END of the grid: they start as slots free up under the last full round and level the finish line.
    if (bid >= p.split_at) {
      const int q = bid - p.split_at;
      int Ti, Tj;
      decode_tile(p.split_at + (q >> 2), MODE, p.tiles_m, p.tiles_n, Ti, Tj, p.band_log2);
      const int ti = __builtin_amdgcn_readfirstlane(2 * Ti + ((q >> 1) & 1));
      const int tj = __builtin_amdgcn_readfirstlane(2 * Tj + (q & 1));
      if (ti * 64 >= p.m || tj * 64 >= p.n || (MODE == TILES_LOWER && tj > ti)) return;
      gemm_one_tile<OPA, OPB, MODE, 64, 64>(p, Ag, Bg, Cg, smem, ti, tj, tid, bid);
      return;
    }
  }
  // XCD-aware bijective remap of the block id, then banded tile order (hardware deals workgroup ids round-robin
  // over the 8 XCDs).
  // (triangular-operand launches have k ranges that shrink along the tile order: giving each XCD a contiguous
  //  chunk would leave all the long tiles on XCD 0, so those launches keep the round-robin block order)
  int t = bid;
  if (!(p.lo_i | p.lo_j | p.hi_i | p.hi_j)) {
    const int nwg = p.total_tiles;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  int ti, tj;
  decode_tile(t, MODE, p.tiles_m, p.tiles_n, ti, tj, p.band_log2);
  // wave-uniform by construction; pin them to SGPRs so every tile base below is scalar arithmetic
  ti = __builtin_amdgcn_readfirstlane(ti);
  tj = __builtin_amdgcn_readfirstlane(tj);
  gemm_one_tile<OPA, OPB, MODE, TM, TN>(p, Ag, Bg, Cg, smem, ti, tj, tid, bid);
}

#ifdef FFGP_DEV_OPTIONS
// the direct form's own launch: the same grid and tile order as the LDS kernel's; a workgroup whose tile is interior and off the
// diagonal computes it (four independent waves, no LDS), every other workgroup leaves -- its tile is the LDS kernel's
template <int MODE>
__global__ __launch_bounds__(256, 2) void ffgp_gemm_f64_direct(GemmArgs p) {
  const int tid = threadIdx.x;
  const int bid = blockIdx.x;
  const int nwg = p.total_tiles;
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  int ti, tj;
  decode_tile(t, MODE, p.tiles_m, p.tiles_n, ti, tj, p.band_log2);
  ti = __builtin_amdgcn_readfirstlane(ti);
  tj = __builtin_amdgcn_readfirstlane(tj);
  const int m0 = ti * 128, n0 = tj * 128;
  if (!((m0 + 128 <= p.m) && (n0 + 128 <= p.n)) || (MODE == TILES_LOWER && m0 < n0 + 128)) return;
  if (p.alpha < 0.0) gemm_tile_direct<true>(p, p.C, ti, tj, tid);
  else gemm_tile_direct<false>(p, p.C, ti, tj, tid);
}
#endif

template <int OPA, int OPB, int MODE, int TAG, int TM, int TN>
__global__ __launch_bounds__(256, 2) void ffgp_gemm_f64(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) double smem[2 * (opbuf<OPA, TM>() + opbuf<OPB, TN>())];
  const int tid = threadIdx.x;
  // (look-ahead hand-off published by this launch's first workgroup: the launches before it on this stream have completed)
  if (p.pub_word && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
    __hip_atomic_store(p.pub_word, p.pub_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (p.prio) __builtin_amdgcn_s_setprio(2);  // panel GEMMs of the look-ahead chain outrank the trailing update
  // batched launches (gridDim.y > 1): identical problems at fixed strides (the levels of the blocked TRTRI)
  const double* __restrict__ Ag = p.A + (size_t)blockIdx.y * p.sA + (size_t)blockIdx.z * p.sA2;
  const double* __restrict__ Bg = p.B + (size_t)blockIdx.y * p.sB + (size_t)blockIdx.z * p.sB2;
  double* __restrict__ Cg = p.C + (size_t)blockIdx.y * p.sC + (size_t)blockIdx.z * p.sC2;
  gemm_block<OPA, OPB, MODE, TM, TN>(p, Ag, Bg, Cg, smem, tid, blockIdx.x);
}

// Half-width throughput tiles (option syrk_h64; experiment of round 5 -- measured neutral at C3, 28.70 ms either way: development build only): every 128 x 128 tile of the trailing update as two 128 x 64
// workgroups with 64 accumulator registers per lane, so that THREE workgroups fit a CU (the 128 x 128 tile's 232 registers allow
// two: while one of them is in its prologue -- C tile and first operand tiles in flight, 8 % of a K = 512 tile's life -- the
// other has the matrix pipe to itself and drives it at ~70 %).  The two halves of a tile are blocks b and b + 8 (one XCD: they
// share the A panel in that L2); tile order and XCD chunks as in gemm_block, on the index of the 128 x 128 tile.
#ifdef FFGP_DEV_OPTIONS
template <int OPA, int OPB, int MODE, int TAG>
__global__ __launch_bounds__(256, 3) void ffgp_gemm_f64_h64(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) double smem[2 * (opbuf<OPA, 128>() + opbuf<OPB, 64>())];
  const int tid = threadIdx.x;
  const double* __restrict__ Ag = p.A + (size_t)blockIdx.y * p.sA;
  const double* __restrict__ Bg = p.B + (size_t)blockIdx.y * p.sB;
  double* __restrict__ Cg = p.C + (size_t)blockIdx.y * p.sC;
  const int bid = blockIdx.x;
  const int half = (bid >> 3) & 1;
  const int mt = ((bid >> 4) << 3) | (bid & 7);
  if (mt >= p.total_tiles) return;
  const int nwg = p.total_tiles;
  const int q = nwg >> 3, r = nwg & 7, xcd = mt & 7;
  const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (mt >> 3);
  int ti, tj;
  decode_tile(t, MODE, p.tiles_m, p.tiles_n, ti, tj, p.band_log2);
  ti = __builtin_amdgcn_readfirstlane(ti);
  tj = __builtin_amdgcn_readfirstlane(2 * tj + half);
  if (tj * 64 >= p.n) return;
  gemm_one_tile<OPA, OPB, MODE, 128, 64>(p, Ag, Bg, Cg, smem, ti, tj, tid, bid);
}
#endif

// Ragged launch (the shared factorisation chain of blocks of DIFFERENT sizes, ffgp_potrf_ragged): gridDim.y members, each with its
// own operands, sizes, leading dimensions and tile counts -- every member's launch as its own single call would make it (same tile
// shape: the launcher groups members by that decision), so that its values are the single call's bit for bit.  gridDim.x is the
// largest member's grid; a member's surplus workgroups leave at once.
template <int OPA, int OPB, int MODE, int TAG, int TM, int TN>
__global__ __launch_bounds__(256, 2) void ffgp_gemm_f64_rag(GemmRag q) {
  __shared__ __attribute__((aligned(16))) double smem[2 * (opbuf<OPA, TM>() + opbuf<OPB, TN>())];
  const int tid = threadIdx.x;
  const GemmRagMember& r = q.mem[blockIdx.y];
  if ((int)blockIdx.x >= r.grid) return;
  if (q.base.prio) __builtin_amdgcn_s_setprio(2);
  GemmArgs p = q.base;
  p.A = r.A; p.B = r.B; p.C = r.C;
  p.m = r.m; p.n = r.n; p.k = r.k;
  p.lda = r.lda; p.ldb = r.ldb; p.ldc = r.ldc;
  p.tiles_m = r.tiles_m; p.tiles_n = r.tiles_n; p.total_tiles = r.total_tiles;
  p.grid = r.grid; p.split_at = r.split_at;
  p.fast = r.fast; p.avec = r.avec; p.bvec = r.bvec;
  gemm_block<OPA, OPB, MODE, TM, TN>(p, p.A, p.B, p.C, smem, tid, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------------------
// host launcher
// ------------------------------------------------------------------------------------------------------------
template <int OPA, int OPB, int MODE, int TAG, int TM, int TN>
static int launch_t(ffgp_handle* h, const GemmArgs& a) {
  // a.pad_lds > 0: reserve that much extra LDS so that only ONE of these workgroups fits a CU ("polite" trailing update,
  // see ffgp_gemm_launch) -- the kernel never touches it
  if (a.pad_lds > 0)   // (set per launch: the attribute is per device context, and this path runs ~10 times per factorisation)
    FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffgp_gemm_f64<OPA, OPB, MODE, TAG, TM, TN>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  hipLaunchKernelGGL((ffgp_gemm_f64<OPA, OPB, MODE, TAG, TM, TN>), dim3(a.grid, a.batch, a.batch2), dim3(256), a.pad_lds, h->stream, a);
  return FFGP_OK;
}

template <int TS>
static int dispatch(ffgp_handle* h, int opa, int opb, int mode, int syrk_tag, const GemmArgs& a) {
  if (syrk_tag) {
    // the trailing update of the blocked Cholesky gets its own instantiation so rocprofv3 --stats separates it
    if (opa == OP_KMAJOR && opb == OP_KMAJOR && mode == TILES_LOWER) return launch_t<OP_KMAJOR, OP_KMAJOR, TILES_LOWER, 1, TS, TS>(h, a);
  } else if (mode == TILES_LOWER) {
    if (opa == OP_KMAJOR && opb == OP_KMAJOR) return launch_t<OP_KMAJOR, OP_KMAJOR, TILES_LOWER, 0, TS, TS>(h, a);
    if (opa == OP_MNMAJOR && opb == OP_MNMAJOR) return launch_t<OP_MNMAJOR, OP_MNMAJOR, TILES_LOWER, 0, TS, TS>(h, a);
  } else {
    if (opa == OP_KMAJOR && opb == OP_KMAJOR) return launch_t<OP_KMAJOR, OP_KMAJOR, TILES_FULL, 0, TS, TS>(h, a);
    if (opa == OP_KMAJOR && opb == OP_MNMAJOR) return launch_t<OP_KMAJOR, OP_MNMAJOR, TILES_FULL, 0, TS, TS>(h, a);
    if (opa == OP_MNMAJOR && opb == OP_MNMAJOR) return launch_t<OP_MNMAJOR, OP_MNMAJOR, TILES_FULL, 0, TS, TS>(h, a);
    if (opa == OP_MNMAJOR && opb == OP_KMAJOR) return launch_t<OP_MNMAJOR, OP_KMAJOR, TILES_FULL, 0, TS, TS>(h, a);
  }
  return FFGP_ERR_ARG;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int count_tiles(int mode, int m, int n, int tsm, int tsn, int& tm, int& tn) {
  tm = (m + tsm - 1) / tsm;
  tn = (n + tsn - 1) / tsn;
  if (mode != TILES_LOWER) return tm * tn;
  int total = 0;
  for (int ti = 0; ti < tm; ++ti) total += (ti + 1 < tn) ? ti + 1 : tn;
  return total;
}

// ------------------------------------------------------------------------------------------------------------
// Split-K for thin products.  K_s^T alpha (nt x d), V^T V (nt x nt), the kernel's input-gradient products: a handful of
// output tiles with k = N -- four workgroups grinding through k = 16384 take 1 ms for 8 MFLOP.  Such launches are cut
// along k into a batched launch (each member one k chunk, partial products into a workspace) and summed in a FIXED
// order by a second kernel, so the result does not depend on scheduling.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ffgp_splitk_reduce(const double* __restrict__ P, int ldp, long sP, int parts, double* __restrict__ C,
                                                         int ldc, int m, int n, double beta) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)m * n) return;
  const int r = (int)(e / n), c = (int)(e - (long)r * n);
  double acc = 0.0;
  const double* p = P + (size_t)r * ldp + c;
  for (int s = 0; s < parts; ++s) acc += p[(size_t)s * sP];
  double* dst = C + (size_t)r * ldc + c;
  *dst = (beta != 0.0) ? acc + beta * *dst : acc;
}

static int ensure_skw(ffgp_handle* h, size_t bytes) {
  if (bytes <= h->skw_bytes) return FFGP_OK;
  if (h->skw) {
    hipStreamSynchronize(h->stream);
    if (h->aux) hipStreamSynchronize(h->aux);
    hipFree(h->skw);
  }
  h->skw = nullptr;
  h->skw_bytes = 0;
  const size_t gran = (size_t)16 << 20;
  const size_t want = (bytes + gran - 1) / gran * gran;
  if (hipMalloc(&h->skw, want) != hipSuccess) return FFGP_ERR_ALLOC;
  h->skw_bytes = want;
  ++h->alloc_epoch;
  return FFGP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Skinny products: C[m x n] with n <= 8 (single-output GPs: alpha = Sigma^-1 y, the d = 1 sweeps).  On the 64 x 64 MFMA tile
// such a launch does 8-64x the arithmetic it needs and, worse, walks its k loop with m / 64 workgroups; it is a
// matrix-vector product and bound by reading A once.  K-major A: one wave per row, lanes across k.  MN-major A (op(A) =
// A^T): lanes across the rows (coalesced), the 8 waves of a workgroup split k and combine through LDS in a fixed order.
// ------------------------------------------------------------------------------------------------------------
struct SkinnyArgs {
  const double* A;
  const double* B;
  double* C;
  int m, n, k, lda, ldb, ldc, opb;
  int kc;            // k range per blockIdx.y (gridDim.y parts, partial results sP doubles apart; kc >= k: one part)
  long sP;
  int klo;           // 1: row r of op(A) is structurally zero for k < r (a lower-triangular factor read as op(A)^T ...): the k loop starts there
  int ldcj;          // stride between the (<= 8) output columns: 1 for C[m x n], ldc of the caller for the transposed use
  double alpha, beta;
};

__device__ __forceinline__ double skinny_b(const SkinnyArgs& p, int kk, int j) {
  return (p.opb == OP_KMAJOR) ? p.B[(size_t)j * p.ldb + kk] : p.B[(size_t)kk * p.ldb + j];
}

template <int NC>
__global__ __launch_bounds__(256) void ffgp_skinny_kmajor(SkinnyArgs p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.m) return;
  const double* __restrict__ a = p.A + (size_t)row * p.lda;
  const int kbeg = blockIdx.y * p.kc, kend = min(p.k, kbeg + p.kc);     // (kc is even)
  double acc[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) acc[j] = 0.0;
#pragma unroll 4
  for (int k0 = max(kbeg, p.klo ? (row & ~127) : 0) + lane * 2; k0 < kend; k0 += 128) {
    const double a0 = a[k0];
    const bool two = k0 + 1 < kend;
    const double a1 = two ? a[k0 + 1] : 0.0;
#pragma unroll
    for (int j = 0; j < NC; ++j)
      if (j < p.n) {
        acc[j] = __builtin_fma(a0, skinny_b(p, k0, j), acc[j]);
        if (two) acc[j] = __builtin_fma(a1, skinny_b(p, k0 + 1, j), acc[j]);
      }
  }
#pragma unroll
  for (int j = 0; j < NC; ++j)
    for (int o = 32; o > 0; o >>= 1) acc[j] += __shfl_down(acc[j], o);
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < NC; ++j)
      if (j < p.n) {
        double* dst = p.C + blockIdx.y * p.sP + (size_t)row * p.ldc + (size_t)j * p.ldcj;
        *dst = (p.beta != 0.0) ? p.alpha * acc[j] + p.beta * *dst : p.alpha * acc[j];
      }
  }
}

template <int NC>
__global__ __launch_bounds__(512) void ffgp_skinny_mnmajor(SkinnyArgs p) {
  __shared__ double red[8][NC][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 64 + lane;
  const int kbeg = blockIdx.y * p.kc, kend = min(p.k, kbeg + p.kc);
  const int k00 = max(kbeg, p.klo ? (int)blockIdx.x * 64 : 0);           // (klo: the workgroup's first row)
  const int kchunk = (max(kend - k00, 0) + 7) / 8;
  const int kb = k00 + w * kchunk, ke = min(kend, kb + kchunk);
  double acc[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) acc[j] = 0.0;
  if (row < p.m) {
    const double* __restrict__ a = p.A + row;
#pragma unroll 8
    for (int kk = kb; kk < ke; ++kk) {
      const double av = a[(size_t)kk * p.lda];
#pragma unroll
      for (int j = 0; j < NC; ++j)
        if (j < p.n) acc[j] = __builtin_fma(av, skinny_b(p, kk, j), acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < NC; ++j) red[w][j][lane] = acc[j];
  __syncthreads();
  if (w == 0 && row < p.m) {
#pragma unroll
    for (int j = 0; j < NC; ++j)
      if (j < p.n) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) s += red[q][j][lane];
        double* dst = p.C + blockIdx.y * p.sP + (size_t)row * p.ldc + (size_t)j * p.ldcj;
        *dst = (p.beta != 0.0) ? p.alpha * s + p.beta * *dst : p.alpha * s;
      }
  }
}

template <int NC>
static void launch_skinny(ffgp_handle* h, int opa, const SkinnyArgs& a, int parts = 1) {
  if (opa == OP_KMAJOR)
    hipLaunchKernelGGL(ffgp_skinny_kmajor<NC>, dim3((a.m + 3) / 4, parts), dim3(256), 0, h->stream, a);
  else
    hipLaunchKernelGGL(ffgp_skinny_mnmajor<NC>, dim3((a.m + 63) / 64, parts), dim3(512), 0, h->stream, a);
}

// The shape decision of one launch: tile shape (128 / 64 / 32-row, see below), fast form, split tail, polite padding -- everything
// between ffgp_gemm_launch's special paths and the launch itself, as a function of the operands' sizes alone, so that a ragged
// launch (ffgp_gemm_launch_rag) can take it member by member exactly as each member's own single launch would.
struct GemmPlan {
  GemmArgs a;
  int tsm, tsn;
  int syrk_tag;     // 1 only when the launch is the 128-tile trailing update (the roofline kernel's own instantiation)
};

static int gemm_plan(ffgp_handle* h, int opa, int opb, int mode, int syrk_tag, const double* A, int lda, const double* B, int ldb, double* C,
                     int ldc, int m, int n, int k, double alpha, double beta, int tri, int alias, int batch, long sA, long sB, long sC,
                     GemmPlan& pl) {
  GemmArgs& a = pl.a;
  a = GemmArgs();
  // vector (16-byte) operand loads need even leading dimensions and 16-byte aligned bases
  a.avec = (!(lda & 1) && aligned16(A)) ? 1 : 0;
  a.bvec = (!(ldb & 1) && aligned16(B)) ? 1 : 0;
  a.A = A; a.B = B; a.C = C;
  a.m = m; a.n = n; a.k = k;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.alpha = alpha; a.beta = beta;
  a.prio = (h->stream == h->aux && h->aux_prio) ? 1 : 0;
  a.batch2 = 1;
  a.sA2 = a.sB2 = a.sC2 = 0;
  // batch < 0: |batch| members, with every shape decision below taken as ONE member's launch would take it -- the batched
  // factorisation (ffgp_nlml_fused_batch) promises the single call's values bit for bit, and the forms differ in rounding (the fast
  // 128-tile accumulates onto C, the general tiles add alpha * (sum) to beta * C once)
  const int dec_batch = (batch < 0) ? 1 : (batch > 1 ? batch : 1);
  if (batch < 0) batch = -batch;
  a.batch = batch > 1 ? batch : 1;
  a.sA = sA; a.sB = sB; a.sC = sC;
  if ((sA & 1) || (sB & 1)) a.avec = a.bvec = 0;  // odd strides break the 16-byte alignment of later batch members
  a.lo_i = (tri & TRI_LO_I) ? 1 : 0;
  a.lo_j = (tri & TRI_LO_J) ? 1 : 0;
  a.hi_i = (tri & TRI_HI_I) ? 1 : 0;
  a.hi_j = (tri & TRI_HI_J) ? 1 : 0;
  // tile shape: the 128-tile is the throughput shape; below ~1.5 tiles per CU the launch is latency-bound and
  // the 64-tile (4x the workgroups, a quarter of the per-tile MFMA chain) finishes sooner; the kernels of the
  // factorisation's dependency chain (K-major operands) go one step further to 32-row tiles when even the 64-tiles
  // would leave most CUs with a single 4-16 us MFMA chain
  int tsm = 128, tsn = 128;
  a.total_tiles = count_tiles(mode, m, n, 128, 128, a.tiles_m, a.tiles_n);
  int level = 0;
  if (h->force_ts == 64) level = 1;
  else if (h->force_ts == 32) level = 2;
  else if (h->force_ts == 0 && a.total_tiles * dec_batch < h->small_tile_threshold) level = 1;
  const bool kk = (opa == OP_KMAJOR && opb == OP_KMAJOR);
  if (level >= 1) {
    if (alias == 0 || (alias == ALIAS_A && n <= 64) || (alias == ALIAS_B && m <= 64)) {
      tsm = tsn = 64;
    } else if (alias == ALIAS_A && kk) {
      tsm = 64;  // 64 x 128: the whole panel-block width in one column tile
    }
    a.total_tiles = count_tiles(mode, m, n, tsm, tsn, a.tiles_m, a.tiles_n);
    if (h->force_ts == 0 && level == 1 && kk && dec_batch == 1 && tsm == 64 && a.total_tiles < h->tile32_threshold) level = 2;
    // 32-row tiles: 32 x 32 for products that alias nothing; an in-place product must keep ONE column tile (C = A's
    // buffer: a second column tile would overwrite columns the first still reads as its k range), so it takes
    // 32 x 128 whatever its width; C = B's buffer (one ROW tile needed) stays on the 64-tile
    if (level == 2 && kk && tsm == 64 && dec_batch == 1 && alias != ALIAS_B) {
      tsm = 32;
      tsn = (alias == ALIAS_A) ? 128 : 32;
      a.total_tiles = count_tiles(mode, m, n, tsm, tsn, a.tiles_m, a.tiles_n);
    }
  }
  // invariant of the in-place products, whatever shape was chosen above: ONE column tile when C is A's buffer, ONE row
  // tile when C is B's (a violation is a data race between workgroups, not an error the GPU would report)
  if ((alias == ALIAS_A && a.tiles_n != 1) || (alias == ALIAS_B && a.tiles_m != 1)) {
    fprintf(stderr, "[ffgp] gemm: in-place launch would be split across %d x %d tiles (alias %d)\n", a.tiles_m, a.tiles_n, alias);
    return FFGP_ERR_ARG;
  }
  // fast form (see gemm_tile_fast): alpha = +-1, beta in {0, 1}; per-lane byte offsets must fit 32 bits
  a.fast = ((alpha == 1.0 || alpha == -1.0) && (beta == 0.0 || beta == 1.0) && (size_t)lda * 8 * 130 < 0xffffffffull &&
            (size_t)ldb * 8 * 130 < 0xffffffffull && (size_t)ldc * 8 * 130 < 0xffffffffull)
               ? 1 : 0;
  a.grid = a.total_tiles;
  a.split_at = 0x7fffffff;
  a.band_log2 = h->band_log2;
  // Split tail (see ffgp_gemm_f64): with T equal tiles on 256 CUs the last (T mod 256) tiles run on otherwise idle CUs for a
  // whole tile time; when that remainder is small, hand it out as 64 x 64 quarters -- 4x the workgroups, a quarter of the
  // chain each -- which start under the last full round.  (An in-place or batched launch never splits.)
  if (tsm == 128 && tsn == 128 && alias == 0 && dec_batch == 1 && h->force_ts == 0 && h->split_rem_max > 0 && a.total_tiles > 256 &&
      !(syrk_tag && (h->syrk_h64 || h->syrk_direct))) {
    const int rem = a.total_tiles % 256;
    if (rem > 0 && rem <= h->split_rem_max) {
      a.split_at = a.total_tiles - rem;
      a.total_tiles = a.split_at;          // the XCD remap permutes the whole-tile part only
      a.grid = a.split_at + 4 * rem;
    }
  }
  // "Polite" trailing update: once the factorisation is bound by its dependency chain (trailing matrix below polite_m
  // rows) the 128-tile SYRK is launched with LDS padding so that only one of its workgroups fits a CU.  Alone it still
  // runs the MFMA pipe at ~70 %, and the other half of every CU -- VGPRs, LDS, issue slots -- is free for the chain's
  // kernels at all times instead of only when a SYRK workgroup happens to exit.
  a.pad_lds = 0;
  if (syrk_tag && tsm == 128 && h->lookahead && h->polite_m > 0 && m < h->polite_m && h->stream != h->aux) a.pad_lds = h->polite_pad_kb * 1024;
  // ... and so are its 64-tile launches (trailing matrices below ~4 600 rows: always the chain-bound regime): 20 KiB of LDS let four of
  // their workgroups sit on a CU, and the chain's first kernels of a panel then run at a third of their speed beside them (the first
  // panel update 60 us instead of 14, the first diagonal block 50 instead of 28: profiles/r05g_c2_chain_timeline.txt).  With 60 KiB of
  // unused LDS two fit: N = 4096 -1.7 %, N = 6144 -3.3 %, N = 8192 -2.0 % (same tiles, same values).  Only in the carry form of a single
  // block's look-ahead (polite64_active, set by ffgp_potrf_impl): the round-1 form above 12 288 rows lost 0.7 % with it at C3, and a
  // shared chain's launches (batched or ragged: throughput-bound) are left alone.
  if (syrk_tag && tsm == 64 && tsn == 64 && h->polite64_active && h->polite64_pad_kb > 0 && h->stream != h->aux && dec_batch == 1 &&
      m - n <= 1024)      // (many passenger rows below the matrix: their updates are throughput work -- d = 4096 lost 1 % with it)
    a.pad_lds = h->polite64_pad_kb * 1024;
  if (syrk_tag && tsm == 32 && tsn == 32 && h->polite64_active && h->polite32_pad_kb > 0 && h->stream != h->aux && dec_batch == 1 && m - n <= 1024)
    a.pad_lds = h->polite32_pad_kb * 1024;
  a.pub_word = nullptr;
  a.pub_val = 0;
  if (h->ho_gdefer_slot >= 0 && h->ho_gdefer_stream == h->stream && h->ob_F <= 1) {      // a pending hand-off rides on this launch
    a.pub_word = h->ho_mem + h->ho_gdefer_slot * 16;
    a.pub_val = h->ho_seq[h->ho_gdefer_slot];
    h->ho_launched[h->ho_gdefer_slot] = h->ho_seq[h->ho_gdefer_slot];      // (the launch being planned publishes: submission-order rule)
    h->ho_gdefer_slot = -1;
  }
  if (tsm != 128) syrk_tag = 0;  // only the 128x128 trailing update is the roofline kernel (own instantiation + stats)
  pl.tsm = tsm; pl.tsn = tsn; pl.syrk_tag = syrk_tag;
  return FFGP_OK;
}

// alias: 0 = C aliases neither operand; ALIAS_A = C is A's buffer (row-wise in place: needs ONE column tile so
// that no other workgroup reads the rows a workgroup re-writes); ALIAS_B = C is B's buffer (needs ONE row tile)
int ffgp_gemm_launch(ffgp_handle* h, int opa, int opb, int mode, int syrk_tag, const double* A, int lda, const double* B,
                     int ldb, double* C, int ldc, int m, int n, int k, double alpha, double beta, int tri, int alias, int batch,
                     long sA, long sB, long sC) {
  if (m <= 0 || n <= 0) return FFGP_OK;
  if (k <= 0) {
    // degenerate: C = beta*C handled by callers (never used on the hot path)
    return FFGP_ERR_ARG;
  }
  if (!A || !B || !C) return FFGP_ERR_ARG;
  if (mode == TILES_LOWER && m < n) return FFGP_ERR_ARG;
  if (alias == ALIAS_A && (n > 128 || mode != TILES_FULL)) return FFGP_ERR_ARG;
  if (alias == ALIAS_B && (m > 128 || mode != TILES_FULL)) return FFGP_ERR_ARG;
  // outer batch (ffgp_handle::ob_F): only the triangular products of the inverse's levels and the LAUUM go through it -- none of
  // which may take one of the special paths below (their single-block launches never do either)
  const bool ob = h->ob_F > 1;
  long ob_sA = 0, ob_sB = 0, ob_sC = 0;
  if (ob) {
    auto stride_of = [&](const double* ptr, long& st) {
      for (int i = 0; i < h->ob_n; ++i)
        if (ptr >= h->ob_rng[i].lo && ptr < h->ob_rng[i].hi) { st = h->ob_rng[i].stride; return true; }
      return false;
    };
    if (!stride_of(A, ob_sA) || !stride_of(B, ob_sB) || !stride_of(C, ob_sC) || ((ob_sA | ob_sB | ob_sC) & 1) || alias != 0 || syrk_tag) {
      fprintf(stderr, "[ffgp] gemm: outer-batch launch with an operand outside the registered ranges\n");
      return FFGP_ERR_ARG;
    }
    if (tri == 0 || m <= 8 || n <= 8) {   // (a ragged last pair of a level can be a one-row product: the single block's own path, block by block)
      const int F = h->ob_F;
      h->ob_F = 0;
      int rc = FFGP_OK;
      for (int f = 0; f < F && rc == FFGP_OK; ++f)
        rc = ffgp_gemm_launch(h, opa, opb, mode, syrk_tag, A + (size_t)f * ob_sA, lda, B + (size_t)f * ob_sB, ldb, C + (size_t)f * ob_sC, ldc,
                              m, n, k, alpha, beta, tri, alias, batch, sA, sB, sC);
      h->ob_F = F;
      return rc;
    }
  }
  // few output ROWS (single-output GPs: A^T = Gamma^T L^-1, 1 x n): the transposed matrix-vector product -- C^T = op(B)^T op(A)^T
  // reads B once with the rows of C^T across the lanes; on the 64 x 64 tile the launch above cost 0.21 ms at n = k = 4096
  // (0.6 TB/s of B) inside every training step of a d = 1 model
  if (h->skinny_max_n > 0 && m <= h->skinny_max_n && m <= 8 && n > 8 && alias == 0 && (batch == 0 || batch == 1) && (tri == 0 || tri == TRI_LO_J) &&
      mode == TILES_FULL && !syrk_tag && h->stream != h->aux) {
    SkinnyArgs sk = {};
    sk.A = B; sk.B = A; sk.C = C;
    sk.m = n; sk.n = m; sk.k = k;
    sk.lda = ldb; sk.ldb = lda; sk.ldc = 1; sk.ldcj = ldc;
    sk.klo = (tri == TRI_LO_J) ? 1 : 0;      // op(B)(k, c) = 0 for k < c: row c of the transposed operand starts at k = c
    sk.opb = (opa == OP_KMAJOR) ? OP_KMAJOR : OP_MNMAJOR;   // op(A)^T (k x m): element (kk, j) at A[j * lda + kk] for a K-major A
    sk.alpha = alpha; sk.beta = beta;
    sk.kc = k + 1; sk.sP = 0;
    const int opa_t = (opb == OP_MNMAJOR) ? OP_MNMAJOR : OP_KMAJOR;   // op(B)^T (n x k): row c, column kk at B[kk * ldb + c] for an MN-major B
    // lanes-across-rows launches have n / 64 workgroups: cut k as well until the chip is full (partials + a fixed-order reduction)
    int parts = 1;
    if (opa_t == OP_MNMAJOR) {
      const int wg = (n + 63) / 64;
      parts = max(1, min(min((2048 + wg - 1) / wg, k / 256), 16));
    }
    if (parts >= 2) {
      const int ldp = (n + 1) & ~1;
      const long sP = (long)m * ldp;
      FFGP_CHECK(ensure_skw(h, (size_t)parts * sP * sizeof(double)));
      sk.C = h->skw;
      sk.ldcj = ldp;
      sk.kc = ((k + parts - 1) / parts + 1) & ~1;
      sk.sP = sP;
      sk.beta = 0.0;
      parts = (k + sk.kc - 1) / sk.kc;
    }
    if (m == 1) launch_skinny<1>(h, opa_t, sk, parts);
    else if (m == 2) launch_skinny<2>(h, opa_t, sk, parts);
    else if (m <= 4) launch_skinny<4>(h, opa_t, sk, parts);
    else launch_skinny<8>(h, opa_t, sk, parts);
    if (parts >= 2) {
      const long total = (long)m * n;
      hipLaunchKernelGGL(ffgp_splitk_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, h->skw, sk.ldcj, sk.sP, parts, C,
                         ldc, m, n, beta);
    }
    return hipGetLastError() == hipSuccess ? FFGP_OK : FFGP_ERR_HIP;
  }
  if (h->splitk_min_k > 0 && alias == 0 && (batch == 0 || batch == 1) && tri == 0 && mode == TILES_FULL && !syrk_tag && k >= h->splitk_min_k &&
      h->stream != h->aux) {
    const int t64 = ((m + 63) / 64) * ((n + 63) / 64);
    if (t64 <= 64) {
      const int ldp = (n + 1) & ~1;
      int parts = min(k / 256, (768 + t64 - 1) / t64);   // chunks of at least 256, enough of them to fill the chip
      parts = (int)min((size_t)parts, ((size_t)64 << 20) / sizeof(double) / ((size_t)m * ldp) - 1);   // workspace cap: 64 MiB
      if (parts >= 2) {
        const int kc = (k / parts) & ~31;                 // chunk length: a multiple of the k tile (and even)
        const int rem = k - kc * parts;                   // < 32 * parts: its own, short, launch
        const long sP = (long)m * ldp;
        FFGP_CHECK(ensure_skw(h, (size_t)(parts + 1) * sP * sizeof(double)));
        const long sA = (opa == OP_KMAJOR) ? kc : (long)kc * lda;
        const long sB = (opb == OP_KMAJOR) ? kc : (long)kc * ldb;
        const int saved = h->splitk_min_k;
        h->splitk_min_k = 0;                              // the chunk launches themselves are never split again
        int rc = ffgp_gemm_launch(h, opa, opb, TILES_FULL, 0, A, lda, B, ldb, h->skw, ldp, m, n, kc, alpha, 0.0, 0, ALIAS_NONE, parts,
                                  sA, sB, sP);
        if (rc == FFGP_OK && rem > 0)
          rc = ffgp_gemm_launch(h, opa, opb, TILES_FULL, 0, A + (size_t)parts * sA, lda, B + (size_t)parts * sB, ldb,
                                h->skw + (size_t)parts * sP, ldp, m, n, rem, alpha, 0.0);
        h->splitk_min_k = saved;
        FFGP_CHECK(rc);
        const long total = (long)m * n;
        hipLaunchKernelGGL(ffgp_splitk_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, h->skw, ldp, sP,
                           parts + (rem > 0 ? 1 : 0), C, ldc, m, n, beta);
        return hipGetLastError() == hipSuccess ? FFGP_OK : FFGP_ERR_HIP;
      }
    }
  }
  if (h->skinny_max_n > 0 && n <= h->skinny_max_n && n <= 8 && alias == 0 && (batch == 0 || batch == 1) && tri == 0 && mode == TILES_FULL && !syrk_tag &&
      h->stream != h->aux) {
    SkinnyArgs sk;
    sk.A = A; sk.B = B; sk.C = C;
    sk.m = m; sk.n = n; sk.k = k;
    sk.lda = lda; sk.ldb = ldb; sk.ldc = ldc; sk.opb = opb;
    sk.ldcj = 1;
    sk.klo = 0;
    sk.kc = k + 1; sk.sP = 0;
    sk.alpha = alpha; sk.beta = beta;
    if (n == 1) launch_skinny<1>(h, opa, sk);
    else if (n == 2) launch_skinny<2>(h, opa, sk);
    else if (n <= 4) launch_skinny<4>(h, opa, sk);
    else launch_skinny<8>(h, opa, sk);
    return hipGetLastError() == hipSuccess ? FFGP_OK : FFGP_ERR_HIP;
  }
  GemmPlan pl;
  FFGP_CHECK(gemm_plan(h, opa, opb, mode, syrk_tag, A, lda, B, ldb, C, ldc, m, n, k, alpha, beta, tri, alias, batch, sA, sB, sC, pl));
  GemmArgs& a = pl.a;
  const int tsm = pl.tsm, tsn = pl.tsn;
  syrk_tag = pl.syrk_tag;
  if (ob) {
    a.batch2 = h->ob_F;
    a.sA2 = ob_sA; a.sB2 = ob_sB; a.sC2 = ob_sC;
  }
  // direct form of the trailing update (see gemm_tile_direct): the panel in the MFMA operand layout, written right before the launch
  a.pack = nullptr;
#ifdef FFGP_DEV_OPTIONS
  if (syrk_tag && h->syrk_direct && A == B && lda == ldb && (k & 15) == 0 && a.batch == 1 && !ob && a.fast && a.avec && a.bvec && tri == 0 &&
      a.split_at == 0x7fffffff) {
    const int rbs = (m + 15) / 16;
    const size_t need = (size_t)rbs * 16 * k * sizeof(double);
    if (need > h->pack_bytes) {
      if (h->pack_buf) {
        hipStreamSynchronize(h->stream);
        if (h->aux) hipStreamSynchronize(h->aux);
        hipFree(h->pack_buf);
        h->pack_buf = nullptr;
        h->pack_bytes = 0;
      }
      if (hipMalloc(&h->pack_buf, need + need / 8) == hipSuccess) h->pack_bytes = need + need / 8;
      else (void)hipGetLastError();
    }
    if (h->pack_buf) {
      // (one buffer per handle: consecutive trailing updates on DIFFERENT streams would race on it -- the look-ahead forms issue
      // them on the main stream only, the side stream's S_a keeps the LDS form)
      if (h->stream != h->aux) {
        if (a.pub_word) {   // (published before the direct kernel, not by the LDS kernel behind it)
          FFGP_HIP(hipStreamWriteValue32(h->stream, a.pub_word, a.pub_val, 0));
          a.pub_word = nullptr;
        }
        hipLaunchKernelGGL(ffgp_pack_panel_kernel, dim3(rbs, (k + 127) / 128), dim3(256), 0, h->stream, A, lda, m, k,
                           reinterpret_cast<d2_t*>(h->pack_buf));
        a.pack = h->pack_buf;
        if (mode == TILES_LOWER) hipLaunchKernelGGL(ffgp_gemm_f64_direct<TILES_LOWER>, dim3(a.grid), dim3(256), 0, h->stream, a);
        else hipLaunchKernelGGL(ffgp_gemm_f64_direct<TILES_FULL>, dim3(a.grid), dim3(256), 0, h->stream, a);
      }
    }
  }
#endif
  // timing == 2: bracket every trailing-update launch with its own event pair (no host sync inside the timed
  // region; ffgp_syrk_stats drains the pool afterwards)
  const bool timed = (syrk_tag && h->timing == 2);
  hipEvent_t ev_stop = nullptr;
  if (timed) {
    if (h->syrk_pool_used + 2 > (int)h->syrk_pool.size()) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      h->syrk_pool.push_back(e0);
      h->syrk_pool.push_back(e1);
    }
    hipEventRecord(h->syrk_pool[h->syrk_pool_used], h->stream);
    ev_stop = h->syrk_pool[h->syrk_pool_used + 1];
    h->syrk_pool_used += 2;
  }
  int rc;
  if (tsm == 64 && tsn == 128)
    rc = launch_t<OP_KMAJOR, OP_KMAJOR, TILES_FULL, 0, 64, 128>(h, a);
  else if (tsm == 32 && tsn == 128)
    rc = launch_t<OP_KMAJOR, OP_KMAJOR, TILES_FULL, 0, 32, 128>(h, a);
  else if (tsm == 32)
    rc = (mode == TILES_LOWER) ? launch_t<OP_KMAJOR, OP_KMAJOR, TILES_LOWER, 0, 32, 32>(h, a)
                               : launch_t<OP_KMAJOR, OP_KMAJOR, TILES_FULL, 0, 32, 32>(h, a);
#ifdef FFGP_DEV_OPTIONS
  else if (tsm == 128 && syrk_tag && h->syrk_h64 && a.split_at == 0x7fffffff && a.pad_lds == 0 && !ob) {
    // (experiment: the trailing update on 128 x 64 half tiles, three workgroups per CU)
    const int gx = ((a.total_tiles + 7) / 8) * 16;
    if (a.pub_word) {   // only ffgp_gemm_f64 stores the word itself: a hand-off the plan picked up is written plainly before any other kernel
      FFGP_HIP(hipStreamWriteValue32(h->stream, a.pub_word, a.pub_val, 0));
      a.pub_word = nullptr;
    }
    hipLaunchKernelGGL((ffgp_gemm_f64_h64<OP_KMAJOR, OP_KMAJOR, TILES_LOWER, 1>), dim3(gx, a.batch), dim3(256), 0, h->stream, a);
    rc = FFGP_OK;
  }
#endif
  else if (tsm == 128)
    rc = dispatch<128>(h, opa, opb, mode, syrk_tag, a);
  else
    rc = dispatch<64>(h, opa, opb, mode, syrk_tag, a);
  if (rc != FFGP_OK) return rc;
  if (hipGetLastError() != hipSuccess) return FFGP_ERR_HIP;
  if (syrk_tag) {
    h->syrk_launches += 1;
    // algorithmic flops: 2k per element of the lower trapezoid (n(n+1)/2 + (m-n)n elements)
    // (a launch of the shared chain covers a.batch blocks, one of the outer-batched stages a.batch2 more: the event pair brackets all of them)
    h->syrk_flops += 2.0 * (double)k * ((double)n * ((double)n + 1.0) * 0.5 + ((double)m - (double)n) * (double)n) * (double)a.batch *
                     (double)a.batch2;
    if (timed) hipEventRecord(ev_stop, h->stream);
  }
  return FFGP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// Ragged launch: R independent products of ONE kind (same operand layouts, tile set, alpha / beta, aliasing) whose sizes, operands
// and leading dimensions differ -- the shared factorisation chain of blocks of different sizes (ffgp_potrf_ragged).  Every member's
// shape decision is gemm_plan's for that member alone, i.e. what its own ffgp_gemm_launch would decide; members that decide alike
// share a launch (gridDim.y, at most FFGP_RAG_MAX per launch), so a member's values are its single call's, bit for bit.  Only the
// K-major products of the chain exist in this form (TRSM by the inverted diagonal block, panel update, trailing update).
// ------------------------------------------------------------------------------------------------------------
template <int MODE, int TAG, int TM, int TN>
static int launch_rag_t(ffgp_handle* h, const GemmRag& q, int gx, int gy) {
  if (q.base.pad_lds > 0)
    FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffgp_gemm_f64_rag<OP_KMAJOR, OP_KMAJOR, MODE, TAG, TM, TN>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  hipLaunchKernelGGL((ffgp_gemm_f64_rag<OP_KMAJOR, OP_KMAJOR, MODE, TAG, TM, TN>), dim3(gx, gy), dim3(256), q.base.pad_lds, h->stream, q);
  return FFGP_OK;
}

int ffgp_gemm_launch_rag(ffgp_handle* h, int mode, int syrk_tag, int R, const GemmRagIn* in, double alpha, double beta, int alias) {
  if (R <= 0) return FFGP_OK;
  if (!in || h->ob_F > 1) return FFGP_ERR_ARG;
  struct Item { GemmPlan pl; long key; };
  std::vector<Item> items;
  items.reserve(R);
  for (int r = 0; r < R; ++r) {
    const GemmRagIn& g = in[r];
    if (g.m <= 0 || g.n <= 0) continue;
    if (g.k <= 0 || !g.A || !g.B || !g.C) return FFGP_ERR_ARG;
    if (mode == TILES_LOWER && g.m < g.n) return FFGP_ERR_ARG;
    if (alias == ALIAS_A && (g.n > 128 || mode != TILES_FULL)) return FFGP_ERR_ARG;
    if (alias == ALIAS_B) return FFGP_ERR_ARG;
    Item it;
    FFGP_CHECK(gemm_plan(h, OP_KMAJOR, OP_KMAJOR, mode, syrk_tag, g.A, g.lda, g.B, g.ldb, g.C, g.ldc, g.m, g.n, g.k, alpha, beta, 0, alias,
                         0, 0, 0, 0, it.pl));
    if (it.pl.a.pub_word) {   // (the ragged kernel does not publish: a hand-off the plan picked up is written plainly, never dropped)
      FFGP_HIP(hipStreamWriteValue32(h->stream, it.pl.a.pub_word, it.pl.a.pub_val, 0));
      it.pl.a.pub_word = nullptr;
    }
    // members share a launch when they chose the same kernel instantiation and launch attributes
    it.key = ((long)it.pl.tsm << 40) | ((long)it.pl.tsn << 28) | ((long)it.pl.syrk_tag << 24) | (long)(it.pl.a.pad_lds >> 10);
    items.push_back(it);
  }
  std::vector<char> done(items.size(), 0);
  for (size_t i0 = 0; i0 < items.size(); ++i0) {
    if (done[i0]) continue;
    // the members of this group, FFGP_RAG_MAX per launch
    std::vector<size_t> grp;
    for (size_t j = i0; j < items.size(); ++j)
      if (!done[j] && items[j].key == items[i0].key) { grp.push_back(j); done[j] = 1; }
    for (size_t g0 = 0; g0 < grp.size(); g0 += FFGP_RAG_MAX) {
      const int cnt = (int)std::min<size_t>(FFGP_RAG_MAX, grp.size() - g0);
      GemmRag q;
      q.base = items[grp[g0]].pl.a;
      int gx = 0;
      double flops = 0.0;
      for (int c = 0; c < cnt; ++c) {
        const GemmArgs& a = items[grp[g0 + c]].pl.a;
        GemmRagMember& mm = q.mem[c];
        mm.A = a.A; mm.B = a.B; mm.C = a.C;
        mm.m = a.m; mm.n = a.n; mm.k = a.k;
        mm.lda = a.lda; mm.ldb = a.ldb; mm.ldc = a.ldc;
        mm.tiles_m = a.tiles_m; mm.tiles_n = a.tiles_n; mm.total_tiles = a.total_tiles; mm.grid = a.grid; mm.split_at = a.split_at;
        mm.fast = a.fast; mm.avec = a.avec; mm.bvec = a.bvec;
        gx = std::max(gx, a.grid);
        flops += 2.0 * (double)a.k * ((double)a.n * ((double)a.n + 1.0) * 0.5 + ((double)a.m - (double)a.n) * (double)a.n);
      }
      for (int c = cnt; c < FFGP_RAG_MAX; ++c) { q.mem[c] = q.mem[0]; q.mem[c].grid = 0; }
      const GemmPlan& pl = items[grp[g0]].pl;
      const int tsm = pl.tsm, tsn = pl.tsn, tag = pl.syrk_tag;
      const bool timed = (tag && h->timing == 2);
      hipEvent_t ev_stop = nullptr;
      if (timed) {
        if (h->syrk_pool_used + 2 > (int)h->syrk_pool.size()) {
          hipEvent_t e0, e1;
          hipEventCreate(&e0);
          hipEventCreate(&e1);
          h->syrk_pool.push_back(e0);
          h->syrk_pool.push_back(e1);
        }
        hipEventRecord(h->syrk_pool[h->syrk_pool_used], h->stream);
        ev_stop = h->syrk_pool[h->syrk_pool_used + 1];
        h->syrk_pool_used += 2;
      }
      int rc = FFGP_ERR_ARG;
      if (mode == TILES_FULL) {
        if (tsm == 128 && tsn == 128) rc = launch_rag_t<TILES_FULL, 0, 128, 128>(h, q, gx, cnt);
        else if (tsm == 64 && tsn == 128) rc = launch_rag_t<TILES_FULL, 0, 64, 128>(h, q, gx, cnt);
        else if (tsm == 32 && tsn == 128) rc = launch_rag_t<TILES_FULL, 0, 32, 128>(h, q, gx, cnt);
        else if (tsm == 64 && tsn == 64) rc = launch_rag_t<TILES_FULL, 0, 64, 64>(h, q, gx, cnt);
        else if (tsm == 32 && tsn == 32) rc = launch_rag_t<TILES_FULL, 0, 32, 32>(h, q, gx, cnt);
      } else {
        if (tsm == 128 && tag) rc = launch_rag_t<TILES_LOWER, 1, 128, 128>(h, q, gx, cnt);
        else if (tsm == 128) rc = launch_rag_t<TILES_LOWER, 0, 128, 128>(h, q, gx, cnt);
        else if (tsm == 64) rc = launch_rag_t<TILES_LOWER, 0, 64, 64>(h, q, gx, cnt);
        else if (tsm == 32) rc = launch_rag_t<TILES_LOWER, 0, 32, 32>(h, q, gx, cnt);
      }
      if (rc != FFGP_OK) return rc;
      if (hipGetLastError() != hipSuccess) return FFGP_ERR_HIP;
      if (tag) {
        h->syrk_launches += 1;
        h->syrk_flops += flops;
        if (timed) hipEventRecord(ev_stop, h->stream);
      }
    }
  }
  return FFGP_OK;
}

// ------------------------------------------------------------------------------------------------------------
// fp64 MFMA peak probe: register-resident v_mfma_f64_16x16x4_f64 stream, 4 waves per CU x 2 workgroups
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void ffgp_mfma_peak_kernel(double* out, int iters, double seed) {
  d4_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (d4_t){0.0, 0.0, 0.0, 0.0};
  double a = seed + threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" int ffgp_mfma_f64_peak(ffgp_handle* h, double* tflops_out) {
  if (!h || !tflops_out) return FFGP_ERR_ARG;
  const int blocks = 512, iters = 4096;
  FFGP_CHECK(ffgp_ensure_ws(h, (size_t)blocks * 256 * sizeof(double)));
  hipEvent_t e0, e1;
  FFGP_HIP(hipEventCreate(&e0));
  FFGP_HIP(hipEventCreate(&e1));
  hipLaunchKernelGGL(ffgp_mfma_peak_kernel, dim3(blocks), dim3(256), 0, h->stream, h->ws, 64, 0.5);  // warm-up
  FFGP_HIP(hipEventRecord(e0, h->stream));
  hipLaunchKernelGGL(ffgp_mfma_peak_kernel, dim3(blocks), dim3(256), 0, h->stream, h->ws, iters, 0.5);
  FFGP_HIP(hipEventRecord(e1, h->stream));
  FFGP_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  FFGP_HIP(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)blocks * 4.0 * iters * 8.0 * 2048.0;
  *tflops_out = flops / (ms * 1e-3) / 1e12;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return FFGP_OK;
}
