// Symmetric eigensolver, stage 2: band (bandwidth 32) -> tridiagonal by bulge chasing, and the back-transformation with the
// reflectors it produces.  (Second quarter of LAPACK's syevd underneath `torch.linalg.eigh(K)` in the HOGP block:
// FidelityFusion_Models/two_fidelity_models/hogp_simple.py:15-19,97-100.)  CPU restatement: the tests' numpy model eigh_twostage.py (sb2st, apply_q2).
//
// sb2st_chase: one 64-lane wavefront per sweep.  Sweep s annihilates column s below the sub-diagonal with a reflector of
//   length <= 32 and chases the bulge down the band in steps of 32 rows; step k touches the diagonal block D_k (two-sided) and
//   the block B_k below it (right-apply, new reflector from its first column, left-apply).  The 32 x 32 blocks live in
//   registers: lane = (row i, column half h), 16 columns each, so every product with the current reflector is lane-local and
//   the one product that is not (v'^T B) goes through a 32 x 32 LDS transpose.  Sweeps are pipelined two steps apart:
//   step k of sweep s needs steps k and k+1 of sweep s-1; a per-sweep progress counter in global memory carries that
//   (release store after the step's last band store, acquire load before the step's first band load, agent scope: the waves
//   of neighbouring sweeps sit on different XCDs/L2s).  Workgroup w runs sweeps w, w + G, ...: a sweep only ever waits for a
//   lower-numbered one, which is running or finished (workgroups are dispatched in order), every wait is bounded (watchdog
//   -> status word), so the grid always drains.
// q2_prep: the reflectors of 32 consecutive sweeps at the same step k form a 63 x 32 staircase V; its compact-WY T comes from
//   T^-1 = striu(V^T V) + diag(1 / tau); stored per block: V (64 x 32) and (V T)^T (32 x 64).
// q2_apply: Z <- Q2 Z.  One workgroup per slab of 32 columns of Z walks the blocks in the order (sweep group descending,
//   step ascending -- the only order in which overlapping blocks commute into place, see the numpy model) with a sliding 64-row
//   window in LDS: X = V^T Zw, Zw -= W X on the fp64 matrix cores.
#include "ffgp_internal.h"
#include "syevd_internal.h"

#define CH_DONE 0x3fffffff
#ifdef FFGP_CH_STAMPS   // development probe (tools/native/chase_phases.hip): 100 MHz clock stamps of one steady-state step, lane 0
__device__ unsigned long long ffgp_ch_stamp[16];
#define CH_STAMP(idx, waitmem, kk)                                                               \
  do {                                                                                           \
    if (s == FFGP_CH_STAMP_S && ((kk) == FFGP_CH_STAMP_K || (kk) == FFGP_CH_STAMP_K + 1)) {       \
      __builtin_amdgcn_sched_barrier(0);                                                         \
      if (waitmem) __builtin_amdgcn_s_waitcnt(0);                                                \
      if (lane == 0) ffgp_ch_stamp[((kk) - FFGP_CH_STAMP_K) * 8 + (idx)] = wall_clock64();        \
      __builtin_amdgcn_sched_barrier(0);                                                         \
    }                                                                                            \
  } while (0)
#else
#define CH_STAMP(idx, waitmem, kk)
#endif

template <int CTRL>
__device__ __forceinline__ double dpp_add(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return x + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rdlane(double x, int l) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}
// sum over the 32 lanes of a half-wave, returned in all of them: four DPP steps inside each row of 16 lanes (quad swaps, half
// mirror, mirror -- no LDS crossbar on the way), then the four row totals through scalar registers
__device__ __forceinline__ double wsum32(double x, int lane) { return qr_wsum32(x, lane); }   // (syevd_internal.h)
// x + (the value 32 lanes away), in every lane: both halves exchanged in registers by v_permlane32_swap (gfx950) -- through
// __shfl_xor it was an LDS crossbar round trip, three of them on every step's critical path (bit-identical: tools/native/wsum_check.hip)
__device__ __forceinline__ double halves_sum(double x) {
  const int lo = __double2loint(x), hi = __double2hiint(x);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

// the band lives in memory that waves on other compute units (and other XCDs) read and write while this kernel runs: every access
// is a device-scope access (sc1: served from the coherent level, never from this CU's L1); ordering against the progress counters
// is by waiting for the wave's outstanding memory operations (the workgroup-scope fence) around relaxed device-scope flag accesses
__device__ __forceinline__ double ldb(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stb(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// the same device-scope accesses as buffer instructions: (SGPR resource of the whole band) + (SGPR byte offset of the step's first
// column) + (one 32-bit per-lane byte offset).  `__hip_atomic_load(base + off)` compiles to flat-style global loads with a 64-bit
// address pair per entry -- two VALU adds per access and, for the block below, 32 more live registers: with the interior-step fast
// path the kernel needed 364 registers, one wave per SIMD, which made it a bad neighbour (config 5's eight blocks: 1.46 -> 1.58 s).
typedef int ch_v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double ldb_buf(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const ch_v2i v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 16);     // aux 16: sc1
  return __hiloint2double(v.y, v.x);
}
// XL ("XCD-local", round 5): every working wave of the launch sits on ONE XCD (the kernel checks HW_REG_XCC_ID and the others leave),
// so that XCD's L2 is the coherence point: band stores are PLAIN stores (written through, and the line STAYS in that L2 -- an sc1
// store drops it, after which even a same-XCD reader fetches from the memory side), band loads stay sc1 (they bypass this CU's L1 and
// are served by the L2 the producer just wrote).  Correct by construction: a wave on any other XCD never touches the band.
template <bool XL>
__device__ __forceinline__ void stb_buf(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double x) {
  ch_v2i v;
  v.x = __double2loint(x);
  v.y = __double2hiint(x);
  __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)voff, (int)soff, XL ? 0 : 16);
}
template <bool XL>
__device__ __forceinline__ void stb_x(__amdgpu_buffer_rsrc_t r, const double* AB, double* p, double v) {
  if (XL) stb_buf<true>(r, (unsigned)((size_t)(p - AB) * 8), 0u, v);
  else stb(p, v);
}

__device__ __forceinline__ int chase_wait(const int* p, int need, int* err) {
  int it = 0, v;
  while ((v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < need) {
    __builtin_amdgcn_s_sleep(1);
    // watchdog: ~0.3 s without progress, or any other sweep already gave up -> report and go on, so that the grid always drains
    if (++it > (1 << 19) || ((it & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
      __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v = need;
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return v;
}

__device__ __forceinline__ double ch_rcp(double d) {
  double y = __builtin_amdgcn_rcp(d);
  y = __builtin_fma(y, __builtin_fma(-d, y, 1.0), y);
  y = __builtin_fma(y, __builtin_fma(-d, y, 1.0), y);
  return y;
}

// Householder vector of x (lane i holds x_i, i < len valid, both halves hold the same values): returns v_i, sets tau and beta
__device__ __forceinline__ double house32(double x, int i, int lane, int len, double& tau, double& beta) {
  const double xv = (i < len) ? x : 0.0;
  const double sigma = wsum32((i >= 1) ? xv * xv : 0.0, lane);
  const double alpha = rdlane(xv, 0);
  tau = 0.0;
  beta = alpha;
  double scale = 0.0;
  if (sigma != 0.0) {
    const double q = __builtin_fma(alpha, alpha, sigma);
    double r = __builtin_amdgcn_rsq(q);                         // 1 / sqrt(q), two Newton steps
    r = __builtin_fma(0.5 * r, __builtin_fma(-q * r, r, 1.0), r);
    r = __builtin_fma(0.5 * r, __builtin_fma(-q * r, r, 1.0), r);
    double nrm = q * r;
    nrm = __builtin_fma(0.5 * r, __builtin_fma(-nrm, nrm, q), nrm);   // one step on the root itself
    beta = (alpha >= 0.0) ? -nrm : nrm;
    tau = (beta - alpha) * ch_rcp(beta);
    scale = ch_rcp(alpha - beta);
  }
  return (i == 0) ? 1.0 : xv * scale;
}

struct ChaseArgs {
  double* AB; int n;
  double* d; double* e;
  double* V2; double* tau2; int K;   // reflector (s, k): V2[(s K + k) 32 + i], tau2[s K + k]
  int* prog;                         // [n] steps completed per sweep (CH_DONE when the sweep has ended)
  int* err;
  int pack;                          // only workgroups with blockIdx % pack == 0 work (pack = 8: all of them on one XCD, one L2)
  int* ticket;                       // XL form: the next sweep to hand out (waves take their sweeps in the order they ask)
  int xcc;                           // XL form: the XCD whose waves work
  int s_begin, s_end;                // sweeps of this launch (the chase may be cut into several launches: prog carries over)
  const int* begin_from;             // chip-wide form behind an XL launch: the first sweep NOT yet handed out is read here (the XL ticket)
};

template <bool XL>
__global__ __launch_bounds__(64) void sb2st_chase(ChaseArgs p) {
  __shared__ double vsA[32], vsB[32], wsh[32], ush[32];
  __shared__ double Mt[32][33];
  int wg = 0, nwg = 1;
  if (XL) {
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((int)(xcc & 0xfu) != p.xcc) return;        // (placement is observed behaviour, not a contract: so it is CHECKED, wave by wave)
  } else {
    if (blockIdx.x % p.pack) return;
    wg = blockIdx.x / p.pack;
    nwg = (gridDim.x + p.pack - 1) / p.pack;
  }
  // The chip-wide launch that follows every XL launch: where a wave runs is observed behaviour, not a contract (partition modes, a
  // CU-masked stream, a part with fewer XCDs: possibly NO wave saw p.xcc and the ticket never moved), so whatever the XL launch did
  // not hand out is chased here.  Every sweep below the ticket was taken by a wave that ran it to its end (stream order: that launch
  // is complete), so the normal case is ticket >= s_end and every workgroup leaves at once.
  int s_first = p.s_begin;
  if (!XL && p.begin_from) {
    s_first = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p.begin_from, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    if (s_first < p.s_begin) s_first = p.s_begin;
    if (s_first >= p.s_end) return;
  }
  const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
  // XL: sweeps are handed out by a ticket counter -- a wave that holds sweep s knows every lower sweep has been taken by a wave that is
  // already running (or done), so it only ever waits for a running wave, however many waves the XCD received
  auto next_sweep = [&](int cur) -> int {
    if (!XL) return cur + nwg;
    int t = 0;
    if (threadIdx.x == 0) t = atomicAdd(p.ticket, 1);
    return __builtin_amdgcn_readfirstlane(t);
  };
  const int n = p.n;
  double* AB = p.AB;
  double* vs = vsA;
  double* v2s = vsB;
  // raw buffer over the whole band (stride 0, range check off the table: num_records = 2 GiB - 1; the band is n * SB_LDB doubles)
  const __amdgpu_buffer_rsrc_t band = __builtin_amdgcn_make_buffer_rsrc(AB, 0, 0x7fffffff, 0x00020000);
  for (int s = XL ? next_sweep(0) : s_first + wg; s < p.s_end; s = next_sweep(s)) {
    int c0 = s + 1;
    int len = min(32, n - c0);
    int seen = (s > 0) ? 0 : CH_DONE;    // progress of sweep s - 1 as last observed
    if (seen < 2) seen = chase_wait(p.prog + s - 1, 2, p.err);
    double tau, beta;
    {
      const double x = (i < len) ? ldb(AB + (size_t)s * SB_LDB + 1 + i) : 0.0;
      const double v = house32(x, i, lane, len, tau, beta);
      if (h == 0) {
        vs[i] = (i < len) ? v : 0.0;
        if (i < len) stb_x<XL>(band, AB, AB + (size_t)s * SB_LDB + 1 + i, (i == 0) ? beta : 0.0);
        p.V2[((size_t)s * p.K) * 32 + i] = (i < len) ? v : 0.0;
      }
      if (lane == 0) {
        p.e[s] = beta;
        p.d[s] = ldb(AB + (size_t)s * SB_LDB);
        p.tau2[(size_t)s * p.K] = tau;
      }
    }
    int k = 0;
    {
      // INTERIOR steps first (both blocks full: every step of a sweep but its last one or two), in their own loop: the per-lane byte
      // offsets below live only here, not across the general body further down (whose own ~240 registers they would add to)
      // per-lane byte offsets of the interior step's entries from AB + c0 * SB_LDB (lane (i, h) holds row i, columns cc = 16 h + q):
      // diagonal block entry (i, cc) lives at min(i, cc) * (SB_LDB - 1) + max(i, cc), the block below's (32 + i, cc) at cc * (SB_LDB - 1) + 32 + i
      unsigned offD[16], offB[16];
      int io = i;
      asm volatile("" : "+v"(io));      // (opaque: keeps the 32 offsets from being hoisted out of the sweep loop, where they would be live across the general body)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int cc = h * 16 + q;
        offD[q] = (unsigned)(min(io, cc) * (SB_LDB - 1) + max(io, cc)) * 8u;
        offB[q] = (unsigned)(cc * (SB_LDB - 1) + 32 + io) * 8u;
      }
      while (len == 32 && n - c0 >= 64) {
        const int r0 = c0 + 32;
        CH_STAMP(0, 0, k);    // the predecessor's counter has been seen
        {
        // ---- INTERIOR step (both blocks full: every step but the last one or two of a sweep).  Round 4: the general body below
          // spends ~85 % of its ~1500 instructions on per-entry bounds predicates (exec-mask juggling around every load and store) and
          // 64-bit address arithmetic; one wave issues an instruction every ~5 cycles, so the step was issue-bound at ~4 us.  Here
          // every entry exists: the loads and stores are unconditional with per-lane byte offsets computed once per kernel (scalar
          // base + 32-bit offset addressing), and the diagonal block's mirrored lanes compute bit-identical values (t1 + t2 with plain
          // multiplies and one add: the two products only swap places), so BOTH triangles store -- to the same address, the same bits.
          const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(c0) * (unsigned)(SB_LDB * 8);
          double D[16], B[16];
  #pragma unroll
          for (int q = 0; q < 16; ++q) D[q] = ldb_buf(band, offD[q], soff);
  #pragma unroll
          for (int q = 0; q < 16; ++q) B[q] = ldb_buf(band, offB[q], soff);
          CH_STAMP(1, 1, k);
          __syncthreads();   // vs complete
          double vq[16];
  #pragma unroll
          for (int q = 0; q < 16; ++q) vq[q] = vs[h * 16 + q];
          const double vi = vs[i];
          double pr = 0.0;
  #pragma unroll
          for (int q = 0; q < 16; ++q) pr = __builtin_fma(D[q], vq[q], pr);
          pr = halves_sum(pr);
          const double a2 = wsum32(vi * pr, lane);
          const double w = tau * pr - 0.5 * tau * tau * a2 * vi;
          if (h == 0) wsh[i] = w;
          __syncthreads();
  #pragma unroll
          for (int q = 0; q < 16; ++q) {
            const double t1 = __dmul_rn(vi, wsh[h * 16 + q]), t2 = __dmul_rn(w, vq[q]);      // (no contraction: see above)
            D[q] = __dsub_rn(D[q], __dadd_rn(t1, t2));
            stb_buf<XL>(band, offD[q], soff, D[q]);
          }
          CH_STAMP(2, 0, k);
          double sb = 0.0;
  #pragma unroll
          for (int q = 0; q < 16; ++q) sb = __builtin_fma(B[q], vq[q], sb);
          sb = halves_sum(sb);
          const double ts = tau * sb;
  #pragma unroll
          for (int q = 0; q < 16; ++q) B[q] = __builtin_fma(-ts, vq[q], B[q]);
          double tau_n, beta_n;
          const double x0 = __shfl(B[0], i);           // first column: lane i of half 0
          const double v2i = house32(x0, i, lane, 32, tau_n, beta_n);
          if (h == 0) {
            B[0] = (i == 0) ? beta_n : 0.0;
            v2s[i] = v2i;
          }
  #pragma unroll
          for (int q = 0; q < 16; ++q) Mt[i][h * 16 + q] = (h * 16 + q == 0) ? 0.0 : v2i * B[q];
          __syncthreads();
          {
            double u = 0.0;
  #pragma unroll
            for (int r = 0; r < 16; ++r) u += Mt[h * 16 + r][i];
            u = halves_sum(u);
            if (h == 0) ush[i] = u;
          }
          __syncthreads();
          const double tv = tau_n * v2i;
  #pragma unroll
          for (int q = 0; q < 16; ++q) {
            B[q] = __builtin_fma(-tv, ush[h * 16 + q], B[q]);
            stb_buf<XL>(band, offB[q], soff, B[q]);
          }
          CH_STAMP(3, 0, k);
          ++k;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the publish protocol of the general body, below)
          CH_STAMP(4, 0, k - 1);
          if (lane == 0) __hip_atomic_store(p.prog + s, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (h == 0) p.V2[((size_t)s * p.K + k) * 32 + i] = v2i;
          if (lane == 0) p.tau2[(size_t)s * p.K + k] = tau_n;
          c0 = r0;
          tau = tau_n;
          double* t_ = vs;
          vs = v2s;
          v2s = t_;
          CH_STAMP(5, 0, k - 1);
        if (seen < k + 2) seen = chase_wait(p.prog + s - 1, k + 2, p.err);
        }
      }
    }
    while (true) {
      // (the general body works from opaque copies of the lane coordinates: everything it derives from them -- dozens of per-entry
      //  predicates and addresses -- is then recomputed per step instead of being hoisted in front of the interior loop, where it
      //  would sit in registers across every interior step; these steps are the last one or two of a sweep)
      int i_ = i, h_ = h, lane_ = lane;
      asm volatile("" : "+v"(i_), "+v"(h_), "+v"(lane_));
      const int i = i_, h = h_, lane = lane_;
      const int r0 = c0 + len;
      const bool more = (r0 <= n - 1);
      const int nrow = more ? min(32, n - r0) : 0;
      CH_STAMP(0, 0, k);    // the predecessor's counter has been seen
      // ---- both blocks of the step are requested up front
      double D[16], B[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int cc = h * 16 + q;
        const bool ok = (i < len) && (cc < len);
        const size_t a = (i >= cc) ? ((size_t)(c0 + cc) * SB_LDB + (i - cc)) : ((size_t)(c0 + i) * SB_LDB + (cc - i));
        D[q] = ok ? ldb(AB + a) : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int cc = h * 16 + q;
        const bool ok = (i < nrow) && (cc < len);
        B[q] = ok ? ldb(AB + (size_t)(c0 + cc) * SB_LDB + (len + i - cc)) : 0.0;
      }
      CH_STAMP(1, 1, k);    // both blocks have arrived
      __syncthreads();   // vs complete
      // ---- diagonal block, two-sided
      double vq[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) vq[q] = vs[h * 16 + q];
      const double vi = vs[i];
      double pr = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) pr = __builtin_fma(D[q], vq[q], pr);
      pr = halves_sum(pr);
      const double a2 = wsum32(vi * pr, lane);
      const double w = tau * pr - 0.5 * tau * tau * a2 * vi;
      if (h == 0) wsh[i] = w;
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int cc = h * 16 + q;
        D[q] -= vi * wsh[cc] + w * vq[q];
        if (i >= cc && i < len) stb_x<XL>(band, AB, AB + (size_t)(c0 + cc) * SB_LDB + (i - cc), D[q]);
      }
      CH_STAMP(2, 0, k);    // diagonal block updated, its stores issued
      if (!more) break;
      // ---- block below: right-apply, new reflector, left-apply
      double sb = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) sb = __builtin_fma(B[q], vq[q], sb);
      sb = halves_sum(sb);
      const double ts = tau * sb;
#pragma unroll
      for (int q = 0; q < 16; ++q) B[q] = __builtin_fma(-ts, vq[q], B[q]);
      double tau_n, beta_n;
      const double x0 = __shfl(B[0], i);           // first column: lane i of half 0
      const double v2 = house32(x0, i, lane, nrow, tau_n, beta_n);
      const double v2i = (i < nrow) ? v2 : 0.0;
      if (h == 0) {
        B[0] = (i == 0) ? beta_n : 0.0;
        v2s[i] = v2i;
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int cc = h * 16 + q;
        Mt[i][cc] = (cc == 0) ? 0.0 : v2i * B[q];
      }
      __syncthreads();
      {
        double u = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) u += Mt[h * 16 + r][i];
        u = halves_sum(u);
        if (h == 0) ush[i] = u;
      }
      __syncthreads();
      const double tv = tau_n * v2i;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int cc = h * 16 + q;
        B[q] = __builtin_fma(-tv, ush[cc], B[q]);
        if (i < nrow && cc < len) stb_x<XL>(band, AB, AB + (size_t)(c0 + cc) * SB_LDB + (len + i - cc), B[q]);
      }
      CH_STAMP(3, 0, k);    // block below updated, its stores issued
      ++k;
      // publish: k steps of this sweep are complete.  Every band store above is an sc1 (write-through) store of THIS wave; the
      // explicit wait drains them to the coherent level before the counter moves (a workgroup-scope fence alone compiles to
      // lgkmcnt(0) only: the counter could overtake the band).  tools/check_isa.py asserts the vmcnt(0) in front of both stores.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      CH_STAMP(4, 0, k - 1);    // every band store of the step has completed
      if (lane == 0) __hip_atomic_store(p.prog + s, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (h == 0) p.V2[((size_t)s * p.K + k) * 32 + i] = v2i;
      if (lane == 0) p.tau2[(size_t)s * p.K + k] = tau_n;
      c0 = r0;
      len = nrow;
      tau = tau_n;
      double* t_ = vs;
      vs = v2s;
      v2s = t_;
      CH_STAMP(5, 0, k - 1);    // counter published, reflector stored
      if (seen < k + 2) seen = chase_wait(p.prog + s - 1, k + 2, p.err);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(p.prog + s, CH_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }
}

__global__ void sb2st_ticket_init(int* ticket, int v) { ticket[0] = v; }

__global__ void sb2st_tail(const double* __restrict__ AB, int n, double* d, double* e) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    d[n - 2] = AB[(size_t)(n - 2) * SB_LDB];
    e[n - 2] = AB[(size_t)(n - 2) * SB_LDB + 1];
    d[n - 1] = AB[(size_t)(n - 1) * SB_LDB];
    e[n - 1] = 0.0;
  }
}

static inline int chase_K(int n) { return n / 32 + 1; }

// AB [n, 64] band in (destroyed), d [n], e [n] out, V2 [n * K * 32], tau2 [n * K], prog [n + 1] ints (last: status word).
// init clears the stores; chunk runs the sweeps [s_begin, s_end) (sweeps of later launches find their predecessors' counters at
// CH_DONE); finish writes the last two diagonal entries.  All on stream st.
int ffgp_sb2st_init(ffgp_handle* h, hipStream_t st, int n, double* V2, double* tau2, int* prog) {
  if (n < 64 || n % 32) return FFGP_ERR_ARG;
  const int K = chase_K(n);
  FFGP_HIP(hipMemsetAsync(prog, 0, (size_t)(n + 1) * sizeof(int), st));
  FFGP_HIP(hipMemsetAsync(V2, 0, (size_t)n * K * 32 * sizeof(double), st));
  FFGP_HIP(hipMemsetAsync(tau2, 0, (size_t)n * K * sizeof(double), st));
  return FFGP_OK;
}

int ffgp_sb2st_chunk(ffgp_handle* h, hipStream_t st, double* AB, int n, double* d, double* e, double* V2, double* tau2, int* prog, int s_begin,
                     int s_end) {
  s_end = min(s_end, n - 2);
  if (s_begin >= s_end) return FFGP_OK;
  ChaseArgs a;
  a.AB = AB; a.n = n; a.d = d; a.e = e; a.V2 = V2; a.tau2 = tau2; a.K = chase_K(n); a.prog = prog; a.err = prog + n;
  a.s_begin = s_begin; a.s_end = s_end;
  // placement of the working wavefronts: every pack-th workgroup works (workgroups are dealt round-robin to the 8 XCDs, so pack = 8
  // puts all of them on one XCD and one L2, pack = 1 one wave on every CU of the chip).  Measured at N = 8192 (tools/chase_dbg.py):
  // pack 8: 166 ms, 4: 92, 2: 68-69, 1: 64-65 (round 3, with a kernel of 238 registers: 119 / 88 / 80 / 82) -- the waves get in each
  // other's way on a shared SIMD more than the memory-side hand-over between XCDs costs.  Purely a placement: every band access is
  // a device-scope access wherever the wave runs.  Default 1; beside other blocks' kernels (config 5's eight HOGP blocks from four
  // host threads) 1 and 2 measure the same, 1.43-1.47 s per step.
  a.pack = h->chase_pack > 0 ? h->chase_pack : 1;
  a.ticket = nullptr;
  a.begin_from = nullptr;
  a.xcc = h->chase_xcc;
  // XCD-local form (option chase_xl, default 1; see stb_buf).  A sweep follows its predecessor two steps behind, so (n / 32) / 2 sweeps
  // are in flight at most: up to n = 2048 that is <= 32 working waves, ONE per CU of one XCD, and the hand-over through that XCD's L2
  // shortens the step -- sb2st 7.9 -> 6.5 ms at n = 1024, 15.9 -> 13.8 at n = 2048 (same eigenvalues and vectors).  Above that the
  // waves of one XCD share CUs, which costs more than the memory-side hand-over saves (n = 4096: 32.2 -> 41.9 ms, n = 8192: 64.7 ->
  // 132.8 with 128 waves on 32 CUs; pack = 8 above is the same lesson), so larger bands keep the chip-wide form.
  if (h->chase_xl && n <= h->chase_xl_max_n) {
    a.ticket = h->d_info + 12;
    hipLaunchKernelGGL(sb2st_ticket_init, dim3(1), dim3(1), 0, st, a.ticket, s_begin);
    const int grid = min(s_end - s_begin, max(16, n / 64)) * 8;
    hipLaunchKernelGGL(sb2st_chase<true>, dim3(grid), dim3(64), 0, st, a);
    // ... and the sweeps the ticket did not reach (none, unless no wave landed on XCD `xcc`) on the chip-wide form: the chase is
    // complete when this returns, wherever the runtime placed the XL launch's waves
    a.begin_from = a.ticket;
    a.ticket = nullptr;
    hipLaunchKernelGGL(sb2st_chase<false>, dim3(min(s_end - s_begin, 256) * a.pack), dim3(64), 0, st, a);
    return hipGetLastError() == hipSuccess ? FFGP_OK : FFGP_ERR_HIP;
  }
  const int grid = min(s_end - s_begin, 256) * a.pack;
  hipLaunchKernelGGL(sb2st_chase<false>, dim3(grid), dim3(64), 0, st, a);
  return hipGetLastError() == hipSuccess ? FFGP_OK : FFGP_ERR_HIP;
}

int ffgp_sb2st_finish(ffgp_handle* h, hipStream_t st, const double* AB, int n, double* d, double* e) {
  hipLaunchKernelGGL(sb2st_tail, dim3(1), dim3(64), 0, st, AB, n, d, e);
  return hipGetLastError() == hipSuccess ? FFGP_OK : FFGP_ERR_HIP;
}

int ffgp_sb2st_impl(ffgp_handle* h, double* AB, int n, double* d, double* e, double* V2, double* tau2, int* prog) {
  FFGP_CHECK(ffgp_sb2st_init(h, h->stream, n, V2, tau2, prog));
  FFGP_CHECK(ffgp_sb2st_chunk(h, h->stream, AB, n, d, e, V2, tau2, prog, 0, n - 2));
  return ffgp_sb2st_finish(h, h->stream, AB, n, d, e);
}

// ---------------------------------------------------------------------------------------------------------------------
// back-transformation with the chase's reflectors
// ---------------------------------------------------------------------------------------------------------------------
__host__ __device__ static inline int q2_nsteps(int n, int s) { return (s <= n - 3) ? (n - 2 - s) / 32 + 1 : 0; }

size_t ffgp_q2_block_doubles(int n) { return (size_t)(n / 32) * chase_K(n) * 4096; }

struct PrepArgs {
  const double* V2; const double* tau2; int n, K;
  double* blocks;
  int G0;      // first sweep group of this launch (blockIdx.y counts from it)
  int trans;   // 1: W = V T^T (the block of Q2^T), 0: W = V T
  int lanes;   // 1: both factors in the MFMA lane order of q2_apply_wave4 (pairs of k-steps per lane: 16-byte operand loads)
};

__global__ __launch_bounds__(256) void q2_prep(PrepArgs p) {
  __shared__ double Vs[64][33];
  __shared__ double Gm[32][33];
  __shared__ double Tm[32][33];
  __shared__ double dinv[32];
  const int tid = threadIdx.x;
  const int G = p.G0 + blockIdx.y, k = blockIdx.x;
  const int n = p.n;
  const int s0 = 32 * G;
  if (s0 > n - 3 || k >= (n - 2 - s0) / 32 + 1) return;
  for (int idx = tid; idx < 64 * 32; idx += 256) Vs[idx >> 5][idx & 31] = 0.0;
  __syncthreads();
  // member t: sweep s0 + t, rows t .. t + len - 1 of the block
  for (int idx = tid; idx < 32 * 32; idx += 256) {
    const int t = idx >> 5, i = idx & 31;
    const int s = s0 + t;
    bool live = (s <= n - 3) && (k < (n - 2 - s) / 32 + 1);
    double tau = live ? p.tau2[(size_t)s * p.K + k] : 0.0;
    if (tau == 0.0) live = false;
    if (live) Vs[t + i][t] = p.V2[((size_t)s * p.K + k) * 32 + i];
    if (i == 0) dinv[t] = live ? 1.0 / tau : 1.0;
  }
  __syncthreads();
  {
    const int a = tid >> 3, b0 = (tid & 7) * 4;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8   // (fully unrolled, the 320 LDS loads of this loop were hoisted into 512 registers + 124 spilled ones: one wave per SIMD)
    for (int r = 0; r < 64; ++r) {
      const double av = Vs[r][a];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_fma(av, Vs[r][b0 + q], acc[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int b = b0 + q;
      Gm[a][b] = (b > a) ? acc[q] : ((b == a) ? dinv[a] : 0.0);   // T^-1 = striu(V^T V) + diag(1 / tau)
    }
  }
  __syncthreads();
  {   // T = (T^-1)^-1 by back substitution.  Its columns are independent recurrences: half-wave hw takes the columns hw, hw + 8, hw + 16,
      // hw + 24 with entry cc of a column in lane cc -- per step one product per lane and a half-wave sum (syevd_internal.h), the four
      // columns inside the step loop so that their sums overlap.  (As a loop of 32 threads with a serial inner sum this was the longest
      // phase of the kernel, the other 224 threads waiting at the barrier.)
    const int lane = tid & 63, i = tid & 31, hw = tid >> 5;
    const double dii = Gm[i][i];
    double q[4] = {0.0, 0.0, 0.0, 0.0};
    for (int ii = 31; ii >= 0; --ii) {
      const double gv = Gm[ii][i];          // row ii of T^-1 (zero left of the diagonal)
      double sm[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int j = hw + 8 * c;
        sm[c] = qr_wsum32((i > ii && i <= j) ? gv * q[c] : 0.0, lane);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int j = hw + 8 * c;
        if (i == ii) q[c] = (ii > j) ? 0.0 : (((ii == j) ? 1.0 : 0.0) - sm[c]) / dii;
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) Tm[i][hw + 8 * c] = q[c];
  }
  __syncthreads();
  double* out = p.blocks + ((size_t)G * p.K + k) * 4096;
  // W = V T (or V T^T): a thread's column j is the same in all of its eight rows, so that column (row) of T lives in registers, and
  // because T is upper triangular its exact zeros do the masking: every sum is a fixed 32 terms.  (With variable-length inner loops the
  // compiler unrolled all eight rows into 256 VGPRs + 500 bytes of scratch per lane, one wave per SIMD.)
  const int j = tid & 31;
  double tcol[32];
#pragma unroll
  for (int cc = 0; cc < 32; ++cc) tcol[cc] = p.trans ? Tm[j][cc] : Tm[cc][j];
#pragma unroll 1
  for (int it = 0; it < 8; ++it) {
    const int r = (tid >> 5) + 8 * it, idx = 32 * r + j;
    // lane order: A-operand element (row m = 16 t + lr, k = 4 kq + lq) of tile t sits at ((t * nkp + kq / 2) * 64 + 16 lq + lr) * 2 + kq % 2
    if (p.lanes) out[(((j >> 4) * 8 + (r >> 3)) * 64 + 16 * (r & 3) + (j & 15)) * 2 + ((r >> 2) & 1)] = Vs[r][j];   // V^T: m = reflector j, k = row r
    else out[idx] = Vs[r][j];
    double s = 0.0;
#pragma unroll
    for (int cc = 0; cc < 32; ++cc) s = __builtin_fma(Vs[r][cc], tcol[cc], s);
    if (p.lanes) out[2048 + (((r >> 4) * 4 + (j >> 3)) * 64 + 16 * (j & 3) + (r & 15)) * 2 + ((j >> 2) & 1)] = s;   // W: m = row r, k = reflector j
    else out[2048 + j * 64 + r] = s;     // W = V T, stored transposed ([32][64]): the apply kernel's lanes read 16 consecutive rows
  }
}

int ffgp_q2_prep_impl(ffgp_handle* h, const double* V2, const double* tau2, int n, double* blocks, int G0, int G1, int trans) {
  if (G1 <= G0) return FFGP_OK;
  PrepArgs a;
  a.V2 = V2; a.tau2 = tau2; a.n = n; a.K = chase_K(n); a.blocks = blocks; a.G0 = G0; a.trans = trans;
  a.lanes = (!trans && h->q2_wave4) ? 1 : 0;
  h->q2_blocks_lanes = a.lanes;     // (the apply launch must read the blocks the way they were written)
  hipLaunchKernelGGL(q2_prep, dim3(a.K, G1 - G0), dim3(256), 0, h->stream, a);
  return hipGetLastError() == hipSuccess ? FFGP_OK : FFGP_ERR_HIP;
}

struct ApplyArgs {
  const double* blocks; int n, K;
  double* Z; int ldz; int ncols; int dbg;
  int G0, G1;   // sweep groups of this launch
  int skip8;    // 1: workgroups with blockIdx % 8 == 0 (the XCD the chase runs on) leave at once, the others share the slabs
};

// One workgroup of 4 waves per slab of 16 columns of Z; several workgroups share a CU (16 KB of LDS, <= 128 VGPRs), so one
// slab's barriers and memory latencies are filled with another slab's arithmetic.  The 64-row window lives in LDS as two 32-row
// halves whose roles swap from block to block (no copying when the window slides); per block X = V^T Zw (32 x 16, k = 64: wave
// (xa, xk) one 16 x 16 tile over half of k, the two halves summed when X is read) and Zw -= W X (64 x 16: one tile per wave),
// 16 MFMAs per wave.  V and W^T come straight from global memory (L2) in the MFMA lane layout, one block ahead -- every wave
// loads a different quarter of them; the 32 new rows of the window are requested before the block's arithmetic and land in LDS
// after it.
#define XLD 16   // 4 consecutive rows of 16 doubles: the 64 lanes of an operand read cover 64 consecutive doubles, no conflicts
// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global load of the wave
// (loads and stores share one counter on this ISA), which would serialise the operand / window prefetches with the arithmetic
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// FWD = false: Z <- Q2 Z (groups descending, steps ascending, window slides down);  FWD = true: Z <- Q2^T Z with blocks prepared as
// V, (V T^T)^T (groups ascending -- the order the chase produces them --, steps descending, window slides up)
template <bool FWD>
__global__ __launch_bounds__(256, 4) void q2_apply(ApplyArgs p) {
  __shared__ double Zs[2][32 * XLD];      // physical halves of the window
  __shared__ double Xs[2][32 * XLD];      // the two k-halves of X
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = p.n;
  int slab = blockIdx.x;
  if (p.skip8) {
    if ((blockIdx.x & 7) == 0) return;
    slab = blockIdx.x - (blockIdx.x >> 3) - 1;
  }
  const int col0 = slab * 16;
  if (col0 >= p.ncols) return;
  double* __restrict__ Zg = p.Z + col0;
  const int lr = lane & 15, lq = lane >> 4;
  const int trow = tid >> 3, tc2 = (tid & 7) * 2;   // window <-> global map: 256 threads, 32 rows x 8 column pairs
  const int ncv = min(16, p.ncols - col0);
  const bool cok0 = tc2 < ncv, cok1 = tc2 + 1 < ncv;
  // X tile of this wave: rows 16 xa.., k half xk;   Zw tile: logical rows 16 zr.. (half zr >> 1)
  const int xa = wave & 1, xk = wave >> 1;
  const int zr = wave;
  auto load_ops = [&](const double* __restrict__ blk, double (&va)[8], double (&wa)[8]) {
#pragma unroll
    for (int kq = 0; kq < 8; ++kq) va[kq] = blk[(32 * xk + kq * 4 + lq) * 32 + xa * 16 + lr];
#pragma unroll
    for (int kq = 0; kq < 8; ++kq) wa[kq] = blk[2048 + (kq * 4 + lq) * 64 + zr * 16 + lr];
  };
  auto load_rows = [&](int grow) {   // two doubles of one row of the slab (zero beyond the matrix / the live columns)
    d2_t v = {0.0, 0.0};
    if (grow < n) {
      const double* src = Zg + (size_t)grow * p.ldz + tc2;
      if (cok1) v = *reinterpret_cast<const d2_t*>(src);
      else if (cok0) v.x = src[0];
    }
    return v;
  };
  auto store_rows = [&](int grow, d2_t v) {
    if (grow < n) {
      double* dst = Zg + (size_t)grow * p.ldz + tc2;
      if (cok1) *reinterpret_cast<d2_t*>(dst) = v;
      else if (cok0) dst[0] = v.x;
    }
  };
  for (int gi = 0; gi < p.G1 - p.G0; ++gi) {
    const int G = FWD ? p.G0 + gi : p.G1 - 1 - gi;
    const int s0 = 32 * G;
    const int nk = q2_nsteps(n, s0);
    if (nk == 0) continue;
    const int kfirst = FWD ? nk - 1 : 0, kstep = FWD ? -1 : 1;
    double va[8], wa[8];
    load_ops(p.blocks + ((size_t)G * p.K + kfirst) * 4096, va, wa);
    int cur = 0;   // physical half that holds the window's rows 0..31
    {
      const int rb = s0 + 1 + 32 * kfirst;
      __syncthreads();   // the previous group's last reads of Zs are done, its last stores to Z visible to the whole workgroup
      const d2_t a0 = load_rows(rb + trow), a1 = load_rows(rb + 32 + trow);
      *reinterpret_cast<d2_t*>(&Zs[0][trow * XLD + tc2]) = a0;
      *reinterpret_cast<d2_t*>(&Zs[1][trow * XLD + tc2]) = a1;
    }
    for (int ki = 0; ki < nk; ++ki) {
      const int k = kfirst + kstep * ki;
      const int rb = s0 + 1 + 32 * k;
      const bool last = (ki == nk - 1);
      double vn[8], wn[8];
      d2_t znew = {0.0, 0.0};
      if (!last) {
        load_ops(p.blocks + ((size_t)G * p.K + k + kstep) * 4096, vn, wn);
        // the rows that enter the window at the next step: below it when it slides down, above it when it slides up
        znew = load_rows(FWD ? rb - 32 + trow : rb + 64 + trow);
      }
      lds_barrier();   // window complete
      {
        const double* zh = Zs[xk ^ cur];
        double zb[8];
#pragma unroll
        for (int kq = 0; kq < 8; ++kq) zb[kq] = zh[(kq * 4 + lq) * XLD + lr];   // all operand reads first, then the MFMA chain
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kq = 0; kq < 8; ++kq) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[kq], zb[kq], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) Xs[xk][(xa * 16 + 4 * r + lq) * XLD + lr] = acc[r];
      }
      lds_barrier();
      {
        double* zh = Zs[(zr >> 1) ^ cur] + ((zr & 1) * 16) * XLD;
        d4_t acc;
        double xb[8];
#pragma unroll
        for (int kq = 0; kq < 8; ++kq) {
          const int o = (kq * 4 + lq) * XLD + lr;
          xb[kq] = Xs[0][o] + Xs[1][o];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = zh[(4 * r + lq) * XLD + lr];
#pragma unroll
        for (int kq = 0; kq < 8; ++kq) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[kq], xb[kq], acc, 0, 0, 1);   // -A
#pragma unroll
        for (int r = 0; r < 4; ++r) zh[(4 * r + lq) * XLD + lr] = acc[r];
      }
      lds_barrier();
      // the half the window leaves behind (its first 32 rows when it slides down, its last 32 when it slides up) is final for
      // this group; that half receives the incoming rows (the last step writes both halves).  The LDS write of the incoming rows
      // comes BEFORE the global stores: waiting for the incoming rows' load must not also wait for stores issued a moment ago
      // (loads and stores share one counter).
      const int hout = FWD ? (cur ^ 1) : cur;          // physical half that leaves
      const int rout = FWD ? rb + 32 : rb;
      const d2_t fin = *reinterpret_cast<const d2_t*>(&Zs[hout][trow * XLD + tc2]);
      if (last) {
        const d2_t fin2 = *reinterpret_cast<const d2_t*>(&Zs[hout ^ 1][trow * XLD + tc2]);
        store_rows(rout + trow, fin);
        store_rows((FWD ? rb : rb + 32) + trow, fin2);
      } else {
        *reinterpret_cast<d2_t*>(&Zs[hout][trow * XLD + tc2]) = znew;
        store_rows(rout + trow, fin);
        cur ^= 1;
#pragma unroll
        for (int q = 0; q < 8; ++q) va[q] = vn[q], wa[q] = wn[q];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Z <- Q2 Z, four sweep groups per pass over Z (round 3, second half).  The kernel above streams the whole slab through its
// window once per sweep group: n / 32 passes over Z, 8 flops per byte of HBM traffic -- at N = 8192 it ran at 14 TFLOP/s, bound
// by that traffic and by three barriers per 63 x 32 block.  Blocks of different groups commute when their rows are disjoint,
// and block (G, k) covers the row bands G + k and G + k + 1 (32 rows each), so four consecutive groups can travel down the slab
// TOGETHER as a wavefront: wave w of the workgroup owns group Gtop - w and runs two steps behind wave w - 1, which puts it three
// bands higher -- never on the same rows -- and satisfies every ordering constraint of the sequential order (group G + 1's steps
// k - 2 .. k precede step k of group G).  The window is a ring of 12 bands in LDS (the 11 live ones + the one arriving); per time
// step every wave applies one whole block to its two bands (X = V^T Zw and Zw -= W X, 64 MFMAs, X turned around through a
// wave-private LDS tile), then one band leaves for global memory and one arrives: two barriers per FOUR blocks, a quarter of the
// HBM traffic.
// ---------------------------------------------------------------------------------------------------------------------
#ifdef FFGP_Q2_STAMPS   // development probe (tools/native/q2_phases.hip): shader-clock stamps of two consecutive time steps, wave FFGP_Q2_STAMP_W
__device__ unsigned long long ffgp_q2_stamp[32];
#define Q2_STAMP(idx)                                                                                              \
  do {                                                                                                             \
    if (blockIdx.x == FFGP_Q2_STAMP_B && Gtop == p.G1 - 1 - 4 * FFGP_Q2_STAMP_PASS && (t == FFGP_Q2_STAMP_T || t == FFGP_Q2_STAMP_T + 1) && \
        wave == FFGP_Q2_STAMP_W) {                                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
      if (lane == 0) ffgp_q2_stamp[(t - FFGP_Q2_STAMP_T) * 16 + (idx)] = __builtin_readcyclecounter();             \
      __builtin_amdgcn_sched_barrier(0);                                                                           \
    }                                                                                                              \
  } while (0)
#else
#define Q2_STAMP(idx)
#endif
#define Q2W_RING 12
#define Q2W_LDS_DOUBLES(NC) ((NC) * (Q2W_RING * 32 * XLD + 4 * 32 * XLD))

// Round 4, measured with the stamps above (tools/native/q2_phases.hip; one time step of one wave at N = 8192, two workgroups per CU):
// X = V^T Z 1.3 us, X through LDS 0.5-0.7, Z -= W X 1.0, the two barriers 0.1-0.2 each, band out / in 0.3-1.0 -- 3.5-4.2 us per
// step where the two waves of a SIMD need 2.8 us of MFMA issue between them.  Built on that and measured: X handed over in registers
// (the first product's accumulator layout IS the second product's B-operand layout, no LDS trip), the arriving band stored before the
// operand loads are issued (the in-order counter made it wait for them), V and W requested a whole step ahead in a second register
// set.  Each shortens the wave's own path and none moves the kernel (46.7-49.4 ms against 46.7; the 32-column form improves 57 -> 48.5
// because there a SIMD holds one wave): with 2 x 4 waves per CU pulling 26 KB of V / W per block the kernel sits on the L2's
// bandwidth (~13 TB/s over the chip), as the note on NC below says.  Taken out again.
// NC = 1: slabs of 16 columns, 64 KB of LDS, two workgroups per CU.  NC = 2: slabs of 32 columns (two 16-column tiles, stored one
// after the other so that operand reads stay conflict-free), 128 KB, one workgroup per CU -- every block's V and W^T are fetched
// once per 32 columns instead of once per 16: the operand stream (n^3 / 6 doubles per 16 columns of Z) is what bounds NC = 1.
// SPLIT (round 5; NC = 2 only): eight waves -- waves 0-3 own the slab's first 16-column tile, waves 4-7 the second -- so the 32-column slab's
// LDS (one workgroup per CU) no longer means one wave per SIMD: wave w and wave w + 4 apply the SAME block in the same time step to different
// columns and ask for the same V / W^T at the same moment (one L2 request between them when the second hits the first's line in flight).
template <int NC, bool SPLIT = false>
__global__ __launch_bounds__(SPLIT ? 512 : 256, (NC == 1 && !SPLIT) ? 2 : 1) void q2_apply_wave4(ApplyArgs p) {
  static_assert(!SPLIT || NC == 2, "the split form is the 32-column slab on eight waves");
  extern __shared__ double q2w_sm[];
  constexpr int TILE = 32 * XLD;                            // one band of one column tile
  double* ring = q2w_sm;                                    // [Q2W_RING][NC][32][XLD]
  const int tid = threadIdx.x, lane = tid & 63, wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = SPLIT ? (wave_all & 3) : wave_all;       // the wave's place in the wavefront of four sweep groups
  constexpr int NM = SPLIT ? 1 : NC;                        // column tiles a wave applies its block to ...
  const int c0 = SPLIT ? (wave_all >> 2) : 0;               // ... starting at this one
  constexpr int NBT = SPLIT ? 1 : NC;                       // column tiles a thread moves per band row ...
  const int bt0 = SPLIT ? ((tid >> 3) & 1) : 0;             // ... starting at this one
  double* xs = q2w_sm + Q2W_RING * NC * TILE + wave_all * NM * TILE;   // this wave's X tiles [NM][32][XLD]
  const int n = p.n;
  const int col0 = blockIdx.x * 16 * NC;
  if (col0 >= p.ncols) return;
  double* __restrict__ Zg = p.Z + col0;
  const int lr = lane & 15, lq = lane >> 4;
  const int trow = SPLIT ? (tid >> 4) : (tid >> 3), tc2 = (tid & 7) * 2;   // band <-> global map: 32 rows x 8 column pairs per tile
  bool cok0[NBT], cok1[NBT];
#pragma unroll
  for (int ct = 0; ct < NBT; ++ct) {
    cok0[ct] = col0 + 16 * (bt0 + ct) + tc2 < p.ncols;
    cok1[ct] = col0 + 16 * (bt0 + ct) + tc2 + 1 < p.ncols;
  }
  auto load_band = [&](int b, d2_t (&v)[NBT]) {             // two doubles per tile of one row of band b (rows 32 b + 1 .. 32 b + 32)
    const int grow = 32 * b + 1 + trow;
#pragma unroll
    for (int ct = 0; ct < NBT; ++ct) {
      v[ct] = d2_t{0.0, 0.0};
      if (b >= 0 && grow < n) {
        const double* src = Zg + (size_t)grow * p.ldz + 16 * (bt0 + ct) + tc2;
        if (cok1[ct]) v[ct] = *reinterpret_cast<const d2_t*>(src);
        else if (cok0[ct]) v[ct].x = src[0];
      }
    }
  };
  auto store_band = [&](int b, const d2_t (&v)[NBT]) {
    const int grow = 32 * b + 1 + trow;
    if (b < 0 || grow >= n) return;
#pragma unroll
    for (int ct = 0; ct < NBT; ++ct) {
      double* dst = Zg + (size_t)grow * p.ldz + 16 * (bt0 + ct) + tc2;
      if (cok1[ct]) *reinterpret_cast<d2_t*>(dst) = v[ct];
      else if (cok0[ct]) dst[0] = v[ct].x;
    }
  };
  auto slot = [&](int b) { return ring + (((b % Q2W_RING) + Q2W_RING) % Q2W_RING) * NC * TILE; };
  auto put_band = [&](int b, const d2_t (&v)[NBT]) {
#pragma unroll
    for (int ct = 0; ct < NBT; ++ct) *reinterpret_cast<d2_t*>(slot(b) + (bt0 + ct) * TILE + trow * XLD + tc2) = v[ct];
  };
  auto get_band = [&](int b, d2_t (&v)[NBT]) {
#pragma unroll
    for (int ct = 0; ct < NBT; ++ct) v[ct] = *reinterpret_cast<const d2_t*>(slot(b) + (bt0 + ct) * TILE + trow * XLD + tc2);
  };

  for (int Gtop = p.G1 - 1; Gtop >= p.G0; Gtop -= 4) {
    const int Gw = Gtop - wave;
    const int nkw = (Gw >= p.G0) ? q2_nsteps(n, 32 * Gw) : 0;
    int T = 0;
    for (int w = 0; w < 4; ++w) {
      const int g = Gtop - w;
      const int nk = (g >= p.G0) ? q2_nsteps(n, 32 * g) : 0;
      if (nk > 0) T = max(T, nk + 2 * w);
    }
    if (T == 0) continue;
    __syncthreads();       // the previous pass's stores are visible to the whole workgroup, its LDS reads are done
    // the window at t = 0: bands Gtop - 9 .. Gtop + 1
    for (int b = Gtop - 9; b <= Gtop + 1; ++b) {
      d2_t v[NBT];
      load_band(b, v);
      put_band(b, v);
    }
    // this wave's operands of its first block: V (64 x 32) as 2 x 16 A-operands, W^T (32 x 64) as 4 x 8
    double va[2][16], wa[4][8];
    auto load_v = [&](int k) {
      const d2_t* __restrict__ blk = reinterpret_cast<const d2_t*>(p.blocks + ((size_t)Gw * p.K + k) * 4096);
#pragma unroll
      for (int xa = 0; xa < 2; ++xa)
#pragma unroll
        for (int kp = 0; kp < 8; ++kp)
          if (xa == 0 ? kp < 6 : kp >= 2) {
            const d2_t v = blk[(xa * 8 + kp) * 64 + lane];
            va[xa][2 * kp] = v.x;
            va[xa][2 * kp + 1] = v.y;
          }
    };
    auto load_w = [&](int k) {
      const d2_t* __restrict__ blk = reinterpret_cast<const d2_t*>(p.blocks + ((size_t)Gw * p.K + k) * 4096 + 2048);
#pragma unroll
      for (int zr = 0; zr < 4; ++zr)
#pragma unroll
        for (int kp = 0; kp < 4; ++kp)
          if (zr < 3 || kp >= 2) {          // (W = V T: rows 48-63 only carry reflectors 17-31)
            const d2_t v = blk[(zr * 4 + kp) * 64 + lane];
            wa[zr][2 * kp] = v.x;
            wa[zr][2 * kp + 1] = v.y;
          }
    };
    if (nkw > 0) {
      load_v(0);
      load_w(0);
    }
    for (int t = 0; t < T; ++t) {
      const int b0 = Gtop + t;
      Q2_STAMP(0);
      d2_t znew[NBT];
      load_band(b0 + 2, znew);                             // the band that arrives for the next step
      lds_barrier();                                       // window complete
      Q2_STAMP(1);
      const int k = t - 2 * wave;
      if (k >= 0 && k < nkw) {
        const int bw = b0 - 3 * wave;                      // this wave's bands: bw, bw + 1
        double* z0 = slot(bw) + c0 * TILE;
        double* z1 = slot(bw + 1) + c0 * TILE;
        // X = V^T Zw  (32 x 16 per tile, k = 64 window rows)
        d4_t ax[NM][2];
#pragma unroll
        for (int ct = 0; ct < NM; ++ct) ax[ct][0] = ax[ct][1] = d4_t{0.0, 0.0, 0.0, 0.0};
        // (V is a staircase: reflector j lives on window rows j .. j + 31, so reflectors 0-15 never see rows 48-63 and
        // reflectors 16-31 never see rows 0-15: 12 of the 16 k-steps each)
#pragma unroll
        for (int kq = 0; kq < 16; ++kq)
#pragma unroll
          for (int ct = 0; ct < NM; ++ct) {
            const double zb = (kq < 8 ? z0 : z1)[ct * TILE + ((kq & 7) * 4 + lq) * XLD + lr];
            if (kq < 12) ax[ct][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[0][kq], zb, ax[ct][0], 0, 0, 0);
            if (kq >= 4) ax[ct][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[1][kq], zb, ax[ct][1], 0, 0, 0);
          }
        Q2_STAMP(2);
        if (k + 1 < nkw) load_v(k + 1);                    // (V's registers are free: the next block's travel under the second product)
#pragma unroll
        for (int ct = 0; ct < NM; ++ct)
#pragma unroll
          for (int xa = 0; xa < 2; ++xa)
#pragma unroll
            for (int r = 0; r < 4; ++r) xs[ct * TILE + (xa * 16 + 4 * r + lq) * XLD + lr] = ax[ct][xa][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private tiles: written and read by this wave only
        double xb[NM][8];
#pragma unroll
        for (int ct = 0; ct < NM; ++ct)
#pragma unroll
          for (int kq = 0; kq < 8; ++kq) xb[ct][kq] = xs[ct * TILE + (kq * 4 + lq) * XLD + lr];
        Q2_STAMP(3);
        // Zw -= W X  (64 x 16 per tile, k = 32 reflectors): two row tiles at a time, so that two accumulator chains interleave
#pragma unroll
        for (int zp = 0; zp < 2; ++zp)
#pragma unroll
          for (int ct = 0; ct < NM; ++ct) {
            double* zh = (zp == 0 ? z0 : z1) + ct * TILE;
            d4_t acc0, acc1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              acc0[r] = zh[(4 * r + lq) * XLD + lr];
              acc1[r] = zh[(16 + 4 * r + lq) * XLD + lr];
            }
#pragma unroll
            for (int kq = 0; kq < 8; ++kq) {
              acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[2 * zp][kq], xb[ct][kq], acc0, 0, 0, 1);   // -A
              if (zp == 0 || kq >= 4) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[2 * zp + 1][kq], xb[ct][kq], acc1, 0, 0, 1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              zh[(4 * r + lq) * XLD + lr] = acc0[r];
              zh[(16 + 4 * r + lq) * XLD + lr] = acc1[r];
            }
          }
        Q2_STAMP(4);
        if (k + 1 < nkw) load_w(k + 1);
      }
      lds_barrier();                                       // every wave is done with its bands
      Q2_STAMP(5);
      // band b0 - 9 is final for this pass; its slot is NOT the arriving band's (the ring has one spare slot), so the arriving
      // rows go to LDS first and the store follows (a wait for the next load must not also wait for this store)
      d2_t fin[NBT];
      get_band(b0 - 9, fin);
      put_band(b0 + 2, znew);
      store_band(b0 - 9, fin);
      Q2_STAMP(6);
    }
    lds_barrier();
    // what is still in the window after the last step: bands (Gtop + T) - 9 .. (Gtop + T) + 1
    for (int b = Gtop + T - 9; b <= Gtop + T + 1; ++b) {
      d2_t v[NBT];
      get_band(b, v);
      store_band(b, v);
    }
  }
}

// Z [n, ldz] (first ncols columns) <- Q2 Z (fwd = 0; blocks prepared with trans = 0) or Q2^T Z (fwd = 1; trans = 1), the sweep
// groups [G0, G1) only.  skip8: leave the XCD the chase runs on alone.
int ffgp_q2_apply_impl(ffgp_handle* h, const double* blocks, int n, double* Z, int ldz, int ncols, int G0, int G1, int fwd, int skip8) {
  if (G1 <= G0) return FFGP_OK;
  ApplyArgs a;
  a.blocks = blocks; a.n = n; a.K = chase_K(n); a.Z = Z; a.ldz = ldz; a.ncols = ncols; a.dbg = h->diag_dbg;
  a.G0 = G0; a.G1 = G1; a.skip8 = skip8;
  const int nslab = (ncols + 15) / 16;
  if (!fwd && h->q2_blocks_lanes) {
    if (skip8) return FFGP_ERR_ARG;
    static bool attr_set[64] = {false};
    if (h->device >= 0 && h->device < 64 && !attr_set[h->device]) {
      FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(q2_apply_wave4<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   Q2W_LDS_DOUBLES(1) * (int)sizeof(double)));
      FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(q2_apply_wave4<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   Q2W_LDS_DOUBLES(2) * (int)sizeof(double)));
      attr_set[h->device] = true;
    }
    // 16-column slabs (two workgroups per CU) measured faster than 32-column ones at every size: N = 8192 51.9 / 57.4 ms,
    // N = 16384 401 / 424 ms (the old kernel: 80.4 / 712); option value 2 selects the wide form
    const bool wide = (h->q2_wave4 == 2);
    // 32-column slabs on eight waves (the SPLIT form) once there are enough of them to fill the chip: N = 8192 45.4 -> 38.9 ms (256 slabs:
    // one per CU); at N = 4096 its 128 workgroups leave half the CUs empty (7.2 -> 10.2 ms), so smaller matrices keep the 16-column form
    if (h->q2_wave4 == 3 || (h->q2_wave4 == 1 && ncols >= h->q2_split_min_cols)) {
      static bool attr3[64] = {false};
      if (h->device >= 0 && h->device < 64 && !attr3[h->device]) {
        FFGP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(q2_apply_wave4<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     Q2W_LDS_DOUBLES(2) * (int)sizeof(double)));
        attr3[h->device] = true;
      }
      hipLaunchKernelGGL((q2_apply_wave4<2, true>), dim3((ncols + 31) / 32), dim3(512), Q2W_LDS_DOUBLES(2) * sizeof(double), h->stream, a);
    } else if (wide) hipLaunchKernelGGL(q2_apply_wave4<2>, dim3((ncols + 31) / 32), dim3(256), Q2W_LDS_DOUBLES(2) * sizeof(double), h->stream, a);
    else hipLaunchKernelGGL(q2_apply_wave4<1>, dim3(nslab), dim3(256), Q2W_LDS_DOUBLES(1) * sizeof(double), h->stream, a);
    return hipGetLastError() == hipSuccess ? FFGP_OK : FFGP_ERR_HIP;
  }
  const int grid = skip8 ? nslab + (nslab + 6) / 7 + 1 : nslab;
  if (fwd) hipLaunchKernelGGL(q2_apply<true>, dim3(grid), dim3(256), 0, h->stream, a);
  else hipLaunchKernelGGL(q2_apply<false>, dim3(grid), dim3(256), 0, h->stream, a);
  return hipGetLastError() == hipSuccess ? FFGP_OK : FFGP_ERR_HIP;
}
