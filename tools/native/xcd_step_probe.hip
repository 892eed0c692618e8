// One step of the factorisation's dependency chain as an XCD-local cooperative kernel -- the probe VERDICT r5 item 1(a) asks for.
//
// Today a chain step is three dependent launches: diagonal block (28-29.5 us, one workgroup) -> TRSM of the rows below (7.7 us, of which
// 5.0 are the floor of a dependent launch) -> update (5.6-9.4 us, floor 5.0 again) = ~44 us.  The question: what does the step cost when
// the work behind the diagonal block -- for the NEAR rows only, the 128 rows of the next diagonal block -- is done by 8 helper workgroups
// that were launched WITH the diagonal block's workgroup, sit on the SAME XCD (HW_REG_XCC_ID checked, surplus leaves) and are released by
// a plain-store flag in that XCD's L2 (no device-scope release: no L2 write-back / invalidate), with one XCD-local barrier between a
// TRSM-sized and an update-sized MFMA phase, and a flag back?
//
//   chain3     leader kernel (spin T us, writes a 128 x 128 block) ; phase-1 kernel (H workgroups x 16 rows: X = A * Dinv^T on MFMA) ;
//              phase-2 kernel (H workgroups: C -= X * Xnear^T) -- three launches per step on one stream: today's structure
//   fused      ONE launch for all steps (persistent): leader + H helpers on one XCD, flags in L2 as described
//   perstep    one launch PER STEP holding leader + helpers (helpers released by the leader's flag; the next step starts at the kernel
//              boundary instead of a flag back) -- what "near rows inside the diagonal block's launch" would be
// Every wait is bounded (watchdog -> err word, the grid always drains).  Results are checked: C after S steps must equal the host's.
//   hipcc -O3 --offload-arch=gfx950 tools/native/xcd_step_probe.hip -o tools/native/xcd_step_probe && tools/native/xcd_step_probe [T_us] [H]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

struct Args {
  double* D;          // [128 x 128] the "inverted diagonal block" the leader writes every step (value depends on the step)
  double* A;          // [H*16 x 128] near rows before the solve (constant)
  double* X;          // [H*16 x 128] phase-1 result (rows of L)
  double* C;          // [H*16 x 128] phase-2 accumulator
  unsigned* flag;     // leader -> helpers: step + 1
  unsigned* bar;      // helpers' barrier counter (monotone)
  unsigned* done;     // helpers -> leader (monotone)
  unsigned* ticket;   // role hand-out among the workgroups that landed on the XCD
  int* err;
  int H, steps, step0;
  long spin_ticks;    // leader's spin in s_memtime ticks (100 MHz)
  int xcc;            // -1: any XCD (chain3 / placement-free forms)
  long* stamps;       // fused persistent form: wall_clock64 (100 MHz) at the points of a step, [step][8]; nullptr = off
};
// stamp slots of a step: 0 leader's spin over, 1 leader published, 2 helper 0 saw the flag, 3 helper 0's phase 1 stored and drained,
// 4 helper 0 through the barrier, 5 helper 0's phase 2 stored and drained, 6 leader saw every helper done
#define STAMP(k) do { if (p.stamps && threadIdx.x == 0) p.stamps[(size_t)s * 8 + (k)] = wall_clock64(); } while (0)

__device__ __forceinline__ unsigned ld_l2(const unsigned* p) {          // bypasses this CU's L1, served by the XCD's L2 (sc1)
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ldd_l2(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool wait_ge(const unsigned* p, unsigned need, int* err) {
  int it = 0;
  while (ld_l2(p) < need) {
    __builtin_amdgcn_s_sleep(1);
    if (++it > (1 << 22) || ((it & 4095) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
      __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
  return true;
}

// leader's part of a step: spin, then write the block (plain stores: they stay in this XCD's L2), drain, publish
__device__ __forceinline__ void leader_step(const Args& p, int s, bool publish) {
  const int tid = threadIdx.x;
  if (tid == 0) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < p.spin_ticks) __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
  if (publish) STAMP(0);
  const double v = 1.0 / (double)(s + 2);
  for (int i = tid; i < 128 * 128; i += blockDim.x) {
    const int r = i >> 7, c = i & 127;
    p.D[i] = (c <= r) ? v * (1.0 + 0.001 * (double)((r * 7 + c * 3) & 15)) : 0.0;      // lower triangular, like Dinv
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (publish && tid == 0) __hip_atomic_store(p.flag, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // plain store
  if (publish) STAMP(1);
}

typedef double d2 __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
#define LDS_LD 130
// 16-byte load that bypasses this CU's L1 (sc1): what another workgroup of this launch wrote is served by the XCD's L2
__device__ __forceinline__ d2 ld16_l2(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const v4i v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16);
  d2 o;
  o.x = __hiloint2double(v.y, v.x);
  o.y = __hiloint2double(v.w, v.z);
  return o;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
// (as ffgp_trsm128_kernel: one 16-byte load per lane fetches a k-pair of the lane's row, two row swaps re-deal it to MFMA k-steps)
__device__ __forceinline__ void kpair_to_ksteps(d2 in, double& u, double& v) {
  const unsigned long long xb = __builtin_bit_cast(unsigned long long, (double)in.x), yb = __builtin_bit_cast(unsigned long long, (double)in.y);
  auto l1 = __builtin_amdgcn_permlane16_swap((unsigned)xb, (unsigned)yb, false, false);
  auto l2 = __builtin_amdgcn_permlane32_swap(l1[0], l1[1], false, false);
  auto h1 = __builtin_amdgcn_permlane16_swap((unsigned)(xb >> 32), (unsigned)(yb >> 32), false, false);
  auto h2 = __builtin_amdgcn_permlane32_swap(h1[0], h1[1], false, false);
  u = __builtin_bit_cast(double, ((unsigned long long)h2[0] << 32) | l2[0]);
  v = __builtin_bit_cast(double, ((unsigned long long)h2[1] << 32) | l2[1]);
}

// Both phases follow ffgp_trsm128_kernel (potrf.hip): a 256-thread workgroup owns 16 rows; its rows go through LDS (one coalesced
// 16-byte load per lane and chunk, ONE barrier, all A operands into registers in one batch); the B operand's rows come straight
// from L2 as 16-byte k-pairs; wave w owns the 16-column blocks w and 7 - w.
// phase 1: X[rows, :] = A[rows, :] * D^T, D lower triangular (block c of the k range only while c <= column block: 36 MFMA k-steps)
__device__ __forceinline__ void phase1(const Args& p, int hb, double* sA, double* sX) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = hb * 16;
  d2 va[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i, row = idx >> 6, c2 = idx & 63;
    va[i] = *reinterpret_cast<const d2*>(p.A + (size_t)(r0 + row) * 128 + 2 * c2);
  }
  const int bL = wave, bH = 7 - wave;
  const int j = lane & 15, kk = lane >> 4;
  const __amdgpu_buffer_rsrc_t rD = rsrc_of(p.D);
  const unsigned oH = (unsigned)(((16 * bH + j) * 128 + 2 * kk) * 8), oL = (unsigned)(((16 * bL + j) * 128 + 2 * kk) * 8);
  d2 rh[16], rl[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c <= bH) {
      rh[2 * c] = ld16_l2(rD, oH + 128 * c);
      rh[2 * c + 1] = ld16_l2(rD, oH + 128 * c + 64);
    }
    if (c < 4 && c <= bL) {
      rl[2 * c] = ld16_l2(rD, oL + 128 * c);
      rl[2 * c + 1] = ld16_l2(rD, oL + 128 * c + 64);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + 256 * i, row = idx >> 6, c2 = idx & 63;
    *reinterpret_cast<d2*>(sA + row * LDS_LD + 2 * c2) = va[i];
  }
  __syncthreads();
  const double* aP = sA + j * LDS_LD + kk;
  double a[32];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c <= bH) {
#pragma unroll
      for (int q = 0; q < 4; ++q) a[4 * c + q] = aP[16 * c + 4 * q];
    }
  }
  d4 accL = {0.0, 0.0, 0.0, 0.0}, accH = {0.0, 0.0, 0.0, 0.0};
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
    const int row = kk + 4 * r;
    double* out = p.X + (size_t)(r0 + row) * 128 + j;
    out[16 * bL] = accL[r];
    out[16 * bH] = accH[r];
    sX[row * LDS_LD + j + 16 * bL] = accL[r];       // the workgroup's own rows of X stay in LDS for phase 2
    sX[row * LDS_LD + j + 16 * bH] = accH[r];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// phase 2: C[rows, :] -= X[rows, :] * X[near rows, :]^T  (the near rows are rows 0..127: written by helpers 0..7), k = 128: 64 MFMAs per wave
__device__ __forceinline__ void phase2(const Args& p, int hb, const double* sX, bool own_rows_in_lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = hb * 16;
  const int bL = wave, bH = 7 - wave;
  const int j = lane & 15, kk = lane >> 4;
  const __amdgpu_buffer_rsrc_t rX = rsrc_of(p.X);
  const unsigned oH = (unsigned)(((16 * bH + j) * 128 + 2 * kk) * 8), oL = (unsigned)(((16 * bL + j) * 128 + 2 * kk) * 8);
  d2 rh[16], rl[16];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    rh[2 * c] = ld16_l2(rX, oH + 128 * c);
    rh[2 * c + 1] = ld16_l2(rX, oH + 128 * c + 64);
    rl[2 * c] = ld16_l2(rX, oL + 128 * c);
    rl[2 * c + 1] = ld16_l2(rX, oL + 128 * c + 64);
  }
  d4 accL, accH;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double* c_ = p.C + (size_t)(r0 + kk + 4 * r) * 128 + j;
    accL[r] = ldd_l2(c_ + 16 * bL);
    accH[r] = ldd_l2(c_ + 16 * bH);
  }
  double a[32];
  if (own_rows_in_lds) {
    const double* aP = sX + j * LDS_LD + kk;
#pragma unroll
    for (int i = 0; i < 32; ++i) a[i] = aP[4 * i];
  } else {      // (chain3: a kernel of its own -- the rows come from memory)
    const double* aP = p.X + (size_t)(r0 + j) * 128 + kk;
#pragma unroll
    for (int i = 0; i < 32; ++i) a[i] = ldd_l2(aP + 4 * i);
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    double bh[4], bl[4];
    kpair_to_ksteps(rh[2 * c], bh[0], bh[1]);
    kpair_to_ksteps(rh[2 * c + 1], bh[2], bh[3]);
    kpair_to_ksteps(rl[2 * c], bl[0], bl[1]);
    kpair_to_ksteps(rl[2 * c + 1], bl[2], bl[3]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      accH = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[4 * c + q], bh[q], accH, 0, 0, 0);
      accL = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[4 * c + q], bl[q], accL, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double* out = p.C + (size_t)(r0 + kk + 4 * r) * 128 + j;
    out[16 * bL] = accL[r];
    out[16 * bH] = accH[r];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(256) void k_leader(Args p, int s) { leader_step(p, s, false); }
__global__ __launch_bounds__(256) void k_phase1(Args p) {
  __shared__ __attribute__((aligned(16))) double sA[16 * LDS_LD], sX[16 * LDS_LD];
  phase1(p, blockIdx.x, sA, sX);
}
__global__ __launch_bounds__(256) void k_phase2(Args p) { phase2(p, blockIdx.x, nullptr, false); }

// leader + helpers in one launch.  PERSIST: all steps inside (flag back to the leader); else ONE step (p.step0), the kernel boundary
// is the flag back.
template <bool PERSIST>
__global__ __launch_bounds__(256) void k_fused(Args p) {
  __shared__ int role_s;
  __shared__ __attribute__((aligned(16))) double sA[16 * LDS_LD], sX[16 * LDS_LD];
  if (p.xcc >= 0) {
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if ((int)(xcc & 0xfu) != p.xcc) return;
  }
  if (threadIdx.x == 0)      // (per-step launches: every step has its own ticket word)
    role_s = (int)__hip_atomic_fetch_add(p.ticket + (PERSIST ? 0 : p.step0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __syncthreads();
  const int role = role_s;
  if (role > p.H) return;                                     // surplus workgroups of this XCD
  const int s0 = PERSIST ? 0 : p.step0, s1 = PERSIST ? p.steps : p.step0 + 1;
  if (role == 0) {
    for (int s = s0; s < s1; ++s) {
      leader_step(p, s, true);
      if (PERSIST) {
        bool ok = true;
        if (threadIdx.x == 0) ok = wait_ge(p.done, (unsigned)(p.H * (s + 1)), p.err);
        STAMP(6);
        __syncthreads();
        (void)ok;
        if (__hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
      }
    }
    return;
  }
  const int hb = role - 1;
  for (int s = s0; s < s1; ++s) {
    if (threadIdx.x == 0) wait_ge(p.flag, (unsigned)(s + 1), p.err);
    if (hb == 0) STAMP(2);
    __syncthreads();
    if (__hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    phase1(p, hb, sA, sX);
    __syncthreads();
    if (hb == 0) STAMP(3);
    if (threadIdx.x == 0) {                                   // XCD-local barrier of the helpers: an atomic in this XCD's L2, polled there
      __hip_atomic_fetch_add(p.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      wait_ge(p.bar, (unsigned)(p.H * (s + 1)), p.err);
    }
    if (hb == 0) STAMP(4);
    __syncthreads();
    if (__hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    phase2(p, hb, sX, true);
    __syncthreads();
    if (hb == 0) STAMP(5);
    if (PERSIST && threadIdx.x == 0) __hip_atomic_fetch_add(p.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

static void host_ref(int H, int steps, const std::vector<double>& A, std::vector<double>& C) {
  const int R = H * 16;
  std::vector<double> D(128 * 128), X((size_t)R * 128);
  for (int s = 0; s < steps; ++s) {
    const double v = 1.0 / (double)(s + 2);
    for (int r = 0; r < 128; ++r)
      for (int c = 0; c < 128; ++c) D[r * 128 + c] = (c <= r) ? v * (1.0 + 0.001 * (double)((r * 7 + c * 3) & 15)) : 0.0;
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < 128; ++j) {
        double acc = 0.0;
        for (int k = 0; k < 128; ++k) acc += A[(size_t)i * 128 + k] * D[j * 128 + k];
        X[(size_t)i * 128 + j] = acc;
      }
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < 128; ++j) {
        double acc = 0.0;
        for (int k = 0; k < 128; ++k) acc += X[(size_t)i * 128 + k] * X[(size_t)j * 128 + k];
        C[(size_t)i * 128 + j] -= acc;
      }
  }
}

int main(int argc, char** argv) {
  const double T_us = argc > 1 ? atof(argv[1]) : 28.0;
  const int H = argc > 2 ? atoi(argv[2]) : 8;
  const int S = 200;
  if (H < 8 || H > 64) { printf("H in 8..64\n"); return 1; }
  const int R = H * 16;
  Args p;
  CK(hipMalloc(&p.D, 128 * 128 * 8));
  CK(hipMalloc(&p.A, (size_t)R * 128 * 8));
  CK(hipMalloc(&p.X, (size_t)R * 128 * 8));
  CK(hipMalloc(&p.C, (size_t)R * 128 * 8));
  unsigned* words;
  CK(hipMalloc(&words, 4096));
  p.flag = words; p.bar = words + 64; p.done = words + 128; p.ticket = words + 320; p.err = (int*)(words + 256);   // ticket: S + 1 words
  p.H = H; p.steps = S; p.step0 = 0; p.spin_ticks = (long)(T_us * 100.0); p.xcc = -1; p.stamps = nullptr;
  long* d_stamps;
  CK(hipMalloc(&d_stamps, (size_t)S * 8 * sizeof(long)));
  std::vector<double> A((size_t)R * 128), C0((size_t)R * 128, 1.0), Cref;
  for (size_t i = 0; i < A.size(); ++i) A[i] = 0.5 + 0.25 * std::sin(0.37 * (double)i);
  CK(hipMemcpy(p.A, A.data(), A.size() * 8, hipMemcpyHostToDevice));
  Cref = C0;
  host_ref(H, S, A, Cref);
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto reset = [&]() {
    hipMemcpy(p.C, C0.data(), C0.size() * 8, hipMemcpyHostToDevice);
    hipMemset(words, 0, 4096);
    hipMemset(p.X, 0, (size_t)R * 128 * 8);
    hipDeviceSynchronize();
  };
  auto check = [&](const char* name, float ms) {
    std::vector<double> C((size_t)R * 128);
    hipMemcpy(C.data(), p.C, C.size() * 8, hipMemcpyDeviceToHost);
    int err = 0;
    hipMemcpy(&err, p.err, 4, hipMemcpyDeviceToHost);
    double worst = 0.0, scale = 0.0;
    for (size_t i = 0; i < C.size(); ++i) { worst = std::fmax(worst, std::fabs(C[i] - Cref[i])); scale = std::fmax(scale, std::fabs(Cref[i])); }
    printf("%-28s %8.2f us/step   (T = %.1f us spin, H = %d helpers x 16 rows; rel err %.1e%s)\n", name, ms * 1e3 / S, T_us, H, worst / scale,
           err ? "; WATCHDOG FIRED" : "");
  };
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    // --- today's structure: three dependent launches per step
    reset();
    CK(hipEventRecord(e0, st));
    for (int s = 0; s < S; ++s) {
      hipLaunchKernelGGL(k_leader, dim3(1), dim3(256), 0, st, p, s);
      hipLaunchKernelGGL(k_phase1, dim3(H), dim3(256), 0, st, p);
      hipLaunchKernelGGL(k_phase2, dim3(H), dim3(256), 0, st, p);
    }
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) check("chain3 (3 launches/step)", ms);
    // --- leader alone (the floor: what the step costs when the rest is free)
    reset();
    CK(hipEventRecord(e0, st));
    for (int s = 0; s < S; ++s) hipLaunchKernelGGL(k_leader, dim3(1), dim3(256), 0, st, p, s);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("%-28s %8.2f us/step\n", "leader kernel alone", ms * 1e3 / S);
    // --- persistent XCD-local, on each of two XCDs, and without the XCD check (helpers anywhere on the chip: plain stores are then
    //     NOT a correct hand-off; shown for the timing of the cross-XCD case only when it happens to verify)
    for (int xcc : {0, 5}) {
      reset();
      Args q = p;
      q.xcc = xcc;
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(k_fused<true>, dim3(8 * (H + 1 + 7)), dim3(256), 0, st, q);
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventElapsedTime(&ms, e0, e1));
      char nm[64];
      snprintf(nm, sizeof nm, "fused persistent, XCD %d", xcc);
      if (rep) check(nm, ms);
    }
    if (rep) {      // once more with stamps: where the fused step's time goes (averages over the steps, in us; 100 MHz clock)
      reset();
      Args q = p;
      q.xcc = 0;
      q.stamps = d_stamps;
      CK(hipMemset(d_stamps, 0, (size_t)S * 8 * sizeof(long)));
      hipLaunchKernelGGL(k_fused<true>, dim3(8 * (H + 1 + 7)), dim3(256), 0, st, q);
      CK(hipStreamSynchronize(st));
      std::vector<long> t((size_t)S * 8);
      CK(hipMemcpy(t.data(), d_stamps, t.size() * sizeof(long), hipMemcpyDeviceToHost));
      double seg[6] = {0, 0, 0, 0, 0, 0};
      for (int s2 = 10; s2 < S; ++s2)
        for (int k = 0; k < 6; ++k) seg[k] += (double)(t[(size_t)s2 * 8 + k + 1] - t[(size_t)s2 * 8 + k]) * 0.01;
      const char* names[6] = {"leader: block written, drained, flag stored", "hop: flag store -> helper 0 sees it", "phase 1 (loads, 36 MFMA k-steps, stores drained)",
                              "helpers' barrier (atomic in L2 + poll)", "phase 2 (loads, 64 MFMA k-steps, stores drained)", "hop back: done counter -> leader"};
      printf("  fused step behind the leader's spin, by wall_clock64 stamps (us, mean of %d steps):\n", S - 10);
      double tot = 0.0;
      for (int k = 0; k < 6; ++k) { printf("    %-52s %6.2f\n", names[k], seg[k] / (S - 10)); tot += seg[k] / (S - 10); }
      printf("    %-52s %6.2f\n", "sum", tot);
    }
    // --- one launch per step: leader + helpers, the kernel boundary is the flag back
    reset();
    {
      Args q = p;
      q.xcc = 0;
      CK(hipEventRecord(e0, st));
      for (int s = 0; s < S; ++s) {
        q.step0 = s;
        hipLaunchKernelGGL(k_fused<false>, dim3(8 * (H + 1 + 7)), dim3(256), 0, st, q);
      }
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep) check("per-step launch, XCD 0", ms);
    }
  }
  return 0;
}
