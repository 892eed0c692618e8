// Internal declarations of the two-stage symmetric eigensolver (sy2sb.hip, sb2st.hip, stedc.hip, syevd.hip).
#pragma once
#include "ffgp_internal.h"

#define SB_B 32     // bandwidth after stage 1
#define SB_LDB 64   // band storage: element (r, c), r >= c, at AB[c * 64 + (r - c)]; 2b rows leave room for the chase's bulges

// ---- stage 1 (sy2sb.hip)
size_t ffgp_sy2sb_ws_doubles(int n);
int ffgp_sy2sb_impl(ffgp_handle* h, double* A, int n, int lda, double* AB, double* Y, int ldy, double* Tpan, double* ws);
// ---- stage 2 (sb2st.hip)
int ffgp_sb2st_impl(ffgp_handle* h, double* AB, int n, double* d, double* e, double* V2, double* tau2, int* prog);
int ffgp_sb2st_init(ffgp_handle* h, hipStream_t st, int n, double* V2, double* tau2, int* prog);
int ffgp_sb2st_chunk(ffgp_handle* h, hipStream_t st, double* AB, int n, double* d, double* e, double* V2, double* tau2, int* prog, int s_begin,
                     int s_end);
int ffgp_sb2st_finish(ffgp_handle* h, hipStream_t st, const double* AB, int n, double* d, double* e);
size_t ffgp_q2_block_doubles(int n);
int ffgp_q2_prep_impl(ffgp_handle* h, const double* V2, const double* tau2, int n, double* blocks, int G0, int G1, int trans);
int ffgp_q2_apply_impl(ffgp_handle* h, const double* blocks, int n, double* Z, int ldz, int ncols, int G0, int G1, int fwd, int skip8);
// ---- stage 3 (stedc.hip)
size_t ffgp_stedc_ws_doubles(int n);
int ffgp_stedc_impl(ffgp_handle* h, const double* d, const double* e, int n, double* lam, double* Z, int ldz, double* ws);
// ---- back-transformation with the stage-1 reflectors (syevd.hip)
size_t ffgp_q1_ws_doubles(int n, int ncols);
int ffgp_q1_apply_impl(ffgp_handle* h, const double* Y, int ldy, int n, double* Z, int ldz, int ncols, double* ws, int trans);
// eigensolver workspace of the handle
int ffgp_ensure_ews(ffgp_handle* h, size_t bytes);

// ---- half-wave sums (sy2sb.hip: column norms and projections of the panel QRs, T-factor and triangular recurrences; sb2st.hip: the
// T factors of q2_prep).  A recurrence x_j = f(sum_{i < j} a_ji x_i) over 32 unknowns runs on one half-wave with x_i in lane i: one
// product per lane and one of these sums per step, no barrier -- in place of a 32-thread loop with a serial inner sum.
#ifdef __HIPCC__
template <int CTRL>
__device__ __forceinline__ double qr_dpp_add(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return x + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double qr_rdlane(double x, int l) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}
// sum over the 32 lanes of each half-wave, in all of them (DPP inside the rows of 16, then the two rows of a half exchanged)
__device__ __forceinline__ double qr_wsum32(double x, int lane) {
  x = qr_dpp_add<0xB1>(x);
  x = qr_dpp_add<0x4E>(x);
  x = qr_dpp_add<0x141>(x);
  x = qr_dpp_add<0x140>(x);
  // every lane now holds the total of its row of 16; rows 0 + 1 and 2 + 3 through v_permlane16_swap (gfx950): of two copies of x one
  // ends up with the even rows' totals in both rows of a pair, the other with the odd rows' -- five instructions where the route
  // through scalar registers (eight readlanes, two adds, a select) took twelve, and bit-identical to it (tools/native/wsum_check.hip)
  (void)lane;
  const int lo = __double2loint(x), hi = __double2hiint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
#endif
