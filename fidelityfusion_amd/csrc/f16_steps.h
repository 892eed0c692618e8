// The in-register 16 x 16 pivot step of the diagonal-block kernel (potrf.hip, ffgp_potrf_diag128_v2) and its lane primitives: a header
// of their own so that tools/native/f16_probe.hip can time the bare pivot loop with the very same code.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ double readlane_d(double x, int l) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

// value of x held by lane (16*(lane>>4) + J): DPP row broadcast inside each row of 16 lanes
template <int J>
__device__ __forceinline__ double row_bcast_d(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, 0x150 + J, 0xf, 0xf, true);   // every lane receives data: no `old` value to seed
  hi = __builtin_amdgcn_mov_dpp(hi, 0x150 + J, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// value of x held by lane `src` (per-lane source): ds_bpermute
__device__ __forceinline__ double bperm_d(double x, int src) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_ds_bpermute(src << 2, lo);
  hi = __builtin_amdgcn_ds_bpermute(src << 2, hi);
  return __hiloint2double(hi, lo);
}

// one pivot of the pipelined in-register factor (see the header comment).  lane (g = lane>>4, c = lane&15) holds rows
// g+4r of column c of the symmetric block (v) and of the eliminated identity (w); rowA / rowW = current row J of both.
template <int J>
__device__ __forceinline__ void f16_step(double (&v)[4], double (&w)[4], double& rowA, double& rowW, int c, int g) {
  double preA = 0.0, preW = 0.0;
  if constexpr (J < 15) {   // row J+1 as it stands BEFORE this pivot's update; patched below
    constexpr int PR1 = (J + 1) >> 2, G1 = (J + 1) & 3;
    preA = bperm_d(v[PR1], 16 * G1 + c);
    preW = bperm_d(w[PR1], 16 * G1 + c);
  }
  const double d = row_bcast_d<J>(rowA);                     // A[J][J]  (checked for positivity after the 16 steps)
  const double y0 = __builtin_amdgcn_rcp(d);
  const double e = __builtin_fma(-d, y0, 1.0);
  const double f = __builtin_fma(e, e, e);                   // 1/d = y0 (1 + e + e^2)
  const double u = rowA * y0;
  const double t = __builtin_fma(u, f, u);                   // A[J][c] / d
  const double uw = rowW * y0;
  const double tw = __builtin_fma(uw, f, uw);                // W[J][c] / d
  // registers whose four rows (g + 4r, g = 0..3) are all <= J hold finished rows: neither block is updated there
  constexpr int RMIN = (J + 1) >> 2;
  double colj[4];
#pragma unroll
  for (int r = RMIN; r < 4; ++r) colj[r] = row_bcast_d<J>(v[r]);   // A[g+4r][J]
  if constexpr (J < 15) {
    const double s = row_bcast_d<(J + 1) & 15>(rowA);        // A[J][J+1] = A[J+1][J]
    rowA = __builtin_fma(-s, t, preA);
    rowW = __builtin_fma(-s, tw, preW);
  }
  const double tm = (c > J) ? t : 0.0;                       // columns <= J are parked: they keep the unscaled L column
#pragma unroll
  for (int r = RMIN; r < 4; ++r) v[r] = __builtin_fma(-colj[r], tm, v[r]);
  constexpr int PR = J >> 2;
  if constexpr (PR >= RMIN) colj[PR] = (g == (J & 3)) ? 0.0 : colj[PR];   // the pivot row of W stays
#pragma unroll
  for (int r = RMIN; r < 4; ++r) w[r] = __builtin_fma(-colj[r], tw, w[r]);
}

// ---- the same pivot on the DP-ALU DPP forms of gfx950 (round 4) -------------------------------------------------------------
// gfx90a+ let two fp64 instructions read their first operand through DPP with the `row_newbcast` control: v_mov_b64_dpp and
// v_fmac_f64_dpp.  A rank-1 term  x -= A[.][J] * t  whose left factor is a row broadcast is then ONE instruction
// (v_fmac_f64_dpp x, -src row_newbcast:J, t) where the 32-bit form needs two v_mov_b32_dpp and a v_fma_f64; the row mask of the
// DPP control also expresses "the pivot row of W stays" (rows of 16 lanes = the g coordinate) without a select.  Per pivot:
// 20 vector instructions + 4 ds_bpermute instead of 37 + 4, same operations in the same order on the same values (fma(-a, b, c)
// either way), so the factor and the inverse are bit-identical to f16_step's.
// Hazard (VALU writes a VGPR, a DPP instruction reads it within 2 wait states): the compiler cannot see into inline asm, so the
// block is ordered by hand -- the first statement starts with `s_nop 1` and names every operand of the later ones as an input, so
// all of them are defined in front of it; the rows' own updates come first (their result is read through DPP by the next pivot's
// first instruction, >= 2 instructions later even at J = 14); v[r] / w[r] are only DPP-read one whole pivot after they were written.
template <int J>
__device__ __forceinline__ double row_bcast64(double x) {
  return __builtin_amdgcn_update_dpp(0.0, x, 0x150 + J, 0xf, 0xf, true);      // v_mov_b64_dpp ... row_newbcast:J
}

#define F16_FMAC_DPP(dst, src, mul, JJ, RM)                                                                     \
  asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:%4 bank_mask:0xf" : "+v"(dst) : "v"(src), "v"(mul), "n"(JJ), "n"(RM))

// State of the pipelined pivot loop.  The pivot row is replicated over the four lane rows (every lane needs t[c] for its column), so
// the next pivot row has to come from its home lanes through ds_bpermute: ~60 cycles alone, well over 100 while the seven helper
// waves read MFMA operands from the same LDS.  Round 3 fetched row J+1 at the start of step J and needed it at the end of the same
// step: the pivot loop ran at the LDS latency (320 cycles per pivot against 66 of arithmetic dependency), whatever the instruction
// count.  Now row J+2 is requested at the start of step J -- v then carries the updates 0..J-1 -- and patched with the rank-1
// terms of BOTH steps it missed (J and J+1) right before it becomes the pivot row, two steps later:
//   row_{J+1} = fetched_{J+1} - A[J-1][J+1] t_{J-1} - A[J][J+1] t_J      (A[k][J+1] = lane J+1 of pivot row k: a DPP row broadcast)
// The loop is software-pipelined by hand, one asm statement per pivot (the compiler orders volatile statements but pulls its own
// instructions across them: it put the dependent fma right behind the reciprocal and the block updates in front of the row patch):
//   patch row J+1 with the terms of steps J-1 and J  ->  broadcast its diagonal entry, start the reciprocal (what the next pivot
//   cannot start without)  ->  the 8 block updates of step J, which issue under that reciprocal's latency.
// Operands: 0 hA  1 hW  2 dn  3 yn  4-7 w[0..3]  8-11 v[0..3] | 12 pRow 13 pt 14 ptw 15 rowA 16 t 17 tw 18 tm | 19 J+1  20 J  21 row mask of
// the first W update (the pivot row of W stays)
#define F16_D(dst, src, mul, jj, rm) "v_fmac_f64_dpp %" #dst ", -%" #src ", %" #mul " row_newbcast:%" #jj " row_mask:%" #rm " bank_mask:0xf\n\t"
#define F16_F(dst, src, mul, jj) "v_fmac_f64_dpp %" #dst ", -%" #src ", %" #mul " row_newbcast:%" #jj " row_mask:0xf bank_mask:0xf\n\t"
#define F16_PREV F16_F(0, 12, 13, 19) F16_F(1, 12, 14, 19)                       /* the term of step J-1, late */
#define F16_ROW F16_F(0, 15, 16, 19) F16_F(1, 15, 17, 19)                        /* A/W[J+1][c] -= A[J][J+1] t/tw */ \
  "s_nop 0\n\t"                                                                 /* 2 wait states: VALU write -> DPP read */ \
  "v_mov_b64_dpp %2, %0 row_newbcast:%19 row_mask:0xf bank_mask:0xf\n\tv_rcp_f64_e32 %3, %2\n\t"
#define F16_UPD0 F16_D(4, 8, 17, 20, 21) F16_F(5, 9, 17, 20) F16_F(6, 10, 17, 20) F16_F(7, 11, 17, 20) \
                 F16_F(8, 8, 18, 20) F16_F(9, 9, 18, 20) F16_F(10, 10, 18, 20) F16_F(11, 11, 18, 20)
#define F16_UPD1 F16_D(5, 9, 17, 20, 21) F16_F(6, 10, 17, 20) F16_F(7, 11, 17, 20) F16_F(9, 9, 18, 20) F16_F(10, 10, 18, 20) F16_F(11, 11, 18, 20)
#define F16_UPD2 F16_D(6, 10, 17, 20, 21) F16_F(7, 11, 17, 20) F16_F(10, 10, 18, 20) F16_F(11, 11, 18, 20)
#define F16_UPD3 F16_D(7, 11, 17, 20, 21) F16_F(11, 11, 18, 20)
#define F16_ASM(BODY)                                                                                                              \
  asm volatile("s_nop 1\n\t" BODY "s_nop 0"                                                                                        \
               : "+v"(hA), "+v"(hW), "=&v"(dn), "=&v"(yn), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(v[0]), "+v"(v[1]),  \
                 "+v"(v[2]), "+v"(v[3])                                                                                             \
               : "v"(pRow), "v"(pt), "v"(ptw), "v"(rowA), "v"(t), "v"(tw), "v"(tm), "n"((J + 1) & 15), "n"(J), "n"(WMASK))

template <int J>
__device__ __forceinline__ void f16_step_dpp(double (&v)[4], double (&w)[4], double& rowA, double& rowW, double& hA, double& hW,
                                             double& pRow, double& pt, double& ptw, double& d, double& y0, int c, int g) {
  double fA = 0.0, fW = 0.0;
  if constexpr (J < 14) {   // row J+2 as it stands BEFORE this pivot's update: needed two steps from now
    constexpr int PR2 = (J + 2) >> 2, G2 = (J + 2) & 3;
    fA = bperm_d(v[PR2], 16 * G2 + c);
    fW = bperm_d(w[PR2], 16 * G2 + c);
  }
  // d = A[J][J] (checked for positivity after the 16 steps), y0 = v_rcp_f64(d): from the previous step's statement
  const double e = __builtin_fma(-d, y0, 1.0);
  const double f = __builtin_fma(e, e, e);                   // 1/d = y0 (1 + e + e^2)
  const double u = rowA * y0;
  const double t = __builtin_fma(u, f, u);                   // A[J][c] / d
  const double uw = rowW * y0;
  const double tw = __builtin_fma(uw, f, uw);                // W[J][c] / d
  const double tm = (c > J) ? t : 0.0;                       // columns <= J are parked: they keep the unscaled L column
  // registers whose four rows (g + 4r, g = 0..3) are all <= J hold finished rows: neither block is updated there
  constexpr int RMIN = (J + 1) >> 2;
  constexpr int WMASK = ((J & 3) != 3) ? (0xf & ~(1 << (J & 3))) : 0xf;
  double dn = 0.0, yn = 0.0;
  if constexpr (J == 0) F16_ASM(F16_ROW F16_UPD0);
  else if constexpr (J == 15) { }
  else if constexpr (RMIN == 0) F16_ASM(F16_PREV F16_ROW F16_UPD0);
  else if constexpr (RMIN == 1) F16_ASM(F16_PREV F16_ROW F16_UPD1);
  else if constexpr (RMIN == 2) F16_ASM(F16_PREV F16_ROW F16_UPD2);
  else F16_ASM(F16_PREV F16_ROW F16_UPD3);
  d = dn; y0 = yn;
  pRow = rowA; pt = t; ptw = tw;
  rowA = hA; rowW = hW;
  hA = fA; hW = fW;
}

