// Internal declarations shared by the libffgp translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ffgp.h"

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

#define FFGP_NB 128        // diagonal-block size of the blocked factorisation (one LDS-resident panel)
#define FFGP_MAX_STAGES 16 // hipEvent stage timers exposed through ffgp_last_timings

#define FFGP_HIP(call)                                                                        \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      fprintf(stderr, "[ffgp] HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return FFGP_ERR_HIP;                                                                    \
    }                                                                                         \
  } while (0)

#define FFGP_CHECK(call)            \
  do {                              \
    int s_ = (call);                \
    if (s_ != 0) return s_;         \
  } while (0)

#define FFGP_GRAD_LANES 4      // gradient lanes of ffgp_nlml_fused_batch (api.hip)
static inline int ffgp_round_up(int x, int m) { return (x + m - 1) / m * m; }

// radial profiles (include/ffgp.h FFGP_KFUN_*): value phi(s) and  -2 * dphi/ds  (the factor that turns G o K' into the
// W matrix of the length-scale gradient; for the squared exponential it equals phi itself)
__device__ __forceinline__ double ffgp_kfun_val(int kind, double rinv, double s) {
  if (kind == FFGP_KFUN_SE) return exp(-0.5 * s);
  if (kind == FFGP_KFUN_MATERN12) return exp(-sqrt(s) * rinv);
  if (kind == FFGP_KFUN_MATERN32) {
    const double a = sqrt(3.0 * s) * rinv;
    return (1.0 + a) * exp(-a);
  }
  if (kind == FFGP_KFUN_RQ) return pow(1.0 + 0.5 * s * rinv, -1.0 / rinv);   // rinv = 1/alpha
  const double a = sqrt(5.0 * s) * rinv;
  return (1.0 + a + (5.0 / 3.0) * s * rinv * rinv) * exp(-a);
}
__device__ __forceinline__ double ffgp_kfun_m2d(int kind, double rinv, double s) {
  if (kind == FFGP_KFUN_SE) return exp(-0.5 * s);
  if (kind == FFGP_KFUN_MATERN12) {
    const double r = sqrt(s);
    return rinv / r * exp(-r * rinv);
  }
  if (kind == FFGP_KFUN_MATERN32) return 3.0 * rinv * rinv * exp(-sqrt(3.0 * s) * rinv);
  if (kind == FFGP_KFUN_RQ) return pow(1.0 + 0.5 * s * rinv, -1.0 / rinv - 1.0);
  const double a = sqrt(5.0 * s) * rinv;
  return (5.0 / 3.0) * rinv * rinv * (1.0 + a) * exp(-a);
}
// d phi / d kparam for the profiles whose parameter is learnable in the reference (RQ's alpha, kernel.py:295);
// phi is the value already computed.  Matern's rho is a constructor constant there -> 0.
__device__ __forceinline__ double ffgp_kfun_dparam(int kind, double rinv, double s, double phi) {
  if (kind != FFGP_KFUN_RQ) return 0.0;
  const double u = 0.5 * s * rinv;
  return phi * (u / (1.0 + u) - log1p(u));
}

// exp(x) for the argument range of a covariance profile (x <= 0; correct for moderate positive x too): range reduction
// x = n ln 2 + r, |r| <= ln2 / 2, Taylor polynomial of degree 13 (truncation 4e-18 relative), v_ldexp_f64.  ~22 vector
// instructions with every constant in SGPRs -- the library exp() inlines to ~3x that, most of it constant moves and
// special-case handling this call site cannot reach.  Arguments below -745.2 return exactly 0 (as exp() does).
__attribute__((weak)) __constant__ double ffgp_exp_coef[14] = {1.0, 1.0, 0.5, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320, 1.0 / 362880,
                                         1.0 / 3628800, 1.0 / 39916800, 1.0 / 479001600, 1.0 / 6227020800.0};
struct ExpCoef {
  double c[14];
};
__device__ __forceinline__ void ffgp_exp_load(ExpCoef& e) {
#pragma unroll
  for (int i = 0; i < 14; ++i) e.c[i] = ffgp_exp_coef[i];   // uniform addresses: scalar loads into SGPR pairs
}
__device__ __forceinline__ double ffgp_exp_fast(double x, const ExpCoef& e) {
  x = fmax(x, -750.0);                                   // exp(-750) < 2^-1074: ldexp flushes it to exactly 0, like exp() does
  const double n = __builtin_rint(x * 1.4426950408889634);
  double r = __builtin_fma(n, -6.93147180369123816490e-01, x);
  r = __builtin_fma(n, -1.90821492927058770002e-10, r);
  double p = e.c[13];
#pragma unroll
  for (int i = 12; i >= 0; --i) p = __builtin_fma(p, r, e.c[i]);
  return __builtin_amdgcn_ldexp(p, (int)n);
}

// GEMM operand layouts.  "K-major": element (row, k) at P[row*ld + k]; "MN-major": at P[k*ld + row].
enum { OP_KMAJOR = 0, OP_MNMAJOR = 1 };
// tile scheduling modes: full rectangle / lower-triangular tiles of a symmetric update
enum { TILES_FULL = 0, TILES_LOWER = 1 };

struct GemmArgs {
  const double* A;  // op(A) is m x k
  const double* B;  // op(B) is k x n   (K-major B stores B^T as n x k)
  double* C;        // m x n row-major
  int m, n, k;
  int lda, ldb, ldc;
  double alpha, beta;  // C = alpha*op(A)op(B) + beta*C   (beta == 0 -> C not read)
  int tiles_m, tiles_n;
  int total_tiles;
  int grid;            // gridDim.x (= total_tiles, or split_at + 4 * (tiles handed out as quarters))
  int split_at;        // 128-tile launches: blocks >= split_at process 64 x 64 quarters of the tiles [split_at, all) (INT_MAX = none)
  int band_log2;       // tile order: bands of 2^band_log2 tile rows, column-major inside a band
  int pad_lds;         // bytes of unused dynamic LDS requested at launch (occupancy control)
  int fast;            // alpha = +-1, beta in {0,1}, offsets fit 32 bits: interior tiles take the scalar-addressed form
  int avec, bvec;      // 16-byte vector loads allowed for A / B
  int prio;            // raise the wave priority (look-ahead panel GEMMs)
  int batch;           // gridDim.y: identical problems at fixed element strides
  long sA, sB, sC;
  // triangular operands: restrict the k range of tile (ti, tj) to [max(lo_i*ti, lo_j*tj)*128, min(K, hi_i*(ti+1)*128, ...))
  int lo_i, lo_j, hi_i, hi_j;
  // outer batch (gridDim.z): the same launch for every block of a shared gradient stage, at per-operand element strides
  int batch2;
  long sA2, sB2, sC2;
  const double* pack;   // the panel in the MFMA operand layout (gemm_tile_direct), or null
  unsigned* pub_word;   // a look-ahead hand-off this launch publishes as it starts (everything before it on its stream is complete then:
  unsigned pub_val;     // potrf.hip, la_record_on_next_gemm), or null
};

// one member of a ragged GEMM launch (ffgp_gemm_f64_rag): everything of GemmArgs that differs between the members
#define FFGP_RAG_MAX 8
struct GemmRagMember {
  const double* A;
  const double* B;
  double* C;
  int m, n, k;
  int lda, ldb, ldc;
  int tiles_m, tiles_n, total_tiles, grid, split_at;
  int fast, avec, bvec;
};
struct GemmRag {
  GemmArgs base;        // what the members share: alpha, beta, prio, band_log2, pad_lds, triangular hints (none)
  GemmRagMember mem[FFGP_RAG_MAX];
};
// one operand set of a ragged launch as the caller describes it
struct GemmRagIn {
  const double* A; int lda;
  const double* B; int ldb;
  double* C; int ldc;
  int m, n, k;
};

int ffgp_live_handles();      // handles alive in this process (api.hip)

struct ffgp_handle {
  int device;
  hipStream_t stream;   // stream work is enqueued on (caller's, or `own`)
  hipStream_t own;      // the handle's own stream
  hipStream_t aux;      // high-priority side stream for the look-ahead panel factorisation
  hipEvent_t la_ev[10]; // look-ahead hand-off events ([7], [8]: carry mode's "strip Z(k) has run", main -> side stream)
  unsigned* ho_mem;     // one word (64 bytes apart) per look-ahead event: the hand-off as hipStreamWriteValue32 / hipStreamWaitValue32 (potrf.hip)
  unsigned ho_seq[10];  // ... the number its latest "record" wrote
  unsigned ho_launched[10];   // ... the newest number whose producing operation has been ENQUEUED (submission-order rule, la_wait)
  long ho_order_violations;   // waits enqueued before their producers (must stay 0)
  int lds_cap, lds_cap_known;   // hipDeviceAttributeMaxSharedMemoryPerBlock, asked for once (launch_diag)
  int diag_v4;          // option "diag_v4": the diagonal-block kernel with two barriers per stage (ffgp_potrf_diag128_v4) in place of v3
  int ho_gate;          // option "ho_gate" (default 1): waits are the library's own gate kernel with a watchdog; 0 = hipStreamWaitValue32
  int ho_timeout_ms;    // option "ho_timeout_ms" (default 2000): the gate gives up after this long (FFGP_ERR_HANDOFF)
  int* ho_info;         // the status word a gate that gives up writes to (set by the factorisation entry points)
  int ho_withhold;      // test hook, option "ho_withhold" = k: the k-th publication from now on is never written
  int ho_values;        // option "ho_values" (default 1): value hand-offs; 0 = the event pairs
  int ho_active;        // set per factorisation (la_begin): value hand-offs in use (not while a stream is being captured)
  int ho_selftest_failed;   // the cross-stream self-test timed out (serialised dispatches): value hand-offs stay off for this handle
  int ho_selftest_pending;  // create_resources tries the stream value operations once; a runtime without them keeps the event pairs
  int ho_defer;         // option "ho_defer" (default 1): the chain's "panel complete" word is written by the next diagonal-block kernel
  int diag_excl_rows;   // option "diag_excl_rows" (default 4096): carry iterations with at most this many trailing rows launch the panel's first diagonal block
                        // with a whole CU's LDS -- when this is the process's only handle (0 = never)
  int diag_excl_now;    // ... set around that launch
  int counted_live;     // this handle is counted in ffgp_live_handles()
  int ho_defer_slot;    // pending publication (-1 none) ...
  int ho_gdefer_slot;   // the same for the next GEMM launched on ho_gdefer_stream (the trailing update that follows S_bz on the main stream)
  hipStream_t ho_gdefer_stream;
  hipStream_t ho_defer_stream;  // ... for the next diagonal-block kernel launched on this stream
  int aux_prio;         // 1 = look-ahead chain kernels run at raised wave priority
  int force_ts;         // 0 = automatic GEMM tile shape, 32 / 64 / 128 = forced (benchmarks, tests)
  int small_tile_threshold;  // launches with fewer 128-tiles than this use the 64-tile
  int tile32_threshold;      // K-major launches with fewer 64-tiles than this use 32-row tiles
  int polite64_active;  // set while ffgp_potrf_impl's carry-form look-ahead of ONE block issues its launches
  int polite32_pad_kb;  // ... and by its 32-tile ones (default 46: two workgroups per CU and 36 KiB left for the chain's kernels)
  int polite64_pad_kb;  // unused LDS (KiB) requested by the look-ahead's 64-tile trailing updates (default 60: two workgroups per CU instead of four)
  int trsm128;          // 1 = the chain's full-block TRSM runs on its own kernel (ffgp_trsm128_kernel; same values as the general GEMM)
  int trsm128_max_m;    // ... for panels of at most this many rows (taller ones stay on the general GEMM)
  int diag_dbg;         // timing-only ablation mask of potrf_diag128 (0 in production)
  int diag_v2;          // diagonal-block kernel: 4 = round-4 kernel (default: owner-computes helpers, wave 0's SIMD partner steps aside), 1 = round-3 pipeline with the DP-ALU DPP pivot step, 3 = round-3 pipeline as it was, 0 = barrier version
  int la_split;         // 1 = issue the look-ahead part of the trailing update in two launches (first 128 columns first)
  int lookahead;        // 1 = overlap panel k+1 with the trailing update of step k
  int la_min_n;         // blocks up to this size are factored in order (no side stream): default 1024 since the look-ahead's hand-offs are values
                        // (2 - 6 % at 1280 ... 3584 rows; 3584 while they were event pairs)
  int la_carry;         // the panel's own update kernels also carry the next panel's first 128 columns (no S_a on the chain):
                        // 0 = never, 1 = always, 2 (default) = throughout for blocks of at most la_carry_n rows, and for larger blocks in
                        // the iterations whose trailing matrix has at most la_carry_rows rows
  int la_carry_n;       // ... (default 12288; also the largest member a ragged chain accepts)
  int la_carry_rows;    // ... (default 8192)
  int diag_attr_set;    // dynamic-LDS attribute of potrf_diag128 set on this handle's device
  int band_log2;        // GEMM tile order: band height 2^band_log2 tile rows (default 3)
  int split_rem_max;    // split tail of the 128-tile launches: quarter the last (tiles mod 256) tiles when that is <= this (0 = off)
  int polite_pad_kb;    // LDS padding (KiB) of a polite trailing-update workgroup
  int polite_m;         // trailing updates with fewer rows than this run one workgroup per CU (0 = never)
  bool own_stream;
  // workspace (grown on demand, never shrunk)
  double* ws;        // generic workspace
  size_t ws_bytes;
  double* dinv;      // inverses of the NB x NB diagonal blocks of the last factor: nblk * NB*NB
  size_t dinv_bytes;
  const double* dinv_L;  // factor the Dinv store currently belongs to (pointer, n, ld); nullptr = stale
  int dinv_n, dinv_ld;
  // inverses of the S x S diagonal super-blocks of the factor the Dinv store belongs to (built on demand by the
  // triangular sweeps: 8x fewer, 8x deeper steps than the 128-block sweep) + the sweep's output buffer
  double* sinv;
  size_t sinv_bytes;
  const double* sinv_L;
  int sinv_n, sinv_ld, sinv_S;
  double* tsw;       // n x nrhs staging of the super-block sweep
  size_t tsw_bytes;
  double* skw;       // split-K partial products
  size_t skw_bytes;
  double* ews;       // workspace of the symmetric eigensolver (syevd.hip)
  size_t ews_bytes;
  hipEvent_t eig_ev[12];   // hand-offs between the chase (side stream) and the back-transformation (main stream) of ffgp_syevd
  hipStream_t masked;   // CU-masked stream for the trailing updates of the chain-bound tail (tail_mask_m > 0)
  int masked_failed, tail_mask_m, tail_mask_cus;
  int chase_xl_max_n;
  int chase_xcc;        // the XCD this handle's XCD-local chases run on
  int chase_xl;         // bulge chase with every working wave on one XCD, hand-overs through that XCD's L2 (sb2st.hip)
  int syrk_direct;      // trailing update's interior tiles in the direct form (no LDS, no barriers; gemm_tile_direct)
  double* pack_buf;
  size_t pack_bytes;
  int syrk_h64;         // experiment: trailing update on 128 x 64 half tiles, three workgroups per CU
  hipStream_t aux3;     // fourth stream: the passenger rows of a look-ahead factorisation, one panel behind the chain (ffgp_potrf_impl)
  int pass_split_min;   // passenger rows (right-hand sides riding in the factorisation) from this many on leave the chain's launches; 0 = never
  hipStream_t aux2;     // third stream: the head of the triangular inverse under the factorisation's tail (nlml_fused_enqueue)
  hipEvent_t tri_ev[2]; // [0] factor columns < tri_hook_col are final (recorded by ffgp_potrf_impl on the side stream); [1] head done
  int tri_hook_col, tri_hook_fired;
  // batched factorisation (ffgp_nlml_fused_batch): F identical-shape blocks at fixed strides share ONE chain of launches -- every
  // kernel of ffgp_potrf_impl then covers all F blocks (diagonal-block kernel: one workgroup per block; GEMMs: gridDim.y = F)
  int bt_F;             // 0 / 1 = not batched
  long bt_sA, bt_sD;    // element strides between the blocks' workspaces / between their Dinv stores
  // outer batch of the shared chain's gradient stage: while ob_F > 1 EVERY ffgp_gemm_launch covers ob_F blocks (gridDim.z), the stride
  // of each operand looked up from the address range it points into; tile shapes are decided as for one block (same bits)
  int batch_grad_ob;    // option (default 1): the shared chain's gradient stage inverts all blocks in one outer-batched sequence of launches
  int ob_F;
  int ob_n;
  struct { const double* lo; const double* hi; long stride; } ob_rng[6];
  void* asm_collect;    // AsmCollector (assemble.hip): assemblies of a batch's small members parked for one multi-member launch
  int asm_collecting;
  int fold_info;        // ffgp_train_raw with one model: status-word upkeep lives in the Adam kernel
  int defer_info_copy;  // ffgp_train_raw's loop: the enqueue paths skip their per-call read-back of the status word
  double* train_g;      // ffgp_train_raw: gradients of the raw parameters [MAXF x GSTRIDE] + the step's losses [MAXF]
  // gradient lanes of ffgp_nlml_fused_batch: members of different sizes run their inverse / gradient stages side by side
  int grad_lanes;       // option "grad_lanes" (default 4; 1 = member after member)
  hipStream_t lane_st[FFGP_GRAD_LANES];   // [0] unused (lane 0 is the call's stream)
  hipEvent_t lane_ev[FFGP_GRAD_LANES];    // [0] fork, [z] lane z done
  double* lane_skw[FFGP_GRAD_LANES];      // per-lane split-K workspaces (swapped into h->skw while a member is enqueued on the lane)
  size_t lane_skw_bytes[FFGP_GRAD_LANES];
  double* lane_scal;    // per-lane scalar scratch (64 doubles each; swapped into h->d_scal)
  void* train_tab;      // ffgp_train_persist (train.hip): [models | bias corrections | status words] of the current call
  size_t train_tab_bytes;
  double* small_kbuf;   // ffgp_small_mfma_enqueue: the evaluate-mode launches' parked kernel values (8 models x 36 x 256 doubles)
  void* train_host;     // its pinned host mirror (staging of the table, read-back of the status words)
  int train_persist_off;  // option "train_persist" = 0: ffgp_train_raw never takes the one-launch trainer
  int* bt_info;         // [F] device status words (first non-positive pivot of each block)
  int* bt_info_host;    // pinned mirror
  int trtri_overlap;    // option (default 1)
  int trtri_fill;       // option "trtri_fill" (default 0): 1 = zero the whole inverse buffer before the head of the triangular inverse; 0 = only the diagonal blocks' upper parts; 2 = NaN-fill it (test)
  hipEvent_t ev_switch; // ffgp_set_stream: recorded on the stream the handle leaves, waited for by the one it moves to
  hipEvent_t sb_ev[4];  // sy2sb: hand-offs between the trailing update (main stream) and the next panel's QR chain (side stream)
  int sb_lookahead;     // option (default 0: measured 103 -> 108 ms at N = 8192 -- the event hand-offs cost more than the QR chain hides)
  int small_max_n;   // largest n that takes the one-kernel path (0 = the measured default, 40)
  int small_off;     // 1: never take the one-kernel path of small.hip (option "small_fused" = 0)
  double* d_link;    // effective parameters / their gradients of ffgp_nlml_fused_raw (2 x 256 doubles)
  int eig_overlap;   // ffgp_syevd: 1 = chase on the side stream with the Q2^T accumulation behind it (see syevd.hip), 0 = stage after stage
  int chase_pack;    // bulge chasing: every chase_pack-th workgroup works (8 = all on one XCD; 1 = spread over the chip)
  int splitk_min_k;  // thin products (<= 64 tiles of 64 x 64) with k >= this are cut along k (0 = never)
  int skinny_max_n;  // products with at most this many output columns (<= 8) take the matrix-vector kernels (0 = never)
  int super_block;   // S (multiple of 128, power-of-two multiple): 0 = sweeps always go block by block
  int super_min_n;   // factors smaller than this keep the 128-block sweep
  int* d_info;       // device status word(s)
  double* d_scal;    // small device scalar scratch (64 doubles)
  // launch-bound sizes: the raw-parameter likelihood call replayed as a captured graph (api.hip, nlml_fused_raw_enqueue)
  int sb_lower;                // option "sb_lower" (default 1): the band reduction keeps and reads only the LOWER triangle of the trailing matrix (sy2sb_av_sym) ...
  int sb_lower_min_n;          // ... for matrices of at least this many rows (option "sb_lower_min_n", default 6144: below, the full form is as fast or faster)
  int sb_sym_wg;               // option "sb_sym_wg" (default 2048): workgroups the lower-triangle A Y launch aims for (half of them exit: chunks right of the diagonal)
  int sb_av_gemm;              // option "sb_av_gemm" (default 0): 1 = the band reduction's A Y product on the general GEMM again
  int q2_blocks_lanes;         // how the last q2_prep wrote its blocks (1: lane order for q2_apply_wave4)
  int sb_qr4;                  // option "sb_qr4" (default 0): 1 = the band reduction's leaf QRs on 256 threads, four columns per half-wave (sy2sb_leaf_qr4)
  int q2_split_min_cols;       // option "q2_split_min_cols" (default 8192): from this many columns of Z on, Z <- Q2 Z runs 32-column slabs on eight waves
  int q2_wave4;                // option "q2_wave4" (default 1): Z <- Q2 Z with four sweep groups per pass over Z (sb2st.hip)
  int small2_off;              // option "small_finish" (default 0 = off): 1 = 40 < n <= 128 runs assembly + the blocked diagonal-block
                               // factorisation + ONE finishing kernel (7 launches instead of 21); measured +-5-10 % per training step
  unsigned long alloc_epoch;   // bumped whenever one of the handle's device buffers is re-allocated (captured pointers go stale)
  int raw_graph_max_n;         // option "raw_graph_max_n" (default 0 = never capture: measured no faster, see api.hip)
  struct RawGraph* rawg;
  int fwd_graph;               // option "fwd_graph" (default 0): forward-only calls replay a captured hipGraph (api.hip)
  struct RawGraph* fwdg;
  long graph_replays;
  double* d_asm;     // assembly on the matrix cores: shifted + scaled inputs and their squared norms (n (D + 1) doubles)
  size_t asm_bytes;
  int asm_mm;        // option "asm_mm" (default 1): interior squared-exponential tiles through the MFMA chain
  int asm_mm_grid;   // option "asm_mm_grid" (default 768 = 3 per CU): persistent workgroups of the matrix-core assembly
  int asm_mm_min;    // option "asm_mm_min" (default 6144): geometric-mean size below which the difference kernel runs alone
  int* h_info;       // pinned host mirror
  double* h_scal;    // pinned host mirror
  // timing
  int timing;        // 0 = off
  hipEvent_t ev[FFGP_MAX_STAGES + 1];
  float stage_ms[FFGP_MAX_STAGES];
  int n_stages;
  // roofline bookkeeping for the dominant kernel (trailing-update SYRK)
  double syrk_flops;
  double syrk_ms;
  long syrk_launches;
  hipEvent_t syrk_ev[2];
  std::vector<hipEvent_t> syrk_pool;  // (start, stop) pairs recorded around trailing-update launches (timing == 2)
  int syrk_pool_used;
  // tuning knobs
  int nb_outer;      // outer (trailing-update) block size, multiple of FFGP_NB
  int nb_big, nb_big_until;  // wider outer block while more than nb_big_until columns remain (0 = off)
  int use_naive;     // debug: route potrf through the naive kernels
};

// ---- gemm.hip
// tri: OR of TRI_* -- the k range of output tile (ti,tj) is clipped to the structurally non-zero part
enum { TRI_LO_I = 1, TRI_LO_J = 2, TRI_HI_I = 4, TRI_HI_J = 8 };
// alias: which operand (if any) shares its buffer with C -- the launcher then picks a tile shape that is race-free
enum { ALIAS_NONE = 0, ALIAS_A = 1, ALIAS_B = 2 };
int ffgp_gemm_launch(ffgp_handle* h, int opa, int opb, int mode, int syrk_tag, const double* A, int lda, const double* B,
                     int ldb, double* C, int ldc, int m, int n, int k, double alpha, double beta, int tri = 0,
                     int alias = ALIAS_NONE, int batch = 1, long sA = 0, long sB = 0, long sC = 0);
// several independent tiny launches of one kind as ONE launch (the per-member stages around a shared-chain batch)
#define FFGP_MULTI_MAX 8
struct ffgp_multi_red {
  const double* M[FFGP_MULTI_MAX];
  const double* L[FFGP_MULTI_MAX];
  double* out[FFGP_MULTI_MAX];
  double pi_const[FFGP_MULTI_MAX], scale[FFGP_MULTI_MAX];
  int rows[FFGP_MULTI_MAX], cols[FFGP_MULTI_MAX], ldm[FFGP_MULTI_MAX], n[FFGP_MULTI_MAX], ldl[FFGP_MULTI_MAX], d[FFGP_MULTI_MAX],
      blocks[FFGP_MULTI_MAX];
  int first;
};
struct ffgp_multi_tr {
  const double* src[FFGP_MULTI_MAX];
  double* dst[FFGP_MULTI_MAX];
  int rows[FFGP_MULTI_MAX], cols[FFGP_MULTI_MAX], lds[FFGP_MULTI_MAX], ldd[FFGP_MULTI_MAX];
};
#define FFGP_RED_BLOCKS 512
int ffgp_nll_reduce_multi(ffgp_handle* h, int F, const double* const* L, const int* n, const int* ldl, const double* const* M,
                          const int* rows, const int* cols, const int* ldm, const int* d, const double* pi_const, const double* scale,
                          double* const* out, double* partial_ws);
int ffgp_assemble_collect_begin(ffgp_handle* h);
int ffgp_assemble_collect_end(ffgp_handle* h);
void ffgp_assemble_collect_free(ffgp_handle* h);
int ffgp_transpose_multi(ffgp_handle* h, int F, const double* const* src, const int* rows, const int* cols, const int* ld_src,
                         double* const* dst, const int* ld_dst);

// K-steps-per-call training (ffgp_train_raw): the raw parameter storages of up to FFGP_TRAIN_MAXF models, by value to the Adam kernel
#define FFGP_TRAIN_MAXF 16
#define FFGP_TRAIN_GSTRIDE 160       // doubles per model in the gradient buffer: raw w (<= 128) | amp | diag_add
struct ffgp_train_slot {
  double* w[FFGP_TRAIN_MAXF];
  double* amp[FFGP_TRAIN_MAXF];
  double* dadd[FFGP_TRAIN_MAXF];
  int nw[FFGP_TRAIN_MAXF];
};

// one member of a ragged factorisation chain (ffgp_potrf_ragged): its (mtot x n) matrix [Sigma | passenger rows], its slice of the
// Dinv store (n / 128 blocks of 128 x 128, zero above the diagonal) and the index of its status word in ffgp_handle::bt_info
struct ffgp_rag_block {
  double* A;
  int n, mtot, lda;
  double* dinv;
  int info_index;
};
extern "C" int ffgp_ensure_aux2(ffgp_handle* h);
int ffgp_potrf_ragged(ffgp_handle* h, int R, const ffgp_rag_block* mem);
int ffgp_handoff_selftest(ffgp_handle* h);      // 0: a gate on one stream sees a value written from another; 1: it timed out; < 0: error
// ragged form of the chain's K-major products: R members of one kind with their own sizes / operands (see gemm.hip)
int ffgp_gemm_launch_rag(ffgp_handle* h, int mode, int syrk_tag, int R, const GemmRagIn* in, double alpha, double beta, int alias);
// ---- potrf.hip
int ffgp_potrf_impl(ffgp_handle* h, double* A, int n, int mtot, int lda, int sync_info);
int ffgp_ensure_dinv(ffgp_handle* h, int n);
int ffgp_refresh_dinv(ffgp_handle* h, const double* L, int n, int ldl);
int ffgp_map_info(int v);
// ---- solve.hip
int ffgp_trsm_lower_impl(ffgp_handle* h, const double* L, int n, int ldl, double* B, int nrhs, int ldb);
int ffgp_trsm_lower_t_impl(ffgp_handle* h, const double* L, int n, int ldl, double* B, int nrhs, int ldb);
int ffgp_trtri_head(ffgp_handle* h, const double* L, int n, int ldl, double* X, int ldx, double* T, double* Ttop, int n1);
int ffgp_trtri_tail(ffgp_handle* h, const double* L, int n, int ldl, double* X, int ldx, double* T, const double* Ttop, int n1);
// ---- small.hip
bool ffgp_small_ok(const ffgp_handle* h, const ffgp_problem* p, const ffgp_grads* g);
int ffgp_small_enqueue(ffgp_handle* h, const ffgp_problem* p, const ffgp_links* l, double* nll_dev, const ffgp_grads* g,
                       const double* dinv = nullptr);
bool ffgp_small2_ok(const ffgp_handle* h, const ffgp_problem* p, const ffgp_grads* g);
bool ffgp_small_batch_ok(const ffgp_problem* p, const ffgp_grads* g);
// train.hip: K Adam steps of F small models in ONE launch (one persistent workgroup per model)
bool ffgp_train_persist_ok(const ffgp_handle* h, const ffgp_problem* p, const ffgp_links* l);
bool ffgp_small_mfma_ok(const ffgp_handle* h, const ffgp_problem* p, const ffgp_grads* g);      // one likelihood (+ gradients), n <= 128: same kernel, no Adam
int ffgp_small_mfma_enqueue(ffgp_handle* h, int F, const ffgp_problem* p, const ffgp_links* l, double* nll_dev, const ffgp_grads* g, int info_max);
int ffgp_train_persist(ffgp_handle* h, int F, const ffgp_problem* p, const ffgp_links* l, int steps, const ffgp_adam* opt, double* state_dev,
                       long state_stride, long step0, double* trace_dev, long trace_stride);
int ffgp_small_batch_enqueue(ffgp_handle* h, int F, const ffgp_problem* p, const ffgp_links* l, double* nll_dev, const ffgp_grads* g);
// ---- workspace
int ffgp_ensure_ws(ffgp_handle* h, size_t bytes);
// zero `bytes` (a multiple of 4) on the handle's stream with a kernel: small fills on the captured (graph) path go through this
// instead of hipMemsetAsync, whose graph nodes replayed wrong values on this ROCm build; large ones keep the runtime's fill
int ffgp_zero_async(ffgp_handle* h, void* ptr, size_t bytes);
