/*
 * libffgp -- MI355X (gfx950) native Gaussian-process hot path: C ABI.
 *
 * Drop-in boundary for the GP computation path of IceLab-X/FidelityFusion.  The reference has no FFI of its
 * own (it is pure Python on torch); each entry point below names the reference call it replaces
 * (file:line relative to the reference root).  The Python shim (fidelityfusion_amd/_lib.py, ctypes) is the
 * reference-side binding; INTEGRATION.md shows how the reference's modules would call it.
 *
 * Conventions
 *   - every pointer named *_dev is a DEVICE pointer to row-major (C-contiguous) fp64 data;
 *   - every call returns int: 0 = ok; > 0 = 1-based index of the first non-positive pivot (the matrix is not
 *     positive definite -- torch.linalg.cholesky raises torch.linalg.LinAlgError there); < 0 = FFGP_ERR_*;
 *   - work is enqueued on the handle's stream (ffgp_set_stream); calls that return a status derived from
 *     device data (ffgp_potrf, ffgp_nlml_fused, ffgp_predict) synchronise that stream before returning;
 *   - nothing here falls back to the CPU: without a gfx950 device ffgp_create fails;
 *   - THREADING: a handle owns mutable state (its stream binding, workspaces, the cached inverses) that calls modify while
 *     they run; at most ONE host thread may be inside a call on a given handle at any time.  Use one handle per thread
 *     (handles are cheap: two streams + workspaces grown on demand), or serialise calls on a shared one.
 *
 * Parameterisation.  All of the reference's stationary kernels on this path are
 *        K_ij = amp * exp(-1/2 * max(sum_k ((x_ik - x_jk) * w_k)^2, clamp_min))
 *   K1 ARDKernel               (GaussianProcess/kernel.py:88-105):   w = 1/(|length_scales|+1e-9), amp = |signal_variance|, clamp 1e-30
 *   K2 SquaredExponentialKernel(GaussianProcess/kernel.py:258-272):  w = exp(-length_scale) (all D), amp = exp(signal_variance)^2, clamp -inf
 *   K3 SE_kernel 2023          (MFGP_ver2023May/kernel/SE_kernel.py:20-44): w = 1/length_scale (or exp form), amp = scale
 * and every covariance on the path is
 *        Sigma = K + diag_add*I + diag(diag_vec) + add_mat + add_all*11^T + mean_jitter*mean(K)*I
 *   S1 cigp_v10.py:57-60           diag_add = exp(-log_beta)+1e-6, diag_vec = diag(y_var)
 *   S2 gp_computation_pack.py:125  diag_add = exp(-log_beta), mean_jitter = 1e-6
 *   S3 gp_basic.py:63-65,117-119   diag_add = noise_variance^2, add_mat = y_var
 *   S4 MFGP_ver2023May/base_gp/cigp.py:124-127  diag_add = 1e-6 + 1/noise, add_all = y_var
 * The raw->effective maps (abs, exp, pow) are O(D) and stay in the host language (torch autograd chains
 * through them); the ABI takes the effective w/amp/diag_add as DEVICE scalars so no call forces a D2H copy
 * of a parameter, and returns gradients with respect to those effective quantities.
 */
#ifndef FFGP_H
#define FFGP_H

#ifdef __cplusplus
extern "C" {
#endif

/* libffgp.so is linked with -fvisibility=hidden: exactly the functions declared in this header are exported
   (tests/test_abi_exports.py holds `nm -D` to this list, in both directions). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

typedef struct ffgp_handle ffgp_handle;

enum {
  FFGP_OK = 0,
  FFGP_ERR_ARG = -1,    /* bad argument (null pointer, negative size, misaligned leading dimension) */
  FFGP_ERR_HIP = -2,    /* a HIP runtime call failed (message on stderr) */
  FFGP_ERR_ALLOC = -3,  /* device workspace allocation failed */
  FFGP_ERR_NODEVICE = -4,
  FFGP_ERR_HANDOFF = -5 /* a cross-stream hand-off of the factorisation's look-ahead never arrived and its gate gave up after option
                           "ho_timeout_ms" (a tool that serialises this process's kernels); the call's results are invalid */
};

/* likelihood formula variants */
enum {
  FFGP_LL_V1 = 1, /* nll = 1/2||L^-1 Y||^2 + d*sum(log L_ii) + 1/2*N*d*log(2*pi_const)
                     cigp_v10.py:61-69; gp_computation_pack.py:128-136; base_gp/cigp.py:129-136 (pi_const = 3.1415) */
  FFGP_LL_V2 = 2  /* -LL = 1/2(||Sigma^-1 Y||^2 + 2d*sum(log L_ii) + N*d*log(2*pi))  (the Sigma^-2 form of
                     Gaussian_log_likelihood 'cholesky3', gp_computation_pack.py:65-80; gp_basic.py:130-143) */
};

/* radial profile of the stationary kernel, K = amp * phi(s), s = max(||(x - x') o w||^2, clamp_min) */
enum {
  FFGP_KFUN_SE = 0,        /* phi = exp(-s/2)                                                 (K1, K2, K3) */
  FFGP_KFUN_MATERN12 = 1,  /* phi = exp(-sqrt(s)/rho)                       MaternKernel nu = 0.5, GaussianProcess/kernel.py:161-162 */
  FFGP_KFUN_MATERN32 = 2,  /* phi = (1 + a) exp(-a),          a = sqrt(3s)/rho           nu = 1.5, kernel.py:163-164 */
  FFGP_KFUN_MATERN52 = 3,  /* phi = (1 + a + a^2/3) exp(-a),  a = sqrt(5s)/rho           nu = 2.5, kernel.py:165-166 */
  FFGP_KFUN_RQ = 4,        /* phi = (1 + s/(2 alpha))^(-alpha), kparam = alpha (learnable: g_kparam)  RationalQuadraticKernel, kernel.py:297-310 */
  FFGP_KFUN_LINEAR = 5     /* not a radial profile: K = amp * sum_k w_k^2 (x_k - c_k)(x'_k - c_k)   LinearKernel, kernel.py:22-63.
                              Only valid inside an ffgp_kdesc (the two-descriptor entry points below) */
};

/* One part of a composed kernel -- SumKernel / ProductKernel, GaussianProcess/kernel.py:172-236; the reference's demos and
   two-fidelity models all run on SumKernel(LinearKernel, MaternKernel) (cigp_v10.py:81; two_fidelity_models/ResGP.py:25,
   AR_autoRegression.py:31, NAR_NonlinearAR.py:23).  Two descriptors + an operator are evaluated in ONE tile pass. */
typedef struct {
  int kfun;                  /* FFGP_KFUN_*  (FFGP_KFUN_LINEAR allowed) */
  const double* w_dev;       /* [D] inverse length scales */
  const double* amp_dev;     /* [1] amplitude */
  double clamp_min;          /* lower clamp on the squared distance (stationary parts) */
  double kparam;             /* rho / alpha of the profile */
  const double* center_dev;  /* [D] LinearKernel.center (FFGP_KFUN_LINEAR only; NULL = origin) */
} ffgp_kdesc;
typedef struct {             /* gradients w.r.t. one part's effective quantities; any pointer may be NULL */
  double* g_w_dev;           /* [D] */
  double* g_amp_dev;         /* [1] */
  double* g_kparam_dev;      /* [1] (FFGP_KFUN_RQ) */
  double* g_center_dev;      /* [D] (FFGP_KFUN_LINEAR) */
} ffgp_kdesc_grads;
enum { FFGP_KOP_SUM = 0, FFGP_KOP_PRODUCT = 1 };

/* A nested composition -- SumKernel / ProductKernel objects whose parts are themselves Sum / Product kernels (kernel.py:172-236
   compose arbitrary modules) -- with up to four leaves, evaluated in the same single tile pass.  Every binary tree with <= 4
   leaves is, up to the operand order of its (commutative, hence bit-identical) nodes, one of
       n_leaves = 2:  l0 op[0] l1
       n_leaves = 3:  (l0 op[0] l1) op[1] l2
       n_leaves = 4:  FFGP_TREE_CHAIN     ((l0 op[0] l1) op[1] l2) op[2] l3
                      FFGP_TREE_BALANCED  (l0 op[0] l1) op[2] (l2 op[1] l3)                                              */
enum { FFGP_TREE_CHAIN = 0, FFGP_TREE_BALANCED = 1 };
typedef struct {
  int n_leaves;              /* 2, 3 or 4 */
  int shape;                 /* FFGP_TREE_* (n_leaves = 4 only) */
  int op[3];                 /* FFGP_KOP_* of the n_leaves - 1 nodes, as numbered above */
  const ffgp_kdesc* leaf;    /* [n_leaves] */
} ffgp_ktree;

/* prediction outputs */
enum {
  FFGP_VAR_FULL = 0, /* cov[Nt,Nt] = K** - V^T V + var_add_all              (cigp_v10.py:41-44; gp_computation_pack.py:108-110) */
  FFGP_VAR_DIAG = 1  /* var[Nt]    = diag(K**) - colsum(V^2) + var_add_all  (base_gp/cigp.py:91-94) */
};

/* One GP block: inputs, targets, effective kernel/noise quantities.  All pointers are device pointers. */
typedef struct {
  int n;                  /* training points */
  int D;                  /* input dimension */
  int d;                  /* output columns */
  const double* X_dev;    /* [n, D] */
  const double* Y_dev;    /* [n, d] */
  const double* w_dev;    /* [D]  inverse length scales */
  const double* amp_dev;  /* [1]  amplitude */
  double clamp_min;       /* lower clamp on the squared distance (1e-30 for K1, -inf for K2/K3) */
  const double* diag_add_dev; /* [1] scalar added to the diagonal */
  const double* diag_vec_dev; /* optional: entry i at diag_vec_dev[i*diag_stride] (pass the N x N y_var with stride N+1) */
  long diag_stride;
  const double* add_mat_dev;  /* optional full [n, n] matrix added to Sigma (only its lower triangle is read) */
  int ld_add;
  double add_all;         /* scalar added to every entry */
  double mean_jitter;     /* coefficient of mean(K)*I (1e-6 for gp_computation_pack.negative_log_likelihood, else 0) */
  int ll_variant;         /* FFGP_LL_V1 | FFGP_LL_V2 */
  double pi_const;        /* 3.1415 for the V1 call sites, M_PI for V2 */
  int kfun;               /* FFGP_KFUN_* (0 = squared exponential) */
  double kparam;          /* rho of the Matern profiles (MaternKernel(rho=...), default 1); unused for SE */
  const double* cov_dev;  /* optional: a caller-built covariance [n, n] (lower triangle read).  When set, nothing is
                             assembled (X, w, amp, the diag and add fields are ignored) -- the Gaussian_log_likelihood(y, cov) call
                             shape of gp_computation_pack.py:34-91 -- and the gradient comes back through g_cov_dev */
  int ld_cov;
  const ffgp_kdesc* pair; /* optional: [2] descriptors of a composed kernel K = k[0] (+|x) k[1] on X_dev.  When set, w_dev / amp_dev /
                             clamp_min / kfun / kparam above are ignored and the kernel gradients come back through
                             ffgp_grads.g_pair; every Sigma extra (diag / matrix / all-entries / mean jitter) applies as usual.
                             Not accepted by ffgp_predict (the modules compose the posterior from ffgp_assemble_pair pieces). */
  int pair_op;            /* FFGP_KOP_SUM | FFGP_KOP_PRODUCT */
  const ffgp_ktree* tree; /* optional: a nested composition of 2-4 leaves in place of `pair` (same rules; ffgp_grads.g_pair then holds
                             n_leaves entries) */
} ffgp_problem;

/* Gradients of the value returned by ffgp_nlml_fused with respect to the effective quantities.
   Any pointer may be NULL (that gradient is skipped); all are device pointers. */
typedef struct {
  double* g_w_dev;        /* [D] */
  double* g_amp_dev;      /* [1] */
  double* g_diag_add_dev; /* [1]  (= tr G, plus the mean-jitter chain for S2) */
  double* g_Y_dev;        /* [n, d] */
  double* g_diag_vec_dev; /* [n]  (= diag G), optional */
  double* g_cov_dev;      /* [n, n] full symmetric d(value)/d(cov) (what torch's cholesky backward returns), optional */
  int ld_gcov;
  double* g_kparam_dev;   /* [1] d(value)/d(kparam) for FFGP_KFUN_RQ (alpha is an nn.Parameter, kernel.py:295), optional */
  const ffgp_kdesc_grads* g_pair; /* [2] gradients of the two parts when ffgp_problem.pair is set ([n_leaves] for .tree), optional */
} ffgp_grads;

/* ---- lifetime ------------------------------------------------------------------------------------------ */
int ffgp_create(int device, ffgp_handle** out);
int ffgp_destroy(ffgp_handle* h);
int ffgp_set_stream(ffgp_handle* h, void* hip_stream); /* hipStream_t; NULL restores the handle's own stream.  When the stream
                                                           changes, the new one is made to wait (event) for the work this handle
                                                           enqueued on the old one: the handle's workspaces are shared by both */
/* options: "timing" (0/1: record per-stage hipEvents; 2: also an event pair around every trailing-update launch),
            "nb_outer" (trailing-update block, multiple of 128; default 512),
            "naive" (1: route factor kernels through the slow reference kernels; debugging only),
            "lookahead" (default 1: factor panel k+1 on a high-priority side stream under the trailing update of step k),
            "la_min_n" (default 1024: smaller blocks are factored in order on one stream.  With event pairs between the two streams the
                        overlap did not pay for its hand-offs below ~3500 rows; with value hand-offs -- "ho_values" -- it does from two
                        panels on: N = 1280 / 1536 / 2048 / 2560 / 3072 / 3584 -3.4 / -2.1 / -4.3 / -4.2 / -5.0 / -6.1 %; 0 = look ahead at
                        every size),
            "la_split" (default 1: the look-ahead column update covers the next panel's first 128 columns only),
            "la_carry" / "la_carry_n" / "la_carry_rows" (a panel's own update kernels also cover the next panel's first 128 columns, so
                        no strip update sits between two panels on the dependency chain: 1 = always, 0 = never (the S_a / S_b / S_ii
                        form), default 2 = throughout for blocks of at most la_carry_n rows (default 12288), and for larger blocks in
                        the ITERATIONS whose trailing matrix has at most la_carry_rows rows (default 8192): such a block starts in the
                        S_a / S_b / S_ii form -- there the chain hides under the update and the carry only widens its launches -- and
                        changes over.  Headline bench line, la_carry_rows = 0 / 6144 / 8192 / 10240 / 12288: 28.83 / 28.61 / 28.51 / 28.47 /
                        28.43 ms with the roofline kernel at 0.697 / 0.697 / 0.693 / 0.690 / 0.685 of peak beside the heavier chain),
            "ho_values" / "ho_defer" (default 1 / 2: the look-ahead's hand-offs between its two streams are values in device memory --
                        hipStreamWriteValue32 behind the producer, hipStreamWaitValue32 in front of the consumer, 2.9-4.7 us per hop
                        against 10.7-11.1 for hipEventRecord + hipStreamWaitEvent on this runtime -- and the word that says "panel k is
                        complete" is written by the next panel's first diagonal-block kernel as it starts instead of by a 5 us write
                        kernel on the chain: N = 4096 / 8192 / 12288 -5.3 / -2.4 / -1.7 %, values unchanged (ho_defer >= 1); with 2 the
                        hand-off of S_bz is likewise written by the S_ii launch behind it on the update stream: N = 16384 -0.4 %; ho_values = 0: the event
                        pairs, which are also what a stream under graph capture gets.  A value wait is a one-workgroup KERNEL that
                        polls the word, so a process whose kernels run strictly one at a time must not use it: ffgp_create starts a
                        handle with ho_values = 0 when it sees rocprofv3's counter collection (ROCPROF_COUNTER_COLLECTION: --pmc
                        serialises the dispatches of all queues), thread trace, the rocprofiler v1 / v2 tool libraries,
                        HIP_LAUNCH_BLOCKING or AMD_SERIALIZE_KERNEL in the environment; FFGP_HANDOFF=events / values overrides),
            "aux_prio" (default 1: raised wave priority for the side stream's kernels),
            "gemm_tile" (0 = automatic; 32 / 64 / 128 force the GEMM tile shape -- tests and benchmarks),
            "small_tile_threshold" (default 640: launches with fewer 128-tiles use 64-tiles),
            "tile32_threshold" (default 1024: K-major launches with fewer 64-tiles use 32-row tiles),
            "polite_m" (default 6144: trailing updates with fewer rows run one workgroup per CU so that the side
                        stream's kernels always find free registers and LDS),
            "polite_pad_kb" (default 40: the LDS padding of a polite workgroup; 17 would leave room for the diagonal-block kernel
                        beside it -- measured neutral),
            "split_rem_max" (default 180: a 128-tile launch whose tile count leaves a remainder <= this modulo the 256 CUs
                        hands those last tiles out as 64 x 64 quarters -- same bits, a shorter last round; 0 = never),
            "nb_big" / "nb_big_until" (default 0: a wider outer block while more than nb_big_until columns remain; measured
                        neutral at N = 16384 -- tools/ab_forward.py),
            "super_block" / "super_min_n" (default 1024 / 2048: triangular sweeps on factors of at least super_min_n rows go
                        through inverted super_block x super_block diagonal blocks; 0 = always block by block),
            "skinny_max_n" (default 8: products with at most this many output columns run on the matrix-vector kernels; 0 = never),
            "splitk_min_k" (default 1024: products with <= 64 tiles of 64 x 64 and k >= this are cut along k; 0 = never),
            "band_log2" (default 3: the GEMM tile order walks bands of 2^k tile rows, column-major inside a band),
            "diag_dbg" (timing-only ablation mask of the diagonal-block kernel; results are wrong when non-zero),
            "asm_mm" / "asm_mm_min" / "asm_mm_grid" (default 1 / 6144 / 768: squared-exponential assemblies whose geometric-mean size
                        is at least asm_mm_min evaluate their interior 64 x 64 tiles on the matrix cores -- norm expansion, guarded
                        per 32 x 32 block by a fall-back to the difference form wherever a distance is below 1e-6 of the
                        squared norms -- with asm_mm_grid persistent workgroups; 0 = the difference kernel alone),
            "q2_wave4" (default 1: ffgp_syevd / ffgp_ormq2 apply Q2 with four sweep groups per pass over Z; 2 = the same on
                        32-column slabs; 0 = one group per pass -- blocks are prepared in the matching layout by ffgp_sb2st),
            "small_finish" (default 0; 1: 40 < n <= 128 runs assembly + blocked factorisation + ONE finishing kernel, 7 launches
                        instead of 21 -- measured +-5-10 % per training step),
            "raw_graph_max_n" (default 0; > 0: ffgp_nlml_fused_raw calls with n <= this are captured into a hipGraph on their
                        second identical occurrence and replayed afterwards -- measured no faster on ROCm 7.2),
            "trtri_fill" (default 0: the gradient path's triangular inverse, when its head runs under the factorisation, zeroes only the
                        diagonal blocks' upper parts of its N x N buffer -- every consumer reads it tile-wise below the diagonal;
                        1 = zero-fill the whole buffer first (2 GB per step at N = 16384; +0.2 ms); 2 = fill it with NaN, a test mode),
            "sb_av_gemm" (default 0: the band reduction forms A * Y with its own 128-row kernel; 1 = the general GEMM -- measured
                        sy2sb 104 -> 91 ms at n = 8192, equal below n = 4096),
            "sb_lower" (default 1) / "sb_lower_min_n" (default 6144): for matrices of at least that many rows ffgp_syevd's band
                        reduction keeps only the LOWER triangle of the trailing matrix up to date (the rank-64 update is bound by
                        HBM: half the bytes) and forms A * Y from that triangle alone (every element serves the product and its
                        transpose); results agree with the full form to rounding.  n = 8192: update 21.8 -> 12.3 ms, A * Y and its
                        sums 16.1 -> 20.6 ms, eigh 234.3 -> 229.7 ms; below ~6000 rows the full form is as fast.  "sb_sym_wg"
                        (default 2048): workgroups the lower-triangle A * Y launch aims for,
            "sb_qr4" (default 0; 1: the band reduction's leaf QRs on 256-thread workgroups, four columns per half-wave -- 58.9 us per
                        panel against 54.4 us on 1024 threads at n = 8192; with "sb_lookahead" the only form whose leaves overlap
                        the trailing update, stage time equal either way),
            "diag_v2" (default 4: the round-4 diagonal-block kernel ffgp_potrf_diag128_v3 -- owner-computes helper waves, no
                        barrier; 0 = the barrier version (ffgp_nlml_fused_batch then returns FFGP_ERR_ARG: only the default kernel
                        is batched).  The shipped library accepts 0 and 4 only; 1 and 3, the round-3 pipelines, exist in the
                        development build),
            "chase_pack" (placement of the bulge chase's 256 working wavefronts: every pack-th workgroup works.  Default 1 = one per
                        compute unit over the whole chip (N = 8192: 64.6 ms; 2 = on every other XCD: 69.4, 4: 92, 8: 166); beside
                        other blocks' kernels 1 and 2 measure the same (config 5's eight blocks: 1.43-1.47 s per step either way),
            "fwd_graph" (default 0; 1 = forward-only likelihood calls of more than one diagonal block replay a captured hipGraph from
                        their second identical occurrence on: see ffgp_graph_replays -- measured, no gain on this runtime),
            "q2_split_min_cols" (default 8192: from this many columns of Z on, the eigensolver's back-transformation Z <- Q2 Z runs 32-column
                        slabs on eight waves -- ormq2 45.4 -> 38.9 ms at n = 8192, 434 -> 297 at 16384, values identical; below, the
                        slabs would not fill the chip),
            "polite64_pad_kb" (default 60, at most 64: unused dynamic LDS requested by the 64-tile trailing updates of ONE block's carry-form
                        look-ahead -- two of their workgroups per CU instead of four, so the chain's kernels are not slowed to a third
                        beside them; N = 4096 / 6144 / 8192 / 12288: -1.2 / -3.0 / -2.0 / -1.0 %, values unchanged; 0 = off),
            "polite32_pad_kb" (default 46: the same for that form's 32-tile trailing updates -- two workgroups per CU and room for the
                        chain's TRSM beside them: N = 3072 / 4096 / 5120 -1.0 / -0.8 / -0.5 % against 34 (three per CU), which had
                        been -0.8 ... -1.5 % against none; 0 = off),
            "diag_excl_rows" (default 4096: in the look-ahead's chain-bound iterations -- at most this many trailing rows -- a panel's first
                        diagonal-block kernel asks for a whole CU's LDS, so that the trailing-update workgroups it releases cannot land
                        beside it: N = 2048 / 4096 / 8192 -1.7 / -0.8 / -0.8 %, values unchanged; only while the handle is the process's
                        only one -- beside other handles' kernels an empty CU may be long in coming; 0 = never),
            "trsm128" / "trsm128_max_m" (default 1 / 8192: the factorisation chain's full-block TRSM runs on its own latency-shaped
                        kernel for panels of at most trsm128_max_m rows; the values are the general GEMM's bit for bit),
            "chase_xl" / "chase_xl_max_n" (default 1 / 2048: bands of up to chase_xl_max_n columns run their bulge chase with every
                        working wavefront on ONE XCD -- the kernel reads HW_REG_XCC_ID and the others leave -- and hand the band over
                        through that XCD's L2: plain stores, L1-bypassing loads; sb2st 7.9 -> 6.5 ms at n = 1024, 15.9 -> 13.8 at 2048;
                        0 = the chip-wide form with device-scope accesses at every size.  Where the runtime places a wave is observed
                        behaviour, not a contract: every XCD-local launch is followed by a chip-wide launch that chases whatever sweeps
                        the first did not hand out -- none on a full MI355X; all of them when no wave landed on the chosen XCD, as in a
                        partition mode or on a CU-masked stream),
            "diag_v4" (default 1: the factorisation's 128 x 128 diagonal blocks on ffgp_potrf_diag128_v4 -- two workgroup barriers per
                        16-column stage, the inverse's rows formed in the shadow of the next block's pivots: 25.2 us per block against
                        28.6 for the flag-driven pipeline of round 4 (v3, = 0), forward N = 1024 / 4096 / 8192 -6.9 / -5.7 / -2.0 %;
                        the factors differ in the last bits (the updates reach a block in another order)),
            "ho_gate" / "ho_timeout_ms" (default 1 / 2000: the look-ahead's cross-stream waits are the library's own one-wave gate kernel,
                        which gives up after ho_timeout_ms without its value -- the call then returns FFGP_ERR_HANDOFF instead of hanging the
                        GPU, e.g. under a tool that runs this process's kernels one at a time and is not recognised at create time;
                        0 = hipStreamWaitValue32, which has no timeout),
            "ho_withhold" (test hook: k > 0 makes the k-th hand-off publication from now on never happen),
            "grad_lanes" (default 3: in ffgp_nlml_fused_batch, members of DIFFERENT sizes (all <= 6144 rows) run their inverse / gradient
                        stages side by side on up to three of the handle's streams, each with its own scratch; every member's launch
                        sequence is its single call's, so are its bits; 1 = member after member.  Measured -3 ... -6 %),
            "train_persist" (default 1: ffgp_train_raw runs sets of small models -- n <= 128, D, d <= 16 -- as ONE persistent kernel launch,
                        see ffgp_train_raw; 0 = one launch per stage and step, the round-5 form),
            "chase_xcc" (0..15; default: handles of one process take XCDs 0..7 in turn -- the XCD whose wavefronts run the XCD-local
                        chase; a value no wave of the launch reports leaves the whole chase to the chip-wide launch behind it: a test mode),
            "batch_grad_ob" (default 1: the shared chain's gradient stage inverts all blocks in one outer-batched sequence of launches),
            "trtri_overlap", "small_fused", "small_max_n" (round-3 experiment switches).
   Keys the SHIPPED library refuses with FFGP_ERR_ARG (they are accepted by the development build only, `make dev`,
   ffgp_has_dev_options() == 1): "raw_graph_max_n", "diag_dbg", "la_split", "nb_big", "nb_big_until", "sb_lookahead", "sb_av_gemm",
   "sb_qr4", "q2_wave4", "eig_overlap", "band_log2", "polite_pad_kb", "pass_split_min", "tail_mask_m", "tail_mask_cus", "syrk_h64", "syrk_direct" (the
   round-5 experiments on the factorisation's chain, docs/experiments.md), and every value of "diag_v2" other than 0 and 4.  An unknown
   key is FFGP_ERR_ARG in both builds.    */
int ffgp_set_option(ffgp_handle* h, const char* key, double value);
/* Create the handle's side streams now and use each once, so that they bind their hardware queues before streams the process creates
   later (ROCm binds at first use; a late stream shares a queue with an earlier one and runs in line with it).  For the main handle of
   a process that puts several blocks in flight on one GPU: call it before creating the worker streams.  No reference counterpart. */
int ffgp_prepare_streams(ffgp_handle* h);
const char* ffgp_version(void);
/* 1 in the development build (`make dev`: the switches of measured-and-rejected experiments are accepted by ffgp_set_option), 0 in
   the shipped library */
int ffgp_has_dev_options(void);
/* Number of forward calls this handle has served by replaying a captured graph since it was created (option "fwd_graph" of
   ffgp_set_option: the second identical forward-only call -- same problem struct, so the same device buffers; their CONTENTS may
   change -- is captured with both of its streams, later ones are one hipGraphLaunch and a one-word copy: same kernels, same values,
   the status of a failing pivot reported as by the plain call).  Measured on ROCm 7.2 (tools/host_load_probe.py): no gain -- this
   runtime executes a graph's nodes from a host thread one by one, so the step is 0.2-1 % slower at N = 16384, 22 % slower at
   N = 4096, and as exposed to a starved host as the plain launches; the option is off by default.  -1 for a NULL handle.  No reference
   counterpart. */
long ffgp_graph_replays(const ffgp_handle* h);

/* ---- building blocks ------------------------------------------------------------------------------------ */
/* K(x1,x2) [+ Sigma extras when the matrix is square and symmetric].  Replaces kernel.forward
   (GaussianProcess/kernel.py:88-105,258-272; MFGP_ver2023May/kernel/SE_kernel.py:20-44) and the torch.eye
   adds at cigp_v10.py:31-32,57-60; gp_computation_pack.py:125-126; gp_basic.py:63-65; base_gp/cigp.py:79-81,124-127.
   lower_only != 0: only tiles on/below the diagonal are written (n1 must equal n2).                      */
int ffgp_assemble(ffgp_handle* h, const double* X1_dev, int n1, const double* X2_dev, int n2, int D,
                  const double* w_dev, const double* amp_dev, double clamp_min, const double* diag_add_dev,
                  const double* diag_vec_dev, long diag_stride, const double* add_mat_dev, int ld_add,
                  double add_all, double mean_jitter, double* K_dev, int ldk, int lower_only, int kfun, double kparam);

/* The same for a composed kernel K = k[0] (+|x) k[1]  (k: two descriptors, op: FFGP_KOP_*): one write-only pass instead of
   the two kernel evaluations + elementwise combine + torch.eye adds of the reference (kernel.py:172-236 under
   cigp_v10.py:57-60).  Sigma extras as in ffgp_assemble.                                                        */
int ffgp_assemble_pair(ffgp_handle* h, const double* X1_dev, int n1, const double* X2_dev, int n2, int D, const ffgp_kdesc* k,
                       int op, const double* diag_add_dev, const double* diag_vec_dev, long diag_stride,
                       const double* add_mat_dev, int ld_add, double add_all, double mean_jitter, double* K_dev, int ldk,
                       int lower_only);
/* Backward of that call for a dense upstream dK [n1, n2]: g[e] = d sum(dK o K) / d{w, amp, kparam, center} of part e, one
   read of dK (autograd through Sum/ProductKernel.forward in the reference: two kernel backward chains).   */
int ffgp_kernel_grad_pair(ffgp_handle* h, const double* X1_dev, int n1, const double* X2_dev, int n2, int D, const ffgp_kdesc* k,
                          int op, const double* dK_dev, int ldk, const ffgp_kdesc_grads* g);

/* The same three calls for a nested composition (ffgp_ktree).  ffgp_kernel_input_weights_tree: for an upstream dK [n1, n2], leaf
   e's weight matrix lands at Wt_dev + e * leaf_stride (each [n1, ldw]):
       stationary leaf  Wt_e = dK o (d root / d leaf_e) o amp_e (-2 phi'_e):  dX1 = -w_e^2 o (rowsum(Wt_e) o X1 - Wt_e X2)
       linear leaf      Wt_e = dK o (d root / d leaf_e) o amp_e:              dX1 =  w_e^2 o (Wt_e (X2 - c_e))
   (autograd through Sum/ProductKernel.forward w.r.t. the inputs in the reference -- the acquisition loops of
   Bayesian_optimization/acq.py on the demo kernel SumKernel(LinearKernel, MaternKernel)).                         */
int ffgp_assemble_tree(ffgp_handle* h, const double* X1_dev, int n1, const double* X2_dev, int n2, int D, const ffgp_ktree* t,
                       const double* diag_add_dev, const double* diag_vec_dev, long diag_stride, const double* add_mat_dev,
                       int ld_add, double add_all, double mean_jitter, double* K_dev, int ldk, int lower_only);
int ffgp_kernel_grad_tree(ffgp_handle* h, const double* X1_dev, int n1, const double* X2_dev, int n2, int D, const ffgp_ktree* t,
                          const double* dK_dev, int ldk, const ffgp_kdesc_grads* g);
int ffgp_kernel_input_weights_tree(ffgp_handle* h, const double* X1_dev, int n1, const double* X2_dev, int n2, int D,
                                   const ffgp_ktree* t, const double* dK_dev, int ldk, double* Wt_dev, int ldw, long leaf_stride);

/* In-place lower Cholesky, A = L L^T (strictly-upper part is not referenced and not written).
   Replaces torch.linalg.cholesky at cigp_v10.py:35,61; gp_computation_pack.py:67,105,128; gp_basic.py:80,131;
   base_gp/cigp.py:83,129.  lda must be even and A_dev 16-byte aligned.                                     */
int ffgp_potrf(ffgp_handle* h, double* A_dev, int n, int lda);

/* Same factorisation with "passenger" rows: A is mtot x n (mtot >= n); the leading n x n block is factored and
   rows n..mtot-1 come out as A[n:, :] L^-T, i.e. (L^-1 B)^T for a right-hand side stored transposed below Sigma.
   This is how the fused paths obtain Gamma = L^-1 Y (cigp_v10.py:63) and V = L^-1 K_* (cigp_v10.py:36) inside
   the factorisation's own matrix-core GEMMs instead of separate triangular sweeps.                            */
int ffgp_potrf_rows(ffgp_handle* h, double* A_dev, int n, int mtot, int lda);

/* Backward of a standalone kernel evaluation: g_w[D], g_amp[1] = d sum(dK o K(x1,x2)) / d{w, amp} for a dense upstream
   dK [n1, n2] (autograd through kernel.forward when a caller builds its own Sigma, e.g. GaussianProcess/cigp_withMean.py:52). */
int ffgp_kernel_grad(ffgp_handle* h, const double* X1_dev, int n1, const double* X2_dev, int n2, int D,
                     const double* w_dev, const double* amp_dev, double clamp_min, int kfun, double kparam,
                     const double* dK_dev, int ldk, double* g_w_dev, double* g_amp_dev, double* g_kparam_dev);

/* Batched symmetric eigendecomposition of `batch` matrices M_b [n, n] (n <= 64; both triangles read), hand-written
 * two-sided cyclic Jacobi, one workgroup per matrix in LDS: Q_b's columns are the eigenvectors, evals_b the eigenvalues,
 * ascending (descending != 0: descending).  Replaces torch.linalg.eigh for the per-mode kernels of the HOGP block
 * (two_fidelity_models/hogp_simple.py:17-19,99-100) and is the inner solver of functional.eigh's blocked one-sided Jacobi. */
int ffgp_syevj_small(ffgp_handle* h, const double* M_dev, int n, int ldm, int batch, long strideM, double* Q_dev, int ldq,
                     long strideQ, double* evals_dev, long strideE, int descending);

/* ---- symmetric eigensolver (the N x N `torch.linalg.eigh(K_x)` of the HOGP block) ------------------------------------------------
 * Full symmetric eigendecomposition A = Z diag(W) Z^T of a dense symmetric matrix (its LOWER triangle is read, as
 * torch.linalg.eigh's default UPLO does), eigenvalues ascending, eigenvectors in the COLUMNS of Z -- the contract of
 * `torch.linalg.eigh`, which the reference calls at two_fidelity_models/hogp_simple.py:15-19 (`eigen_pairs`), :97-100 (once per
 * likelihood evaluation on the N x N input kernel) and MFGP_ver2023May/base_gp/hogp.py:20-24.  Hand-written two-stage solver:
 * dense -> band (bandwidth 32; TSQR + Householder reconstruction per panel, rank-64 trailing updates on the fp64 matrix cores),
 * band -> tridiagonal (bulge chasing, one wavefront per sweep, pipelined through progress counters), tridiagonal divide &
 * conquer on the device (secular equation per root, Gu-Eisenstat vectors, merges as GEMMs), back-transformation with the
 * chase's reflectors (blocked WY on the matrix cores) and with the panels' (256-wide block reflectors, GEMMs).
 * n <= 32768 (workspace ~ 10 n^2 doubles).  A is not modified.  Synchronises the handle's stream before returning.                                       */
int ffgp_syevd(ffgp_handle* h, const double* A_dev, int n, int lda, double* W_dev, double* Z_dev, int ldz);

/* The stages of ffgp_syevd on caller-owned buffers (n a multiple of 64, 64 <= n <= 32768) -- LAPACK's dsytrd_sy2sb / dsytrd_sb2st /
 * dstedc / dormtr split, exposed for tests, profiling and callers that want eigenvalues only:
 *   ffgp_sy2sb   A [n, n] full symmetric, DESTROYED -> AB [n, 64] band storage (element (r, c), 0 <= r - c <= 32, at AB[c * 64 + r - c])
 *                and Y [n, ldy]: the panels' unit-lower Householder blocks (panel p in columns 32p.., rows 32p + 32..; zero elsewhere)
 *   ffgp_sb2st   AB (destroyed) -> d [n], e [n] (e[n-1] = 0) and the chase's reflectors (refl: ffgp_sb2st_reflector_doubles(n) doubles)
 *   ffgp_stedc   (d, e) -> W [n] ascending, Z [n, ldz] eigenvectors of the tridiagonal matrix in columns
 *   ffgp_ormq2   Z[:, :ncols] <- Q2 Z   (the chase's reflectors);   ffgp_ormq1   Z[:, :ncols] <- Q1 Z   (the panels')
 * so that A = (Q1 Q2 Z) diag(W) (Q1 Q2 Z)^T.                                                                                   */
int ffgp_sy2sb(ffgp_handle* h, double* A_dev, int n, int lda, double* AB_dev, double* Y_dev, int ldy);
long ffgp_sb2st_reflector_doubles(int n);
int ffgp_sb2st(ffgp_handle* h, double* AB_dev, int n, double* d_dev, double* e_dev, double* refl_dev);
int ffgp_stedc(ffgp_handle* h, const double* d_dev, const double* e_dev, int n, double* W_dev, double* Z_dev, int ldz);
int ffgp_ormq2(ffgp_handle* h, const double* refl_dev, int n, double* Z_dev, int ldz, int ncols);
int ffgp_ormq1(ffgp_handle* h, const double* Y_dev, int ldy, int n, double* Z_dev, int ldz, int ncols);

/* ffgp_gemm over `batch` identical problems at fixed element strides (C_b = alpha op(A_b) op(B_b) + beta C_b). */
int ffgp_gemm_batched(ffgp_handle* h, int opa, int opb, int lower_tiles, const double* A_dev, int lda, long strideA,
                      const double* B_dev, int ldb, long strideB, double* C_dev, int ldc, long strideC, int m, int n, int k,
                      double alpha, double beta, int batch);

/* found_dev[i] = 1 iff row i of X1 [n1, D] equals (IEEE ==, element-wise) some row of X2 [n2, D]: the subset / unique-point
 * masks of the reference's data manager, `torch.all(x1.unsqueeze(1) == x2.unsqueeze(0), -1).any(-1)`
 * (FidelityFusion_Models/MF_data.py:196-199,234-237), as a device hash join instead of an N1 x N2 x D boolean temporary. */
int ffgp_rows_in(ffgp_handle* h, const double* X1_dev, int n1, const double* X2_dev, int n2, int D, unsigned char* found_dev);

/* Input gradients of a kernel call -- backward of `kernel(x1, x2)` w.r.t. x1 / x2, which the reference's autograd provides
 * and its acquisition optimisers rely on (Bayesian_optimization/acq.py:10-80 differentiate the posterior w.r.t. the test
 * points; CIGP_withMean.forward Bayesian_optimization/cigp.py:52-70).  For an upstream dK [n1, n2] writes
 *     Wt_ij = dK_ij * amp * (-2 phi'(s_ij))     (0 where the squared distance sits on the clamp)
 * so that dX1 = -w^2 o (rowsum(Wt) o X1 - Wt X2) and dX2 = -w^2 o (colsum(Wt) o X2 - Wt^T X1): two thin ffgp_gemm calls. */
int ffgp_kernel_input_weights(ffgp_handle* h, const double* X1_dev, int n1, const double* X2_dev, int n2, int D,
                              const double* w_dev, const double* amp_dev, double clamp_min, int kfun, double kparam,
                              const double* dK_dev, int ldk, double* Wt_dev, int ldw);

/* Rebuild the handle's store of inverted 128x128 diagonal blocks for a factor L that this handle did not just
   produce (the triangular solves and ffgp_potri consume it; ffgp_potrf leaves it up to date).
   CONTRACT of the cached inverses (ffgp_trsm_lower, ffgp_trsm_lower_t, ffgp_potrs, ffgp_potri): the store -- and the
   inverted super-blocks built on top of it -- is keyed on (L_dev, n, ldl) only.  A factor passed to those calls must be
   unmodified since the ffgp_potrf / ffgp_potrf_rows / ffgp_trtri_diag call ON THIS HANDLE that keyed the store.  If the
   contents at that address changed any other way (a buffer refilled by the caller, a factor written by another handle,
   a recycled allocation), call ffgp_trtri_diag again (or ffgp_invalidate) before solving with it.              */
int ffgp_trtri_diag(ffgp_handle* h, const double* L_dev, int n, int ldl);

/* Drop the handle's cached inverses (diagonal blocks and super-blocks): the next triangular solve rebuilds them from
   the factor it is given.  Cheap; use it whenever a factor buffer is refilled behind the handle's back.        */
int ffgp_invalidate(ffgp_handle* h);

/* B <- L^-1 B.  Replaces torch.triangular_solve(B, L, upper=False) (cigp_v10.py:36,63;
   gp_computation_pack.py:130) and `L.inverse() @ B` (base_gp/cigp.py:131; gp_computation_pack.py:108).      */
int ffgp_trsm_lower(ffgp_handle* h, const double* L_dev, int n, int ldl, double* B_dev, int nrhs, int ldb);

/* B <- L^-T B (second half of cholesky_solve). */
int ffgp_trsm_lower_t(ffgp_handle* h, const double* L_dev, int n, int ldl, double* B_dev, int nrhs, int ldb);

/* B <- (L L^T)^-1 B.  Replaces torch.cholesky_solve (cigp_v10.py:39; gp_computation_pack.py:76,106). */
int ffgp_potrs(ffgp_handle* h, const double* L_dev, int n, int ldl, double* B_dev, int nrhs, int ldb);

/* out_dev[0] = 1/2*sum(M^2) + d*sum(log L_ii) + 1/2*n*d*log(2*pi_const)   (M = Gamma for V1, Sigma^-1 Y for V2 with
   the log-det counted as in FFGP_LL_V2).  Replaces cigp_v10.py:67-68 / gp_computation_pack.py:79.          */
int ffgp_nll_reduce(ffgp_handle* h, int variant, const double* L_dev, int n, int ldl, const double* M_dev, int d,
                    int ldm, double pi_const, double* out_dev);

/* Sinv (lower triangle) <- (L L^T)^-1 from the factor, via blocked TRTRI + LAUUM on the matrix cores. */
int ffgp_potri(ffgp_handle* h, double* L_dev, int n, int ldl);

/* C[m,n] = alpha * op(A) op(B) + beta * C on the fp64 matrix cores (the kernel under every O(N^3) stage).
   opa = 0: A stored m x k (k contiguous); opa = 1: A stored k x m (m contiguous), i.e. op(A) = A^T.
   opb = 0: B stored n x k (k contiguous), i.e. op(B) = B^T;  opb = 1: B stored k x n (n contiguous).
   lower_tiles != 0: only elements with col <= row are computed/written (m >= n, origin on the diagonal).
   tri: OR of 1 (k starts at the tile row), 2 (k starts at the tile column), 4 (k ends after the tile row),
        8 (k ends after the tile column) -- skips k-tiles that are structurally zero for triangular operands.
   Supported (opa, opb, lower_tiles): (0,0,*), (0,1,0), (1,0,0), (1,1,*).                                    */
int ffgp_gemm(ffgp_handle* h, int opa, int opb, int lower_tiles, int tri, const double* A_dev, int lda,
              const double* B_dev, int ldb, double* C_dev, int ldc, int m, int n, int k, double alpha, double beta);

/* ---- fused hot path ------------------------------------------------------------------------------------- */
/* nll_dev[0] <- negative log marginal likelihood of the block (V1: +nll; V2: -LL); if g != NULL also its
   closed-form gradients.  One call = assemble -> potrf -> trsm/potrs -> reduce (-> potri -> fused gradient).
   Replaces cigp.negative_log_likelihood (cigp_v10.py:50-69) + loss.backward() (FidelityFusion_Models/ResGP.py:84-87),
   gp_computation_pack.negative_log_likelihood (:120-136), GP_basic.log_likelihood (gp_basic.py:94-143),
   CIGP.compute_loss (MFGP_ver2023May/base_gp/cigp.py:99-136).  Returns 0, or the failing pivot index.       */
int ffgp_nlml_fused(ffgp_handle* h, const ffgp_problem* p, double* nll_dev, const ffgp_grads* g);

/* The same call on RAW parameters: w_dev / amp_dev / diag_add_dev of the problem hold the module's own parameters and `links` names
   the elementwise maps to the effective quantities (the O(D) maps the reference's modules apply in torch: `abs() + eps` and the
   reciprocal of ARDKernel / MaternKernel, kernel.py:98,157; `exp` of SquaredExponentialKernel, :262-266; `exp(-log_beta)` of cigp,
   cigp_v10.py:57; `noise_variance ** 2` of GP_basic, gp_basic.py:63).  The maps and their chain rule run as two tiny kernels
   around the fused call, so a training step needs ONE library call instead of a dozen elementwise torch kernels and their
   autograd nodes -- which is what a step costs at the sizes the reference's own demos run (N = 16 ... 300).  Gradients in `g`
   come back with respect to the RAW parameters; with w_broadcast the single raw length scale feeds all D dimensions and g_w_dev
   receives ONE value.  D <= 128.                                                                                              */
enum {
  FFGP_LINK_ID = 0,           /* e = p                                                             */
  FFGP_LINK_INV_ABS_EPS = 1,  /* e = 1 / (|p| + c)                                                 */
  FFGP_LINK_EXP_NEG = 2,      /* e = exp(-p) + c                                                   */
  FFGP_LINK_INV = 3,          /* e = 1 / p + c                                                     */
  FFGP_LINK_ABS = 4,          /* e = |p|                                                           */
  FFGP_LINK_EXP_SQ = 5,       /* e = exp(p)^2                                                      */
  FFGP_LINK_SQUARE = 6        /* e = p^2 + c                                                       */
};
typedef struct {
  int w_link; double w_c; int w_broadcast;
  int amp_link; double amp_c;
  int dadd_link; double dadd_c;
  double out_scale;   /* the value and every gradient are multiplied by this (0 is read as 1): -1 turns the nll into the +LL the
                         reference's `negative_log_likelihood` returns (cigp_v10.py:69) without another elementwise kernel */
} ffgp_links;
int ffgp_nlml_fused_raw(ffgp_handle* h, const ffgp_problem* p, const ffgp_links* links, double* nll_dev, const ffgp_grads* g);
/* the same call enqueued only: the status (a not-PD Sigma) is collected by the next ffgp_wait on this handle -- the Python modules
   issue a training step's likelihood this way and collect in backward(), so the host prepares the backward pass while the GPU works */
int ffgp_nlml_fused_raw_async(ffgp_handle* h, const ffgp_problem* p, const ffgp_links* links, double* nll_dev, const ffgp_grads* g);

/* F independent SMALL problems (n <= 128, D <= 16, d <= 16, one radial-profile kernel each, no caller-built covariance) in one
   launch per eight problems -- one workgroup each, everything in LDS: the per-fidelity / per-seed loops of the reference's
   experiments (Experiments/GAR_Aligned/exp_aligned.py:58-126; every model there has N = 16 ... 128) call cigp.negative_log_likelihood
   for one such model after the other.  p, g: arrays of F; links: array of F (raw parameters, as ffgp_nlml_fused_raw) or NULL
   (effective parameters); problem f's value lands in nll_dev[f].  The status is shared: a Sigma that is not positive definite in
   ANY member is reported (as that member's leading minor).  The _async form only enqueues (status: next ffgp_wait).
   Since round 6 problems within the one-launch trainer's limits (V1 likelihood, no add_mat / add_all / mean_jitter, no g_kparam) run on
   its matrix-core kernel (csrc/train.hip, evaluate mode: eight models per launch), as does ffgp_nlml_fused_raw at n <= 128; the others
   keep the scalar one-workgroup kernel.  Option "train_persist" = 0 restores the old routing.                                     */
int ffgp_nlml_fused_small_batch(ffgp_handle* h, int F, const ffgp_problem* p, const ffgp_links* links, double* nll_dev,
                                const ffgp_grads* g);

/* F independent blocks (2 <= F <= 256; n > 128 each; V1 likelihood; one radial-profile kernel each) in one chain of launches: the
   per-fidelity / per-seed loops of the reference evaluate its blocks one after the other
   (Experiments/GAR_Aligned/exp_aligned.py:58-126, FidelityFusion_Models/ResGP.py:78-112), and below N ~ 6000 one block's
   factorisation is a latency-bound chain of short launches.  Here every launch of that chain covers all F blocks (diagonal-block
   kernel: one workgroup per block; GEMMs: the block index in gridDim.y), so the blocks share ONE chain; the arithmetic per block is
   the single call's, and the values are bit-identical to F separate calls.
   The blocks may have ONE shape (any n > 128) or DIFFERENT n and d (n <= 12288 each) -- the reference's fidelities are ragged:
   300 / 300 / 250 points in FidelityFusion_Models/ResGP.py:121-136, 100 low against 4..32 high in
   Experiments/GAR_Aligned/exp_aligned.py:66-74.  In a ragged batch every member follows the launch sequence of its own single call
   (in order up to "la_min_n" rows, the look-ahead's carry form above), launches of one kind at one chain step are merged with
   per-member sizes, and a member leaves the chain's launches when its columns are used up: the chain runs max(n) / 128 steps.
   p, g (may be NULL), links (NULL = effective parameters): arrays of F; nll_dev[f] receives block f's value; status[f] (host, may
   be NULL) its own factorisation status (0, or the 1-based index of the first non-positive pivot of THAT block).  Returns the first
   non-zero status; FFGP_ERR_ARG when the blocks do not meet the conditions -- also with options "naive" = 1 or "diag_v2" = 0, F > 256,
   a ragged member above 12288 rows -- and FFGP_ERR_ALLOC when the F-fold workspace does not fit: evaluate the blocks one by one
   (or in smaller batches) then, as fidelityfusion_amd/nlml.py::_chain_batches does.  Synchronous.                               */
int ffgp_nlml_fused_batch(ffgp_handle* h, int F, const ffgp_problem* p, const ffgp_links* links, double* nll_dev,
                          const ffgp_grads* g, int* status);

/* K Adam training steps of F independent models (F <= 16) in ONE call -- the reference's hot loop
   (FidelityFusion_Models/ResGP.py:78-112; GaussianProcess/cigp_v10.py:92-104: per fidelity 100-1000 iterations of
   optimizer.zero_grad(); loss = -model.negative_log_likelihood(x, y); loss.backward(); optimizer.step() at N = 16 ... 500).
   p[f] / links[f] describe model f on its RAW parameters exactly as for ffgp_nlml_fused_raw (w_dev = raw length scales, amp_dev = raw
   signal variance, diag_add_dev = raw log_beta; links->out_scale = the sign that makes the value the loss to MINIMISE, +1 for
   loss = -negative_log_likelihood).  Per step: the likelihood and its closed-form gradients on the raw parameters (one launch for all
   models when every model is small -- the ffgp_nlml_fused_small_batch limits -- otherwise the launches of ffgp_nlml_fused_raw model
   after model), then one kernel that applies torch.optim.Adam's update (betas, eps, bias corrections as torch computes them; no
   weight decay, no amsgrad) to the three raw parameter tensors IN PLACE and stores the step's loss.  No host synchronisation inside
   the loop.  state_dev: per model [exp_avg (nw + 2) | exp_avg_sq (nw + 2)] at stride state_stride doubles (nw = 1 for a broadcast
   scalar length scale, else D; order: w, amp, diag_add), zero for a fresh optimiser and carried between calls together with step0 =
   the number of steps already taken; trace_dev[f * trace_stride + k] = loss of model f at step k, evaluated BEFORE that step's
   update (what the reference prints).  Returns 0, or the pivot status of the first step whose Sigma was not positive definite
   (torch.linalg.LinAlgError in the reference's loop): from that step on no parameter moves and the trace holds NaN.  Synchronous.
   ONE LAUNCH FOR THE WHOLE LOOP (round 6, csrc/train.hip): when every model has n <= 128, D <= 16, d <= 16, the V1 likelihood and no
   Sigma extra but diag_add / diag_vec -- what the reference's experiments run (Experiments/GAR_Aligned/exp_aligned.py:66-74: 100 low-
   against 4..32 high-fidelity points) -- the call is one kernel launch: a persistent workgroup per model keeps Sigma, its factor, the
   inverse, the parameters and the Adam moments in LDS and runs all `steps` iterations inside the kernel (option "train_persist" = 0
   restores the launch-per-stage loop).  There a failing model stops alone -- its trace holds NaN from its failing step on, its
   parameters are those it had when that step began -- while the other models of the call complete their steps; the return value is
   the first failing model's pivot status.  200 steps at n = 128: 9.7 ms (19.2 ms launch by launch); n = 32: 16 us per step. */
typedef struct ffgp_adam {
  double lr, beta1, beta2, eps;
} ffgp_adam;
int ffgp_train_raw(ffgp_handle* h, int F, const ffgp_problem* p, const ffgp_links* links, int steps, const ffgp_adam* opt,
                   double* state_dev, long state_stride, long step0, double* trace_dev, long trace_stride);
int ffgp_nlml_fused_small_batch_async(ffgp_handle* h, int F, const ffgp_problem* p, const ffgp_links* links, double* nll_dev,
                                      const ffgp_grads* g);

/* Same, enqueue only: returns as soon as the work is on the handle's stream (nll/gradients are valid after
   ffgp_wait).  With one handle + stream per block, independent GP blocks (the fidelities of one model, the seeds
   of an experiment sweep) overlap on one GPU: one block's latency-bound factorisation tail runs under another
   block's trailing updates.                                                                                  */
int ffgp_nlml_fused_async(ffgp_handle* h, const ffgp_problem* p, double* nll_dev, const ffgp_grads* g);
/* Synchronise the handle's stream; returns the status of the last enqueued fused call (0 / failing pivot). */
int ffgp_wait(ffgp_handle* h);

/* Posterior at Xs[nt, D] for the block p (its ll_variant/pi_const are ignored; Sigma extras are honoured
   as given -- the reference's predict paths drop y_var, so callers pass the problem without it).
   mean_dev[nt, d] = K*^T Sigma^-1 Y;  var per var_mode with var_add_all added to every entry.
   Replaces cigp.forward (cigp_v10.py:24-48), conditional_Gaussian (gp_computation_pack.py:103-110),
   GP_basic.forward (gp_basic.py:78-84), CIGP.forward (base_gp/cigp.py:78-95).                               */
int ffgp_predict(ffgp_handle* h, const ffgp_problem* p, const double* Xs_dev, int nt, int var_mode,
                 double var_add_all, double* mean_dev, double* var_dev, int ldv);

/* ---- multi-GPU: the joint likelihood of sharded blocks --------------------------------------------------- */
/* buf_dev[0..count) <- element-wise SUM over the ranks of `comm` (in place, fp64), enqueued on the handle's stream: the ONE
   collective of the per-fidelity sharding -- the F-vector of per-block values, `loss += cigp_list[f].compute_loss(...)`
   in MFGP_ver2023May/ResGP.py:235,245 (the 2024 trainers' per-fidelity loop, FidelityFusion_Models/CIGAR.py:99-134).
   `comm` is the caller's ncclComm_t (RCCL); the library resolves ncclAllReduce from librccl.so.1 at the first call
   (dlopen: libffgp.so has no link-time dependency on RCCL, and a process that already loaded RCCL -- torch -- shares it).
   Returns FFGP_ERR_ARG for bad arguments, FFGP_ERR_HIP if RCCL cannot be loaded or reports an error.              */
int ffgp_allreduce_sum(ffgp_handle* h, void* comm, double* buf_dev, int count);

/* ---- instrumentation ------------------------------------------------------------------------------------ */
/* stage timings (ms) of the last fused call when option "timing" = 1; names are static strings */
int ffgp_last_timings(ffgp_handle* h, float* ms_out, const char** names_out, int max_stages, int* n_stages);
/* accumulated launches / algorithmic flops / device ms of the trailing-update SYRK kernel (the roofline kernel);
   reset = 1 clears the counters after reading.  ms is only accumulated while option "timing" = 2.           */
int ffgp_syrk_stats(ffgp_handle* h, double* flops, double* ms, long* launches, int reset);
/* peak probe: runs a register-resident v_mfma_f64_16x16x4_f64 loop on every CU, returns measured TFLOP/s */
int ffgp_mfma_f64_peak(ffgp_handle* h, double* tflops_out);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* FFGP_H */
