/*
 * gumbi_hip.h -- C ABI of libgumbi_hip.so, the MI355X (gfx950) exact-GP engine behind
 * Gumbi's `GP.fit() / prepare_grid() / predict_grid()`.
 *
 * The reference (JohnGoertz/Gumbi v0.4.1) has no FFI: its backend seam is the Python ABC
 * `gumbi.regression.base.Regressor` (gumbi/regression/base.py:21) whose subclasses implement
 * `fit` (:119-127), `build_model` (:129-137), `predict(points_array, with_noise) -> (mean, var)`
 * (:477-495) and `find_MAP` (called from :1064).  The arithmetic those methods reach lives in
 * PyMC (`pm.gp.Marginal`, `pm.gp.cov.*`, `pm.find_MAP`; call sites
 * gumbi/regression/pymc/GP.py:407-410, 453, 460-462, 560-561, 580, 811, 845-847).  Each entry
 * point below names the reference call it replaces.  INTEGRATION.md shows the ctypes stub a
 * Gumbi maintainer would add.
 *
 * Conventions
 *   - every function returns 0 (GMB_OK) or a negative gmb_status; nothing throws across the ABI;
 *   - a handle is NOT thread-safe, distinct handles are independent;
 *   - all matrices handed in are row-major float64 unless stated; all sizes are int64_t;
 *   - `memspace` says whether caller pointers are host (GMB_HOST) or device (GMB_DEVICE)
 *     memory; the library never retains a caller pointer after the call returns;
 *   - no torch / pybind types anywhere in the signatures.
 *
 * Parameter vector `theta` (natural scale), n = gmb_theta_size(spec):
 *   [ ls (n_cont if ard else 1) | eta | sigma |
 *     c (n_lin), tau                       -- only if n_lin > 0          (pymc/GP.py:451-453)
 *     per coregion dim: W (L x 2 row-major), kappa (L)                   (pymc/GP.py:459-462)
 *     W_out (P x 2), kappa_out (P)         -- only if out_col >= 0       (pymc/GP.py:724-727)
 *     W_noise (P x 2), kappa_noise (P)     -- only if out_col >= 0 && hetero_noise (:565-569) ]
 */
#ifndef GUMBI_HIP_H
#define GUMBI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMB_ABI_VERSION 10
#define GMB_MAX_DIMS 16   /* continuous dims per kernel */
#define GMB_MAX_LIN 8     /* linear dims per kernel (subset of the continuous dims) */
#define GMB_MAX_COREG 4   /* categorical (coregion) dims besides the output column */
#define GMB_MAX_LEVELS 32 /* levels per coregion table */

typedef enum gmb_status {
  GMB_OK = 0,
  GMB_EINVAL = -1,   /* bad argument / call order            (ValueError in Python)            */
  GMB_ENOMEM = -2,   /* hipMalloc failed                                                       */
  GMB_EHIP = -3,     /* a HIP runtime call or kernel launch failed; see gmb_last_error         */
  GMB_ENOTPD = -4,   /* covariance not positive definite; see gmb_notpd_index (LinAlgError)    */
  GMB_ENODEVICE = -5 /* no usable gfx950 device -- there is NO CPU fallback                    */
} gmb_status;

typedef enum gmb_kernel_kind { /* pm.gp.cov.<name>, selected at pymc/GP.py:664-674,684 */
  GMB_EXPQUAD = 0,
  GMB_MATERN52 = 1,
  GMB_MATERN32 = 2,
  GMB_MATERN12 = 3,
  GMB_EXPONENTIAL = 4
} gmb_kernel_kind;

typedef enum gmb_memspace { GMB_HOST = 0, GMB_DEVICE = 1 } gmb_memspace;

/* Kernel structure Gumbi declares in PymcGP._construct_kernels (pymc/GP.py:652-757, non-additive):
 *   K = (eta^2 k_cont [+ tau Linear(c)]) * prod_dims Coregion_dim * Coregion_out
 *   noise = WhiteNoise(sigma) [* Coregion("Output_noise")]     (pymc/GP.py:560-569)
 *   Sigma = K + noise + jitter I                               (pm.gp.Marginal, jitter 1e-6) */
typedef struct gmb_kernel_spec {
  int32_t kind;                      /* gmb_kernel_kind                                        */
  int32_t ard;                       /* 1: one lengthscale per continuous dim (fit(ARD=True))  */
  int32_t n_cont;                    /* number of continuous dims (active_dims of k_cont)      */
  int32_t idx_cont[GMB_MAX_DIMS];    /* their column indices in X                              */
  int32_t n_lin;                     /* linear dims (subset of continuous)                     */
  int32_t idx_lin[GMB_MAX_LIN];
  int32_t n_coreg;                   /* categorical dims other than the output column          */
  int32_t coreg_col[GMB_MAX_COREG];
  int32_t coreg_levels[GMB_MAX_COREG];
  int32_t out_col;                   /* output-coregion column or -1 (single output)           */
  int32_t n_out;                     /* number of outputs P                                    */
  int32_t hetero_noise;              /* heteroskedastic_outputs (pymc/GP.py:565)               */
  int32_t additive;                  /* specify_model(additive=True) (pymc/GP.py:732-754): a global
                                      * kernel plus one kernel per coregion dim, each times that dim's
                                      * table (and the output table); theta gains, after the layout
                                      * above, one block [ls | eta | (c, tau)] per coregion dim     */
  double jitter;                     /* 1e-6 to follow pm.gp.Marginal                          */
} gmb_kernel_spec;

typedef struct gmb_timings { /* milliseconds on the engine's HIP stream (hipEvent pairs); see gmb_evaluate for when they are taken */
  double kbuild_ms;          /* covariance-matrix construction (last gmb_factorize)            */
  double chol_ms;            /* whole Cholesky (leaves + trsm + trailing updates)              */
  double chol_gemm_ms;       /* sum over the MFMA SYRK/GEMM trailing-update launches           */
  double chol_gemm_flops;    /* flops those launches performed (2*m*n*k per computed tile)     */
  int64_t chol_gemm_launches;
  double chol_leaf_ms;       /* diagonal-block potrf + triangular-inverse leaves               */
  double chol_trsm_ms;       /* panel solves                                                   */
  double predict_ms;         /* last gmb_predict, all tiles                                    */
  double predict_gemm_ms;
  double predict_gemm_flops;
  int64_t predict_gemm_launches;
  double grad_ms;            /* last gmb_nlml gradient (inverse + trace reductions)            */
  double grad_gemm_ms;
  double grad_gemm_flops;
  double kbuild_bytes;       /* algorithmic bytes the K-build wrote+read (8N(N+1)/2 + 8N(d+1)) */
  /* cumulative since the last gmb_set_profiling(e, 1): every MFMA GEMM launch of every phase */
  double total_gemm_ms;
  double total_gemm_flops;
  int64_t total_gemm_launches;
  double total_kbuild_ms;
  double total_kbuild_bytes;
  int64_t total_kbuild_launches;
  /* the part of total_gemm_* that ran on the CU-masked bulk stream (trailing updates of the masked
     look-ahead schedule: masked_cus of the device's compute units were available to them)        */
  double masked_gemm_ms;
  double masked_gemm_flops;
  int64_t masked_cus;
  /* wall time with at least one GEMM launch in flight (union of the launch intervals: launches of
     the look-ahead schedules' two streams overlap, total_gemm_ms is the SUM of their durations)    */
  double total_gemm_wall_ms;
  /* the Cholesky's BULK trailing-update launches alone (U1 / U2 of the look-ahead schedules, every update
     of the plain recursion, the UPDATE steps of the multi-GPU driver), cumulative like total_gemm_*: the
     kernel launches the north star's MFMA-utilisation target is stated on.  The small in-panel products
     of the latency-bound panel chain are counted separately (total_chol_panel_gemm_*).               */
  double total_chol_gemm_ms;
  double total_chol_gemm_flops;
  int64_t total_chol_gemm_launches;
  double total_chol_gemm_wall_ms; /* union of their launch intervals */
  double total_chol_panel_gemm_ms;
  double total_chol_panel_gemm_flops;
  /* multi-GPU driver (gmb_dist_*), this rank, HIP events around every collective and at every stream join.
     chol_*: the last gmb_dist_factorize; grad_*: the last gmb_dist_nlml with a gradient.
       comm_ms         sum over the all-gathers of (stream reached the collective -> collective complete):
                       transfer time plus the wait for the slowest peer
       comm_exposed_ms the part of comm_ms that nothing hid.  Factorisation: per panel, min(time the bulk
                       stream sat idle waiting for the panel chain, time that chain spent in its two
                       all-gathers) -- the first chain has nothing to hide behind and counts in full.
                       Gradient: the main stream's wait for the communication stream after the last chunk
                       plus the two small all-gathers (alpha, accumulators) that sit on the main stream
       main_wait_ms    main stream waiting for the bulk stream at the JOINs (trailing update U2 longer than
                       the next panel's chain: communication fully hidden there)
       bulk_wait_ms    bulk stream waiting for the main stream at the FORKs (chain-bound time)          */
  int64_t dist_world;             /* ranks of the communicator the last gmb_dist_* call ran on            */
  int64_t dist_chol_collectives;
  double dist_chol_comm_bytes;    /* bytes received from the other ranks                                  */
  double dist_chol_comm_ms;
  double dist_chol_comm_exposed_ms;
  double dist_chol_main_wait_ms;
  double dist_chol_bulk_wait_ms;
  int64_t dist_grad_collectives;
  double dist_grad_comm_bytes;
  double dist_grad_comm_ms;
  double dist_grad_comm_exposed_ms;
  /* Every rank computes the factorisation's failure index, log-determinant and |v|^2 redundantly (bit-identical by
     construction); all ranks nevertheless LEAVE gmb_dist_factorize with rank 0's copy, so that optimisers running in
     lock step on every rank can never part company.  Cumulative count of ranks whose own copy differed (expected 0). */
  int64_t dist_lockstep_repairs;
  /* The persistent tile Cholesky (csrc/chol_tiles.hpp: the WHOLE factorisation of a small matrix -- contraction, leaves and
     strip solves -- in one launch), cumulative like total_gemm_*: launch durations (HIP events), the flops of the tiles'
     contractions, launches.  Such a factorisation has no separate trailing-update launches: total_chol_gemm_* stay 0. */
  double total_chol_tile_ms;
  double total_chol_tile_flops;
  int64_t total_chol_tile_launches;
  /* The persistent evaluation launch (csrc/eval_tiles.hpp: L^-T by rows and Sigma^-1 = U U^T as tile tasks -- in gmb_evaluate
     together with the factorisation's own tile tasks, ONE launch for the matrix work of a MAP evaluation), cumulative:
     launch durations, contraction flops (factorisation included when it rode along), launches. */
  double total_eval_tile_ms;
  double total_eval_tile_flops;
  int64_t total_eval_tile_launches;
  /* (ABI 10) every launch the Cholesky's schedules issue AS a trailing update, whatever tile shape it runs -- the population
     `total_chol_gemm_*` covered up to round 4; since round 5 total_chol_gemm_* holds only the updates large enough for the
     128 x 128 kernel (gemm_f64_dma_chol_update_kernel).  Cumulative like total_gemm_*. */
  double total_chol_update_all_ms;
  double total_chol_update_all_flops;
  int64_t total_chol_update_all_launches;
  /* (ABI 10) 1 when the last gmb_predict formed A^T = K(X*, X) L^-T as one GEMM against the inverse factor the fit's last
     gradient evaluation left behind (csrc/predict_form.hpp), 0 when it solved against L (gmb_set_predict_form) */
  int64_t predict_gemm_form;
  /* (ABI 10) the bottom panels of the large matrices' recursion as launches of the tile kernel (csrc/engine.hip:
     chol_tiles_panel): durations, contraction flops, launches -- a subset of total_chol_panel_gemm_* */
  double total_chol_panel_tile_ms;
  double total_chol_panel_tile_flops;
  int64_t total_chol_panel_tile_launches;
} gmb_timings;

typedef struct gmb_engine gmb_engine;

/* -- lifecycle ------------------------------------------------------------------------------ */

/* Number of visible HIP devices, or a negative gmb_status.  Never falls back to the CPU. */
int gmb_device_count(void);
int gmb_abi_version(void);

/* Create an engine bound to HIP device `device`.  `stream` is an existing hipStream_t to launch
 * on (e.g. torch's current stream), or NULL to let the engine create its own. */
int gmb_create(gmb_engine** out, int32_t device, void* stream);
/* A second engine on `peer`'s device that borrows ALL of peer's HIP streams (its own state and buffers otherwise):
 * for callers that keep several models resident and work on them one after the other -- the Kronecker
 * multi-output path keeps its P factorisations this way -- without multiplying the process's hardware queues.
 * Destroy it before `peer`; not for concurrent use with `peer`. */
int gmb_create_sibling(gmb_engine** out, const gmb_engine* peer);
void gmb_destroy(gmb_engine* e);
const char* gmb_last_error(const gmb_engine* e); /* valid until the next call on `e` */
void* gmb_stream(const gmb_engine* e);           /* the hipStream_t launches go to   */

/* -- model definition ------------------------------------------------------------------------
 * gmb_set_data replaces `X, y = self.get_shaped_data("mean")` being captured by the PyMC model
 * (pymc/GP.py:521, 580): X is (N, D) with leading dimension ldx (>= D), y is (N,).  Copies in. */
int gmb_set_data(gmb_engine* e, const double* X, int64_t N, int32_t D, int64_t ldx,
                 const double* y, int32_t memspace);
/* New observations for the SAME inputs (length N): kernel, theta, prepared coordinates and every workspace
 * stay; only the factorisation is invalidated.  No reference counterpart (there the model is rebuilt); used by
 * the Kronecker multi-output path, whose P systems share X and differ in the rotated y (regression/icm.py). */
int gmb_set_y(gmb_engine* e, const double* y, int32_t memspace);

/* Replaces PymcGP._construct_kernels + the noise block of build_model (pymc/GP.py:560-569,
 * 652-729): fixes which columns feed which covariance term. */
int gmb_set_kernel(gmb_engine* e, const gmb_kernel_spec* spec);
int gmb_theta_size(const gmb_kernel_spec* spec);

/* Hyper-parameters on the natural scale -- the role of `point=self.MAP` (pymc/GP.py:846). */
int gmb_set_theta(gmb_engine* e, const double* theta, int32_t n);

/* -- hot path --------------------------------------------------------------------------------
 * gmb_factorize: build Sigma = K(X,X) + noise + jitter*I (lower triangle, HBM-resident), Cholesky
 * factor it in place, forward-solve v = L^-1 y, accumulate sum(log L_ii).  This is what every
 * `pm.find_MAP` objective evaluation (pymc/GP.py:811) and every `Marginal.predict` call
 * (pymc/GP.py:845-847) does first.  Returns GMB_ENOTPD if a pivot is <= 0 or NaN. */
int gmb_factorize(gmb_engine* e);
int64_t gmb_notpd_index(const gmb_engine* e); /* 0-based failing row after GMB_ENOTPD, else -1 */
/* 1 while the engine holds a usable factorisation of the current (data, kernel, theta): after gmb_factorize,
 * and still after gmb_nlml with a gradient (the single-GPU gradient puts the factor's diagonal blocks back);
 * 0 after gmb_set_* or after gmb_dist_nlml with a gradient (which leaves U = L^-T in the factor buffer). */
int gmb_factor_valid(const gmb_engine* e);

/* One objective (+ gradient) evaluation at `theta` in ONE call: gmb_set_theta + gmb_factorize + gmb_nlml, with the gradient's
 * launches enqueued right behind the factorisation's -- no host synchronisation or language round trip in between (at
 * N = 2000 that is 0.1 of 1.5 ms per evaluation of pm.find_MAP's objective, gumbi/regression/pymc/GP.py:811).  Same status
 * codes and side effects as the three calls (GMB_ENOTPD with gmb_notpd_index when the covariance is not positive definite;
 * grad may be NULL), same values bit for bit where the three calls take the same schedules.  One difference: with a gradient,
 * on matrices below 4096 rows and without gmb_set_profiling, no per-phase events are recorded (they cost a tenth of such an
 * evaluation) -- gmb_timings' kbuild_ms / chol_ms / grad_ms then read 0 after this call. */
int gmb_evaluate(gmb_engine* e, const double* theta, int32_t n, double* nlml, double* grad);
/* Negative log marginal likelihood  N/2 log 2pi + sum log L_ii + |v|^2/2  of the last
 * factorisation, and (if grad != NULL, length gmb_theta_size) its gradient w.r.t. natural-scale
 * theta: 1/2 tr((Sigma^-1 - a a^T) dSigma/dtheta).  The factorisation stays valid (gmb_predict may follow
 * without another gmb_factorize).  Priors and Jacobians are added by the Python
 * caller so the MAP formula stays auditable.  Replaces the likelihood part of pm.find_MAP's
 * objective/gradient (pymc/GP.py:580, 811). */
int gmb_nlml(gmb_engine* e, double* nlml, double* grad);

/* Posterior mean and variance at M points -- `Regressor.predict` contract (base.py:477-495),
 * replacing `gp.predict(points_array, point=MAP, diag=True, pred_noise=with_noise)`
 * (pymc/GP.py:845-847).  Xs is (M, D) row-major with leading dimension ldxs; mean/var are
 * caller-allocated length-M arrays in the same memspace as Xs.  Reuses the resident factor. */
int gmb_predict(gmb_engine* e, const double* Xs, int64_t M, int64_t ldxs, int32_t with_noise,
                double* mean, double* var, int32_t memspace);

/* Pairwise-distance extrema used for the lengthscale prior (gumbi/utils/gp_utils.py:15-48):
 * for each group (one per column if ard, else all n_cols jointly) the minimum NON-ZERO and the
 * maximum Euclidean distance over all N(N-1)/2 pairs; lower[g] = -1 if every pair coincides.
 * X is (N, n_cols) row-major, host memory.  Runs on `device`. */
int gmb_ls_limits(int32_t device, const double* X, int64_t N, int32_t n_cols, int64_t ldx,
                  int32_t ard, double* lower, double* upper);

/* -- introspection --------------------------------------------------------------------------- */
int gmb_set_profiling(gmb_engine* e, int32_t on); /* per-launch hipEvent timing of GEMMs */
int gmb_timings_get(const gmb_engine* e, gmb_timings* out);

/* Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate of `device` in TFLOP/s (register-only
 * kernel, no memory traffic) -- the measured denominator for the MFMA roofline -- and the shader
 * cycles one wave spends per MFMA (64 = the datasheet issue rate; wall rate / this = clock). */
int gmb_mfma_f64_peak(int32_t device, double* tflops, double* cycles_per_mfma);
/* The same register-only loop launched back to back for `seconds` (<= 60; 37 ms per launch): mean and worst
 * per-launch rate and the shader clock in MHz the loop ran at (2 x s_memtime span / launch duration: the counter
 * ticks every second shader cycle on gfx950) -- the sustained
 * ceiling under the box's power management, to be sampled before AND after a timed region. */
int gmb_mfma_f64_sustained(int32_t device, double seconds, double* mean_tflops, double* min_tflops,
                           double* shader_mhz, int64_t* launches);

/* Copy out pieces of the resident state (tests / multi-GPU driver):
 *   rows [r0, r0+nr) x cols [c0, c0+nc) of the factor buffer (lower triangle meaningful),
 *   written row-major into `out` (nr*nc doubles, host). */
int gmb_copy_factor(const gmb_engine* e, int64_t r0, int64_t nr, int64_t c0, int64_t nc,
                    double* out);
int gmb_copy_v(const gmb_engine* e, double* out); /* v = L^-1 y, length N, host */
/* alpha = Sigma^-1 y (length N, host); exists after gmb_nlml with a gradient.  Used by the
 * Kronecker (ICM) multi-output path, whose outer chain rule needs it (gumbi_amd/regression/icm.py). */
int gmb_copy_alpha(const gmb_engine* e, double* out);

/* -- kernel-level doors (tests, tuning tools; device pointers, column-major) -------------------
 * The leaf, strip-solve and MFMA GEMM kernels of the factorisation on caller-provided blocks, so
 * that each can be checked against LAPACK on its own.  All matrices are column-major float64 in
 * device memory with the given leading dimensions; sizes must be multiples of 128. */
/* Host-only: the tile list of one MFMA GEMM launch exactly as the kernel enumerates it (XCD-balanced runs,
 * triangular skipping, strided rows, longest-first and L2-aware strip orders); out = (block, tm, tn) triples,
 * *grid = the launch's grid size; returns the number of triples. */
int64_t gmb_debug_tile_list(int32_t mt, int32_t nt, int32_t bm, int32_t bn, int32_t k, int32_t tri,
                            int32_t tri_off, int32_t nblk_stride, int32_t klo_n, int32_t khi_n, int32_t order,
                            int32_t strip, int32_t* out, int64_t cap, int32_t* grid);
/* Host-only: the tiles of one covariance-build launch exactly as cov_tile_kernel enumerates them (strips of `strip`
 * tiles of a tile row per workgroup; tri_grid: lower triangle of a ti x tj tile grid; row_stride > 0: one rank's
 * block rows row_first, row_first + row_stride, ...; otherwise the ti x tj rectangle, XCD-remapped unless
 * keep_order); out = (block, tile row, tile column) triples, *grid = the launch's grid size; returns the count. */
int64_t gmb_debug_cov_grid(int32_t ti, int32_t tj, int32_t strip, int32_t tri_grid, int32_t row_first,
                           int32_t row_stride, int32_t keep_order, int32_t* out, int64_t cap, int64_t* grid);
/* Host-only: ticket t of the persistent tile Cholesky (csrc/chol_tiles.hpp; column-major over the lower block triangle of
 * an nrt x nct block grid) -> its tile (*I, *J); returns the number of tasks of that grid (t out of range: I = J = -1). */
int64_t gmb_debug_chol_task(int32_t t, int32_t nct, int32_t nrt, int32_t* I, int32_t* J);
/* Persistent tile Cholesky: switch the per-task wall-clock stamps of the FOLLOWING factorisations on (1) / off (0) / leave
 * them (-1) and, when `out` is not NULL, copy the stamps of the last one: 4 uint64 per task -- ticks of the 100 MHz wall
 * clock at: ticket taken / contraction done / solve input ready / tile published -- tasks in ticket order, at most
 * cap_tasks of them.  Returns the number of tasks of the last tile factorisation (0: the last factorisation used another
 * schedule or recorded nothing), or a negative gmb_status. */
int64_t gmb_chol_task_trace(gmb_engine* e, int32_t enable, uint64_t* out, int64_t cap_tasks);
/* Fault injection for the tests: the NEXT persistent tile factorisation starts its ticket counter at `n`, i.e. the tiles of
 * tickets 0 .. n-1 are never computed and every other task ends up waiting for one of them.  The launch must notice (bounded
 * waits: a few seconds), drain, and gmb_factorize must return GMB_EHIP -- never hang the GPU.  One shot. */
int gmb_debug_chol_lose_tickets(gmb_engine* e, int32_t n);
/* Timing tools only (tools/gpu_dist_emulate.py): declare the last factorisation -- single-engine, replicated or capacity --
 * valid whatever it produced, so that the gradient / prediction passes can be TIMED behind a factorisation that ran on a fake
 * transport (garbage numbers, the real launch and collective pattern) and ended with GMB_ENOTPD.  Their results are garbage
 * too.  GMB_EINVAL when no factorisation was attempted since the last gmb_set_theta. */
int gmb_debug_assume_factored(gmb_engine* e);
/* Schedule of the Cholesky for the following factorisations: -1 = by size (default), 0 = plain recursion, launches only,
 * 2 = masked look-ahead, 3 = persistent tile kernel, 4 = plain recursion whose bottom panels (<= 16 block columns with every
 * row below them) are single launches of the tile kernel -- what matrices beyond the tile kernel's range (224 block columns)
 * take by size since ABI 10.  Returns the previous setting PLUS ONE (0 = by size, 1 = recursion, 3, 4, 5), so that no valid
 * answer collides with a negative gmb_status.
 * (gmb_debug_assume_factored, ABI 10: GMB_EINVAL unless GUMBI_HIP_DEBUG_DOORS=1 is set in the environment.) */
int gmb_set_chol_scheme(gmb_engine* e, int32_t scheme);
/* Schedule of the gradient's inverse and Sigma^-1 for the following evaluations: -1 = by size (default: tile tasks for the
 * matrices the tile Cholesky factors, fused into the factorisation's launch by gmb_evaluate), 0 = the launch tree (recursive
 * block inversion, GEMM launches), 1 = tile tasks in a launch of their own behind the factorisation, 2 = fused whenever the
 * tile Cholesky runs.  `lag` >= 0: in the fused launch the inverse's tasks of block column c - lag follow the factorisation's
 * tasks of column c in the ticket order (negative: keep the current value).  Returns the previous scheme plus one, or a
 * negative gmb_status. */
int gmb_set_grad_scheme(gmb_engine* e, int32_t scheme, int32_t lag);
/* Host-only: the task list of the persistent evaluation launch for an nrt x nct block grid (csrc/eval_tiles.hpp):
 * word = kind << 30 | I << 15 | J, kind 0 = Cholesky tile (I, J), 1 = tile (I, J) of U = L^-T, 2 = tile (I, J) of
 * Sigma^-1; `with_chol` = the factorisation's tasks are part of the launch, `lag` = how many block columns the inverse
 * follows the factorisation by.  Writes at most `cap` words, returns the number of tasks. */
int64_t gmb_debug_eval_tasks(int32_t nct, int32_t nrt, int32_t with_chol, int32_t lag, uint32_t* out, int64_t cap);
/* (ABI 9) `with_chol` bit 1 of gmb_debug_eval_tasks: the Sigma^-1 tasks in PAIRS -- a task word whose J field carries 0x4000 stands
 * for the two tiles (I, J) and (I, J + 1), J + 1 <= I, computed in one pass over block row I of U (128 x 256: the task's own
 * operand row is streamed once for two tiles).  gmb_set_eval_pairs: -1 = by size (default: matrices of at least 72 block
 * columns), 0 = never, 1 = always; same results to the bit either way; returns the previous mode plus one, or a negative
 * gmb_status. */
int gmb_set_eval_pairs(gmb_engine* e, int32_t mode);
/* How gmb_predict forms A^T = K(X*, X) L^-T (pm.gp.Marginal.predict's `solve_lower`, pymc/GP.py:845-847).  Behind a gradient
 * evaluation on the tile path (N <= ~28k: gmb_evaluate, or gmb_nlml with a gradient) the engine still holds U = L^-T of the
 * resident factor; the prediction is then ONE product with a triangular operand (the GEMM that runs at 0.93 of the f64 matrix
 * peak) instead of a triangular solve (0.73 at N = 10k).  -1 = that whenever U is there (default), 0 = always solve, 1 = as -1.
 * Results agree with the solve to ~cond(L) eps (tests: <= 1e-10 relative on the golden cases).  Returns the previous mode + 1. */
int gmb_set_predict_form(gmb_engine* e, int32_t form);
/* (ABI 10) Allocate now what the following calls would allocate on first use: the factor buffer (8 Nr Np bytes: 80 GB at
 * N = 100k), with `gradient` != 0 the Sigma^-1 buffer of gmb_nlml / gmb_evaluate (8 Np^2 bytes), with M > 0 the workspaces of
 * gmb_predict for M points.  Needs gmb_set_data + gmb_set_kernel + gmb_set_theta.  No reference counterpart: a caller that times a
 * first call (bench.py's one-step side figures) takes the tens of gigabytes of hipMalloc out of it. */
int gmb_reserve(gmb_engine* e, int32_t gradient, int64_t M);
/* The covariance build alone: what gmb_factorize factors -- the lower-triangle 128 x 128 tiles of
 * Sigma = K + noise + jitter (pymc/GP.py:580), row N = y, identity padding -- written column-major into `out`
 * (device memory; ceil((N+1)/128)*128 rows x ceil(N/128)*128 columns, leading dimension ldo >= the row count).
 * Needs data, kernel and theta; the resident factor is left alone.  Entries above the tile diagonal are not written. */
int gmb_blk_covariance(gmb_engine* e, double* out, int64_t ldo);
/* Factor one 128 x 128 diagonal block in place (lower triangle; columns >= nvalid are identity
 * padding and are left alone).  dinv16 (optional, 8 x 256 doubles) receives the column-major
 * inverses of the eight 16 x 16 diagonal sub-blocks of the identity-padded factor: the operands
 * gmb_blk_trsm and gmb_blk_invert need.  *logdet_accum += sum log L_cc; *info = first bad pivot. */
int gmb_blk_potrf(gmb_engine* e, double* Akk, int64_t lda, int32_t nvalid, double* dinv16,
                  double* logdet_accum, int32_t* info);
/* invLkk (128 x 128 column-major, identity-padded) = inverse of a block factored by gmb_blk_potrf. */
int gmb_blk_invert(gmb_engine* e, const double* Lkk, int64_t lda, int32_t nvalid,
                   const double* dinv16, double* invLkk);
/* B <- B inv(Lkk)^T in place: B is nrows x 128 (nrows a multiple of 16, leading dimension ldb);
 * rows / columns >= nvalid of the block act as identity whatever the buffer holds there. */
int gmb_blk_trsm(gmb_engine* e, double* B, int64_t ldb, int64_t nrows, const double* Lkk,
                 int64_t ldl, const double* dinv16, int32_t nvalid);
/* C[n + m*ldc] = beta*C + alpha * sum_k A[m + k*lda] * B[n + k*ldb];  tri != 0 skips tiles that
 * lie strictly above the diagonal of the (n, m) index space shifted by tri_shift tiles. */
int gmb_blk_gemm_nt(gmb_engine* e, double* C, int64_t ldc, const double* A, int64_t lda,
                    const double* B, int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha,
                    double beta, int32_t tri, int64_t tri_shift);
/* -- ONE GP over the GPUs of a node (no reference counterpart -- the reference is
 * single-process; the work replaced is what pm.find_MAP / Marginal.predict do per evaluation,
 * pymc/GP.py:811, 845-847) -------------------------------------------------------------------------------
 * One process per GPU, every process holds an engine with the SAME data / kernel / theta.  128-row blocks
 * of the covariance matrix are dealt round-robin over the ranks (1-D block-cyclic rows); the native driver
 * (gumbi_amd/csrc/dist_driver.hpp) runs the panel loop with look-ahead on this rank's HIP streams and
 * exchanges data through ONE collective, an all-gather of float64 buffers in device memory:
 *   all_gather(ctx, send, recv, count, stream): every rank contributes `count` doubles at `send`; `recv`
 *   receives world * count doubles in rank order; the operation is ordered on the hipStream_t `stream`
 *   (enqueue it there, or synchronise the stream, move the data and return).  0 = success.
 * gmb_rccl_comm_create gives the production transport (RCCL over xGMI: ncclAllGather on a communicator of
 * its own; librccl is resolved with dlopen -- pass the path of the copy the process already uses, or NULL
 * for the loader's); callers may supply any other transport with the same contract (the tests use gloo). */
typedef struct gmb_comm {
  int32_t rank, world;
  void* ctx;
  int32_t (*all_gather)(void* ctx, const void* send, void* recv, int64_t count, void* stream);
} gmb_comm;

int gmb_rccl_unique_id(const char* librccl_path, void* id128);  /* rank 0: 128 bytes to hand to every rank */
int gmb_rccl_comm_create(gmb_comm** out, const char* librccl_path, const void* id128, int32_t rank,
                         int32_t world, int32_t device);         /* collective: ncclCommInitRank          */
void gmb_rccl_comm_destroy(gmb_comm* c);
int gmb_rccl_comm_ranks(const gmb_comm* c); /* ncclCommCount of the communicator (what RCCL itself reports) */
/* (ABI 10) 1 when the transport holds TWO communicators over the same ranks (the second split off the first with ncclCommSplit):
 * RCCL executes the collectives of one communicator in issue order whatever their streams, so the driver's collectives on the
 * communication stream (a panel column's TAIL, the gradient's chunks) would hold up the main stream's; with two, the engine's main
 * stream (declared by every gmb_dist_* call) and every other stream have their own.  0: one communicator (ncclCommSplit missing or failed, or GUMBI_RCCL_ONE_COMM set): correct, serialised. */
int gmb_rccl_comm_split(const gmb_comm* c);
const char* gmb_rccl_last_error(void);

/* The schedule one rank executes, as data (host-only, no device needed): step kinds
 *   0 KBUILD  covariance tiles of the rank's block rows
 *   1 SQUARE  all-gather the block rows [lo, hi) = [c0, c1) of the panel's diagonal square (`elems` doubles
 *             per rank: maxcount * 128 rows x (c1 - c0) * 128 columns), then factor it locally
 *   6 SOLVE   solve the rank's block rows in [lo, hi) = [c1, nrt) of the panel against the square (no communication)
 *   2 PANEL   all-gather the HEAD of the solved panel column: block rows [lo, hi) = [c1, min(c1 + w, nrt)) -- the rows of the next
 *             panel's square, all the chain needs from the other ranks -- on the main stream (`elems` as for SQUARE)
 *   7 TAIL    all-gather the rest of the panel column, block rows [lo, hi) = [c1 + w, nrt), on the communication stream
 *             (`stream` = 2) beside the next panel's chain; only the bulk update reads it          (6, 7: since ABI 10)
 *   3 UPDATE  A[rows, lo:hi] -= L[rows, c0:c1] L[lo:hi, c0:c1]^T on the rank's block rows >= lo, on stream
 *             `stream` (0 main, 1 bulk)
 *   4 FORK    the bulk stream waits for everything issued on the main stream so far (and for the TAIL issued since the last
 *             FORK);  5 JOIN  the main stream waits for the bulk stream
 * (first, count): the rank's block rows first, first + world, ... of the step's row range. */
typedef struct gmb_dist_step {
  int32_t op, c0, c1, lo, hi, first, count, maxcount, stream;
  int64_t elems;
} gmb_dist_step;
/* number of steps (the first min(cap, n) are written to out; out may be NULL) or a negative gmb_status;
 * panel_blocks <= 0 selects the default panel width */
int64_t gmb_dist_plan(int64_t N, int32_t rank, int32_t world, int32_t panel_blocks, gmb_dist_step* out,
                      int64_t cap);

/* How the ranks of gmb_dist_* hold the factor.  0 (default) = REPLICATED: compute is partitioned by block rows, every rank
 * ends with the complete factor in a full-size buffer (N = 100k: 80 GB per rank whatever the number of GPUs).  1 = CAPACITY
 * (csrc/dist_capacity.hpp): a rank keeps only its own block rows + two column-panel buffers; the gradient and the prediction
 * stream L through again, panel by panel from its owners -- "N beyond one GPU" (N = 200k over 8 GPUs: ~160 GB per rank).
 * Same results (other grouping of some sums).  Takes effect with the next gmb_dist_factorize; returns the previous mode or a
 * negative gmb_status.  gmb_copy_factor is not available on a capacity-mode engine (no rank holds the factor). */
int gmb_dist_set_mode(gmb_engine* e, int32_t mode);
/* Device bytes the engine holds through its own allocations right now (peak != 0: the largest value so far). */
int64_t gmb_resident_bytes(const gmb_engine* e, int32_t peak);

/* gmb_factorize over comm->world ranks (collective: every rank calls it with the same state).  Every rank
 * ends with the complete factor (replicated mode; its own block rows in capacity mode), v, log-det and -- on failure -- the
 * same GMB_ENOTPD / gmb_notpd_index. */
int gmb_dist_factorize(gmb_engine* e, const gmb_comm* comm, int32_t panel_blocks);
/* gmb_nlml over the ranks; with grad != NULL the inverse, Sigma^-1 and the trace reductions are
 * partitioned by block rows and every rank receives bit-identical (nlml, grad). */
int gmb_dist_nlml(gmb_engine* e, const gmb_comm* comm, double* nlml, double* grad);
/* gmb_predict with the M points sharded over the ranks; every rank passes the same Xs and receives the
 * complete mean / var. */
int gmb_dist_predict(gmb_engine* e, const gmb_comm* comm, const double* Xs, int64_t M, int64_t ldxs,
                     int32_t with_noise, double* mean, double* var, int32_t memspace);

#ifdef __cplusplus
}
#endif
#endif /* GUMBI_HIP_H */
