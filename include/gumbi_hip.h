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

#define GMB_ABI_VERSION 3
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

typedef struct gmb_timings { /* milliseconds on the engine's HIP stream (hipEvent pairs) */
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
} gmb_timings;

typedef struct gmb_engine gmb_engine;

/* -- lifecycle ------------------------------------------------------------------------------ */

/* Number of visible HIP devices, or a negative gmb_status.  Never falls back to the CPU. */
int gmb_device_count(void);
int gmb_abi_version(void);

/* Create an engine bound to HIP device `device`.  `stream` is an existing hipStream_t to launch
 * on (e.g. torch's current stream), or NULL to let the engine create its own. */
int gmb_create(gmb_engine** out, int32_t device, void* stream);
void gmb_destroy(gmb_engine* e);
const char* gmb_last_error(const gmb_engine* e); /* valid until the next call on `e` */
void* gmb_stream(const gmb_engine* e);           /* the hipStream_t launches go to   */

/* -- model definition ------------------------------------------------------------------------
 * gmb_set_data replaces `X, y = self.get_shaped_data("mean")` being captured by the PyMC model
 * (pymc/GP.py:521, 580): X is (N, D) with leading dimension ldx (>= D), y is (N,).  Copies in. */
int gmb_set_data(gmb_engine* e, const double* X, int64_t N, int32_t D, int64_t ldx,
                 const double* y, int32_t memspace);

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

/* Negative log marginal likelihood  N/2 log 2pi + sum log L_ii + |v|^2/2  of the last
 * factorisation, and (if grad != NULL, length gmb_theta_size) its gradient w.r.t. natural-scale
 * theta: 1/2 tr((Sigma^-1 - a a^T) dSigma/dtheta).  Priors and Jacobians are added by the Python
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

/* Multi-GPU gradient (every rank holds the complete factor after the block-cyclic Cholesky of
 * gumbi_amd/distributed.py): rank `shard` of `nshards` computes L^-1 (replicated), its block rows
 * shard, shard + nshards, ... of Sigma^-1 and the trace reductions over those rows, and returns
 * the raw accumulators (gmb_grad_acc_size() doubles, host).  They are plain sums over tiles: the
 * driver all-reduces them and every rank finishes with gmb_nlml_from_acc (chain rule to the packed
 * parameters, same layout as gmb_nlml's grad).  gmb_nlml(e, &v, grad) == shard 0 of 1 + from_acc. */
/* Fully partitioned variant: gmb_inv_rows gives a rank ITS block rows first, first+stride, ... of
 * U = L^-T (V: device, (owned*128) x Np column-major with leading dimension ldv; the predict solve
 * applied to rows of the identity, N^3/G flops) and its rows of alpha = U v (device, owned*128
 * doubles); the driver all-gathers both into the upper triangle of every rank's factor buffer
 * (gmb_factor_buffers) and into the alpha vector (gmb_grad_buffers), then gmb_nlml_shard_u does
 * the rank's share of Sigma^-1 and of the reductions without inverting anything itself. */
int gmb_inv_rows(gmb_engine* e, int32_t first, int32_t stride, double* V, int64_t ldv, double* alpha_rows);
int gmb_grad_buffers(gmb_engine* e, void** alpha);
int gmb_nlml_shard_u(gmb_engine* e, int32_t shard, int32_t nshards, double* acc, int32_t nacc);
int32_t gmb_grad_acc_size(void);
int gmb_nlml_shard(gmb_engine* e, int32_t shard, int32_t nshards, double* acc, int32_t nacc);
int gmb_nlml_from_acc(gmb_engine* e, const double* acc, int32_t nacc, double* nlml, double* grad);

/* -- introspection --------------------------------------------------------------------------- */
int gmb_set_profiling(gmb_engine* e, int32_t on); /* per-launch hipEvent timing of GEMMs */
int gmb_timings_get(const gmb_engine* e, gmb_timings* out);

/* Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate of `device` in TFLOP/s (register-only
 * kernel, no memory traffic) -- the measured denominator for the MFMA roofline -- and the shader
 * cycles one wave spends per MFMA (64 = the datasheet issue rate; wall rate / this = clock). */
int gmb_mfma_f64_peak(int32_t device, double* tflops, double* cycles_per_mfma);

/* Copy out pieces of the resident state (tests / multi-GPU driver):
 *   rows [r0, r0+nr) x cols [c0, c0+nc) of the factor buffer (lower triangle meaningful),
 *   written row-major into `out` (nr*nc doubles, host). */
int gmb_copy_factor(const gmb_engine* e, int64_t r0, int64_t nr, int64_t c0, int64_t nc,
                    double* out);
int gmb_copy_v(const gmb_engine* e, double* out); /* v = L^-1 y, length N, host */
/* alpha = Sigma^-1 y (length N, host); exists after gmb_nlml with a gradient.  Used by the
 * Kronecker (ICM) multi-output path, whose outer chain rule needs it (gumbi_amd/regression/icm.py). */
int gmb_copy_alpha(const gmb_engine* e, double* out);

/* -- block-level operations for the multi-GPU driver (device pointers, column-major) ----------
 * These expose the same kernels the single-GPU path uses so that a 1-D block-cyclic row
 * partition can be driven from one process per GPU with the panel
 * broadcast done by RCCL through torch.distributed.  All matrices are column-major float64 in
 * device memory with the given leading dimensions; sizes must be multiples of 128. */
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
/* As gmb_blk_gemm_nt, for one rank of the block-cyclic row partition: the n index walks 128-row
 * blocks that are `nblk_stride` blocks apart in memory (B and C both), and `tri` skips tiles whose
 * last row (n offset + tri_off) lies above their first column (m offset). */
int gmb_blk_gemm_strided(gmb_engine* e, double* C, int64_t ldc, const double* A, int64_t lda,
                         const double* B, int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha,
                         double beta, int32_t tri, int64_t tri_off, int32_t nblk_stride);
/* Gather (to_packed = 1) / scatter (0) `count` 128 x 128 blocks between block rows
 * mat + t*stride_blocks*128 (t < count, leading dimension ld) and a packed (count*128) x 128
 * column-major buffer (leading dimension ldp) -- the send / receive side of the panel all-gather. */
int gmb_blk_pack(gmb_engine* e, double* mat, int64_t ld, int64_t stride_blocks, int32_t count,
                 double* packed, int64_t ldp, int32_t to_packed);
/* Device pointers of the engine's own resident state, for a driver that runs the factorisation
 * itself (multi-GPU): factor buffer (Nr x Np column-major, leading dimension ld), the
 * ceil(N/128) x 8 x 256 sub-block inverses of the diagonal blocks (see gmb_blk_potrf), the scalar
 * slots ([0] log-det accumulator) and the info word. */
int gmb_factor_buffers(gmb_engine* e, void** A, int64_t* ld, int64_t* Nr, int64_t* Np,
                       void** dinv16, void** scal, void** info);
/* Bracket an externally driven factorisation: begin resets the accumulators; finish takes the
 * global log-det and info (after the driver's reductions), extracts v and marks the engine
 * factorised so that gmb_nlml / gmb_predict work on the resident factor. */
int gmb_begin_external_factorization(gmb_engine* e);
/* this rank's partial log-det (sum over the diagonal blocks it factored) and failure word */
int gmb_local_logdet_info(gmb_engine* e, double* logdet, int64_t* info);
int gmb_finish_external_factorization(gmb_engine* e, double logdet, int64_t info);
/* Covariance tile rows [i0,i0+ni) x cols [j0,j0+nj) of Sigma (+noise/jitter on the diagonal)
 * into a column-major buffer: out[(i-i0) + (j-j0)*ldo]. */
int gmb_blk_kbuild(gmb_engine* e, double* out, int64_t ldo, int64_t i0, int64_t ni, int64_t j0,
                   int64_t nj);

#ifdef __cplusplus
}
#endif
#endif /* GUMBI_HIP_H */
