// covariance.hpp -- pairwise-distance covariance tiles (RBF / Matern family, ARD), the linear
// and coregion terms Gumbi multiplies in, the white-noise diagonal, and the reductions that turn
// the solved cross-covariance into posterior mean / variance.
//
// Restates what PyMC evaluates for the model Gumbi declares
// (gumbi/regression/pymc/GP.py:389-414 continuous kernel, :449-455 linear, :457-464 coregion,
//  :560-569 noise, :711-729 composition; prediction :845-847):
//   k_cont = eta^2 * f(r),  r^2 = sum_k ((x_k - x'_k) / ls_k)^2,  r = sqrt(r^2 + 1e-12)
//   ExpQuad exp(-r^2/2) | Matern52 (1+sqrt5 r+5r^2/3)exp(-sqrt5 r) | Matern32 (1+sqrt3 r)exp(-sqrt3 r)
//   Matern12 exp(-r) | Exponential exp(-r/2)
//   K = (k_cont + tau * <x_l - c, x'_l - c>) * prod_t B_t[cat_t(x), cat_t(x')]
//   Sigma = K + diag(sigma^2 * nmult[cat_out(x)]) + jitter * I
// r^2 is formed from scaled differences directly (PyMC expands it as -2XX'^T+|X|^2+|X'|^2 and
// clips; the two agree to O(1e-16 |x|^2), see oracle/gp_oracle.py square_dist).
//
// HBM-bound: each covariance entry is written once (8 B) and never re-read by this kernel; the
// inputs of a 128 x 128 tile (2 x 128 points) are staged once in LDS.  Lanes run along the
// contiguous (row) index, so every wave store is one 512 B segment.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_f64.hpp"

#ifndef GMB_KB_PROBE
#define GMB_KB_PROBE 0  // 1 / 2: measurement probes of the interior covariance tiles (stores only / arithmetic only)
#endif

namespace gmb {

constexpr int MAX_TABS = 5;     // GMB_MAX_COREG + output coregion
constexpr int MAX_LIN = 8;      // GMB_MAX_LIN

// Pre-processed coordinates of a point set (structure-of-arrays, padded length `npad`):
//   xs[k*npad + i]  continuous dims scaled by 1/ls_k (zero for padded dims / points), k < nc_pad;
//                   row nc_pad holds the squared norm of the scaled point (so xs has nc_pad + 1 <= 17 rows)
//   xl[k*npad + i]  linear dims minus c_k
//   cat[t*npad + i] category index per coregion table
struct PointSet {
  const double* xs;
  const double* xl;
  const int32_t* cat;
  int64_t n;     // real points
  int64_t npad;  // allocated / padded length (multiple of 128)
};

struct CovParams {
  int32_t kind;
  int32_t n_lin;
  int32_t n_tab;
  int32_t tab_levels[MAX_TABS];
  int32_t tab_off[MAX_TABS];  // offset of table t in `tabs`
  const double* tabs;         // B tables, row-major L x L each
  double eta2, tau;
  double sigma2, jitter;
  int32_t noise_tab;          // index into cat[] giving the output level for noise, or -1
  const double* noise_mult;   // diag(B_noise) per output level (device), used if noise_tab >= 0
};

// exp(x) for x <= 0 -- all a stationary covariance ever exponentiates.  Cody-Waite reduction
// x = k ln2 + r (|r| <= ln2 / 2), degree-13 Taylor polynomial (truncation 4e-18), v_ldexp_f64.
// Within 1-2 ulp of libm's exp; about two thirds of its instructions (no overflow / NaN paths).
__device__ __forceinline__ double exp_nonpos(double x) {
  // underflow (and -inf) without a branch: clamp to where the result is the smallest subnormal (v_ldexp then
  // rounds to 5e-324, i.e. zero for every use here); NaN compares false and stays NaN
  x = x < -745.0 ? -745.0 : x;
  const double k = __builtin_rint(x * 1.4426950408889634074);
  double r = fma(-k, 6.93147180369123816490e-01, x);
  r = fma(-k, 1.90821492927058770002e-10, r);
  double p = 1.6059043836821614599e-10;        // 1/13!
  p = fma(p, r, 2.0876756987868098979e-09);    // 1/12!
  p = fma(p, r, 2.5052108385441718775e-08);    // 1/11!
  p = fma(p, r, 2.7557319223985890653e-07);    // 1/10!
  p = fma(p, r, 2.7557319223985892511e-06);    // 1/9!
  p = fma(p, r, 2.4801587301587301566e-05);    // 1/8!
  p = fma(p, r, 1.9841269841269841253e-04);    // 1/7!
  p = fma(p, r, 1.3888888888888889419e-03);    // 1/6!
  p = fma(p, r, 8.3333333333333332177e-03);    // 1/5!
  p = fma(p, r, 4.1666666666666664354e-02);    // 1/4!
  p = fma(p, r, 1.6666666666666665741e-01);    // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)k);
}

// The covariance build at C3 (N = 50k, d = 8, Matern-5/2) is bound by fp64 VALU issue, not by HBM: ~75
// instructions per entry with libm's exp / sqrt (special-case handling included) against ~1.6 ms of store
// time for the 10 GB triangle.  exp_nonpos / sqrt_pos drop the paths a stationary kernel can never take
// (-DGMB_LIBM_EXP restores libm for A/B runs).
#ifdef GMB_LIBM_EXP
#define GMB_EXP(x) exp(x)
#define GMB_SQRT(x) sqrt(x)
#else
#define GMB_EXP(x) exp_nonpos(x)
#define GMB_SQRT(x) sqrt_pos(x)
#endif

// sqrt(p) for p in the normal range (here p = r^2 + 1e-12 >= 1e-12): hardware rsq estimate, one coupled
// Goldschmidt step and one residual correction -- correctly rounded in all but ~1e-3 of cases, <= 1 ulp
// otherwise; libm's version spends as many instructions again on scaling subnormals and on 0 / inf.
__device__ __forceinline__ double sqrt_pos(double p) {
  const double y = __builtin_amdgcn_rsq(p);
  double g = p * y;
  const double h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  const double hh = fma(h, r, h);
  const double d = fma(-g, g, p);
  return fma(d, hh, g);
}

template <int KIND>
__device__ __forceinline__ double stationary(double r2) {
  if constexpr (KIND == 0) {
    return GMB_EXP(-0.5 * r2);
  } else {
    const double r = GMB_SQRT(r2 + 1e-12);
    if constexpr (KIND == 1) {
      const double s5 = 2.23606797749978969641;
      return (1.0 + s5 * r + (5.0 / 3.0) * (r * r)) * GMB_EXP(-s5 * r);
    } else if constexpr (KIND == 2) {
      const double s3 = 1.73205080756887729353;
      return (1.0 + s3 * r) * GMB_EXP(-s3 * r);
    } else if constexpr (KIND == 3) {
      return GMB_EXP(-r);
    } else {
      return GMB_EXP(-0.5 * r);
    }
  }
}

// --- interior tiles: squared distances on the matrix pipe -------------------------------------------
// exp(x) for FINITE x <= ~0 (interior tiles only, whose 256 points were checked finite): the reduction's
// integer part comes out of the mantissa of x*log2(e) + 1.5*2^52 (no v_rndne / v_cvt), and 2^k goes straight
// into the exponent field (x >= -700 keeps the result normal, so no v_ldexp).
__device__ __forceinline__ double exp_interior(double x) {
  x = fmax(x, -700.0);
  const double magic = 6755399441055744.0;  // 1.5 * 2^52
  const double t = fma(x, 1.4426950408889634074, magic);
  const double k = t - magic;
  double r = fma(-k, 6.93147180369123816490e-01, x);
  r = fma(-k, 1.90821492927058770002e-10, r);
  // degree-10 minimax polynomial of exp on |r| <= 0.34665 with p(0) = p'(0) = 1 (weighted Remez in 60-digit arithmetic,
  // coefficients rounded to double): 2.9e-16 relative before rounding, <= 4 ulp in double Horner form against long-double
  // exp.  Interior tiles only: their r^2 comes from the expansion |x|^2 + |x'|^2 - 2 x.x', whose own rounding
  // (~eps |x/ls|^2 absolute in r^2, i.e. tens of ulp in k) is what bounds the entry -- the degree-11 polynomial that kept
  // exp itself below 0.98 ulp (round 2) bought nothing there and cost one more issue slot per entry.  The direct-form
  // tiles (diagonal, boundary, kinked kernels) keep exp_nonpos.
  double p = 0x1.2707a770dc38cp-22;
  p = fma(p, r, 0x1.72e91aefc6956p-19);
  p = fma(p, r, 0x1.a01b7c4deaf70p-16);
  p = fma(p, r, 0x1.a0198d585c94ap-13);
  p = fma(p, r, 0x1.6c16c0c831ce8p-10);
  p = fma(p, r, 0x1.11111125b3e47p-7);
  p = fma(p, r, 0x1.55555555890bfp-5);
  p = fma(p, r, 0x1.55555555507c5p-3);
  p = fma(p, r, 0x1.ffffffffffed2p-2);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  const int hi = __double2hiint(p) + (__double2loint(t) << 20);
  return __hiloint2double(hi, __double2loint(p));
}

// sqrt(p), p >= 1e-12 finite: hardware rsq estimate (~26 bits) and ONE coupled Goldschmidt step -- ~2 ulp.  The residual
// correction of sqrt_pos (two more slots) would bring it to <= 1 ulp; on interior tiles p already carries the
// expansion's rounding (see exp_interior), which is an order of magnitude above that.
__device__ __forceinline__ double sqrt_interior(double p) {
  const double y = __builtin_amdgcn_rsq(p);
  const double g = p * y;
  const double h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  return fma(g, r, g);
}

// eta2 * k(r2) from the contraction's output q: q = -r2/2 for ExpQuad (the -1/2 rides in the operands),
// q = r2 otherwise.  Rounding of the expansion |a|^2 + |b|^2 - 2ab can leave q a few ulp(|a|^2) on the wrong
// side of zero: clamped, as PyMC clips its own expansion (pm.gp.cov.Stationary.square_dist).
template <int KIND>
__device__ __forceinline__ double stationary_interior(double q, double eta2) {
  if constexpr (KIND == 0) {
    return eta2 * exp_interior(fmin(q, 0.0));
  } else {
    const double p = fmax(q, 1e-12);  // q = r^2 + 1e-12 already (the offset rides in the accumulator's row norm), so this
                                      // is max(r^2, 0) + 1e-12: PyMC's clip and its sqrt(r^2 + 1e-12) in one slot
    const double r = sqrt_interior(p);
    if constexpr (KIND == 1) {
      const double s5 = 2.23606797749978969641;
      const double poly = fma(eta2 * (5.0 / 3.0), p, fma(eta2 * s5, r, eta2));
      return poly * exp_interior(-s5 * r);
    } else if constexpr (KIND == 2) {
      const double s3 = 1.73205080756887729353;
      return fma(eta2 * s3, r, eta2) * exp_interior(-s3 * r);
    } else if constexpr (KIND == 3) {
      return eta2 * exp_interior(-r);
    } else {
      return eta2 * exp_interior(-0.5 * r);
    }
  }
}

// d k / d r2 (for the NLML gradient)
template <int KIND>
__device__ __forceinline__ double stationary_dr2(double r2) {
  if constexpr (KIND == 0) {
    return -0.5 * GMB_EXP(-0.5 * r2);
  } else {
    const double r = GMB_SQRT(r2 + 1e-12);
    if constexpr (KIND == 1) {
      const double s5 = 2.23606797749978969641;
      return -(5.0 / 6.0) * (1.0 + s5 * r) * GMB_EXP(-s5 * r);
    } else if constexpr (KIND == 2) {
      const double s3 = 1.73205080756887729353;
      return -1.5 * GMB_EXP(-s3 * r);
    } else if constexpr (KIND == 3) {
      return -GMB_EXP(-r) / (2.0 * r);
    } else {
      return -0.25 * GMB_EXP(-0.5 * r) / r;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// point preparation: raw row-major X (n x D) -> PointSet arrays
// ---------------------------------------------------------------------------------------------
struct PrepArgs {
  const double* X;
  int64_t n, ldx, npad;
  int32_t nc, nc_pad;            // real / padded continuous dims
  int32_t idx_cont[16];
  double inv_ls[16];
  int32_t n_lin;
  int32_t idx_lin[MAX_LIN];
  double c_lin[MAX_LIN];
  int32_t n_tab;
  int32_t tab_col[MAX_TABS];
  int32_t tab_levels[MAX_TABS];
  double* xs;
  double* xl;
  int32_t* cat;
  // gmb_evaluate's first launch also clears what the evaluation's later launches expect zeroed (the engine's scalar block, the
  // tile launch's control words and flags) -- two memsets less per evaluation; 8-byte words, null = nothing
  unsigned long long* zero[2];
  int64_t zero_words[2];
};

__device__ __forceinline__ void prep_one_point(const PrepArgs& a, const int64_t i);

__global__ void prep_points_kernel(PrepArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int z = 0; z < 2; ++z)
    if (a.zero[z])
      for (int64_t w = i; w < a.zero_words[z]; w += (int64_t)gridDim.x * blockDim.x) a.zero[z][w] = 0ull;
  if (i >= a.npad) return;
  prep_one_point(a, i);
}

// prepared coordinates of point i (i < npad): scaled continuous dims + |x / ls|^2, centred linear dims, table indices
__device__ __forceinline__ void prep_one_point(const PrepArgs& a, const int64_t i) {
  const bool real = i < a.n;
  const double* row = a.X + i * a.ldx;
  double nrm = 0.0;
  for (int k = 0; k < a.nc_pad; ++k) {
    double v = 0.0;
    if (real && k < a.nc) v = row[a.idx_cont[k]] * a.inv_ls[k];
    a.xs[(int64_t)k * a.npad + i] = v;
    nrm = fma(v, v, nrm);
  }
  a.xs[(int64_t)a.nc_pad * a.npad + i] = nrm;  // row nc_pad: |x / ls|^2 (the interior tiles' augmented coordinate)
  for (int k = 0; k < a.n_lin; ++k)
    a.xl[(int64_t)k * a.npad + i] = real ? row[a.idx_lin[k]] - a.c_lin[k] : 0.0;
  for (int t = 0; t < a.n_tab; ++t) {
    int c = 0;
    if (real) {
      c = (int)row[a.tab_col[t]];  // PyMC Coregion casts the coordinate to int32
      c = c < 0 ? 0 : (c >= a.tab_levels[t] ? a.tab_levels[t] - 1 : c);
    }
    a.cat[(int64_t)t * a.npad + i] = c;
  }
}

// ---------------------------------------------------------------------------------------------
// covariance tile kernel
// ---------------------------------------------------------------------------------------------
enum CovMode : int { COV_TRAIN = 0, COV_CROSS = 1 };

struct CovTileArgs {
  CovParams p;
  PointSet rows;   // fast (contiguous) index of the output
  PointSet cols;   // slow index
  double* out;     // out[(i - i0) + (j - j0) * ldo]
  int64_t ldo;
  int64_t i0, j0;  // global offsets of the region (multiples of 128)
  int32_t ti, tj;  // tile counts
  int32_t mode;    // CovMode
  int32_t lower_only;   // COV_TRAIN: skip tiles strictly above the diagonal
  const double* y;      // COV_TRAIN: observations, written into row n (the "y row")
  int32_t tri_grid;     // the grid enumerates only the tiles on / below the diagonal (i0 == j0 == 0)
  // Additive models (sum of kernels): passes after the first ADD their term to the real entries and
  // leave padding, y row and the noise diagonal (written by the first pass) alone.
  int32_t accumulate;
  // One rank's block rows of the lower triangle (multi-GPU K-build, i0 == j0 == 0, lower_only): the grid
  // enumerates, block row by owned block row row_first, row_first + row_stride, ... (< ti), the tiles
  // tj = 0 .. min(row, tj - 1).  row_stride == 0: off.
  int32_t row_first, row_stride;
  // rows_packed (with row_first / row_stride): `out` holds ONLY the owned block rows, packed -- block row
  // row_first + t * row_stride sits at rows 128 t .. of `out` (the multi-GPU driver's capacity mode, where no rank holds the
  // whole matrix)
  int32_t rows_packed;
  int32_t stream_stores;  // interior tiles use non-temporal stores (set by launch_cov for outputs of >= 1 GiB)
  int32_t strip;          // tiles per workgroup along the column index (set by launch_cov; see cov_tile_kernel)
  // Small grids (launch_cov: <= 128 strips of ONE tile): every tile is shared by `gsplit` workgroups, each with a quarter (half)
  // of its columns -- a direct-loop tile is a serial chain of 16 entry groups per thread (~30 us whatever the matrix), and a
  // matrix of four block columns has ten tiles for 256 compute units.  1 = off.
  int32_t gsplit;
  // Small evaluations (gmb_evaluate below 4096 rows; round 6): the covariance build prepares the points itself -- every workgroup
  // the rows and the columns of ITS strip, redundantly, before it stages them (the values are a function of X and theta only, so
  // equal writes from different workgroups are harmless) -- and workgroup 0 clears what prep_points_kernel would have cleared:
  // one launch and one launch boundary (~10 us of a 205 us evaluation at N = 392) less.
  int32_t inline_prep;
  PrepArgs prep;
};

// row of `out` where the tile row that starts at global row gi0 begins
__device__ __forceinline__ int64_t cov_out_row(const CovTileArgs& a, const int64_t gi0) {
  if (a.rows_packed) return (int64_t)(((int)(gi0 / TILE) - a.row_first) / a.row_stride) * TILE;
  return gi0 - a.i0;
}

// A run of interior 128 x 128 tiles of one tile row (every row and column real, strictly below the diagonal of
// the training matrix, stationary term only, all points finite): the squared distances come off the MATRIX pipe.
// With augmented coordinates  a' = (-2 x'_1 .. -2 x'_d, 1, |x'|^2),  b' = (x_1 .. x_d, |x|^2, 1)  the whole
// r^2 = |x|^2 + |x'|^2 - 2 x.x' is ONE contraction of length d + 2 (PyMC's own expansion,
// pm.gp.cov.Stationary.square_dist; for ExpQuad the operands carry the -1/2 as well), i.e. ceil((d+2)/4)
// v_mfma_f64_16x16x4_f64 per 16 x 16 entries, and the vector instructions are left with the transcendental
// part alone: Matern-5/2, d = 8: 54 -> 33 VALU instructions per entry and no LDS traffic at all (the direct form
// spends 16 of them on differences and 4 ds_read_b128 on broadcasting the column point).  It is NOT free: an f64
// MFMA keeps the SIMD's sixteen double-precision lanes busy for its 64 cycles (the probes of tools/gpu_kbuild_ab.py:
// MFMAs + tile set-up alone take 0.72 of the 2.5 ms at N = 50k): one MFMA per 256 entries = 4 issue slots per entry.
// With the norms in the accumulator (below) and the degree-11 exp, d = 8 Matern-5/2 costs 31 + 1 + 8 = 40
// slot-equivalents against 54 + LDS: C3 3.33 -> 4.1 TB/s, C5 (ExpQuad) 4.46 -> 5.55, d = 16 2.3 -> 3.3 (wall clock of
// gmb_blk_covariance; a torch fill of the same buffer: 6.7 TB/s; stores alone in this pattern: 5.5 - 6.0).
// A wave owns 32 rows (its row operand stays in registers) and walks the strip's `ncb` blocks of 16 columns;
// the D layout (n = lane & 15 <-> row, m = (lane >> 4) + 4 reg <-> column) with the wave's two row blocks interleaved
// makes every store instruction a set of four 256 B column segments (16-byte stores).
template <int KIND, int NC, bool NT>
__device__ __forceinline__ void cov_interior_tile(const CovTileArgs& a, const int64_t gi0, const int64_t gj0, const int ncb) {
  // Where the two norm slots would cost a further MFMA (d = 4, 8, 16: ceil((d + 2) / 4) > d / 4) the norms enter
  // through the accumulator instead -- C = |x|^2 + |x'|^2 (one v_add per entry, 1 slot against the MFMA's 4),
  // contraction over the d coordinates alone; for d = 1, 2 they ride in the single MFMA's spare slots.
  constexpr bool NORM_IN_C = (NC + 2 + 3) / 4 > (NC + 3) / 4;
  constexpr int KA = NORM_IN_C ? NC : NC + 2;  // contraction length
  constexpr int NG = (KA + 3) / 4;             // MFMAs per 16 x 16 entries
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  constexpr double sc = KIND == 0 ? 1.0 : -2.0;  // scale of the column point's coordinates
  constexpr double sn = KIND == 0 ? -0.5 : 1.0;  // scale of both squared norms
  constexpr double off12 = KIND == 0 ? 0.0 : 1e-12;  // PyMC's sqrt(r^2 + 1e-12): the offset enters with the row norm
  double brow[2][NG], nrow[2];
#pragma unroll
  for (int ib = 0; ib < 2; ++ib) {
    const double* src = a.rows.xs + gi0 + 32 * wave + 2 * r16 + ib;  // slot r16 of block ib <-> row 2 r16 + ib: see the stores
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int k = 4 * g + kq;
      double v = k < (NORM_IN_C ? NC : NC + 1) ? src[(int64_t)k * a.rows.npad] : 0.0;
      if (!NORM_IN_C && k == NC) v = fma(v, sn, off12);  // the row norm's slot meets the column operand's constant 1
      if (!NORM_IN_C && k == NC + 1) v = 1.0;
      brow[ib][g] = v;
    }
    nrow[ib] = NORM_IN_C ? fma(sn, src[(int64_t)NC * a.rows.npad], off12) : 0.0;  // this lane's row (r16 of block ib)
  }
  // column operand of one 16-column block: coordinate k of the column point (augmented form: slot NC is the
  // constant 1, slot NC + 1 the norm, row NC of xs); NORM_IN_C: the norms of the lane's 4 columns kq + 4 r
  const double* csrc = a.cols.xs + gj0 + r16;
  const double* nsrc = a.cols.xs + (int64_t)NC * a.cols.npad + gj0 + kq;
  auto load_cols = [&](int jb, double (&dst)[NG], double (&nc)[4]) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int k = 4 * g + kq;
      double v;
      if constexpr (NORM_IN_C) {
        v = k < NC ? sc * csrc[(int64_t)k * a.cols.npad + 16 * jb] : 0.0;
      } else {
        v = k <= NC + 1 ? csrc[(int64_t)(k == NC + 1 ? NC : k) * a.cols.npad + 16 * jb] : 0.0;
        v *= k < NC ? sc : sn;
        if (k == NC) v = 1.0;
      }
      dst[g] = v;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) nc[r] = NORM_IN_C ? sn * nsrc[16 * jb + 4 * r] : 0.0;
  };
  const double eta2 = a.p.eta2;
  const uint64_t col_bytes = (uint64_t)a.ldo * 8u;
  char* tile = reinterpret_cast<char*>(a.out + cov_out_row(a, gi0) + (gj0 - a.j0) * a.ldo);  // uniform
  const uint32_t lane_off = (uint32_t)((32 * wave + 2 * r16) * 8) + (uint32_t)kq * (uint32_t)col_bytes;
#if GMB_KB_PROBE >= 2
  double probe_sum = 0.0;
#endif
  // A rolled loop over the `ncb` 16-column blocks of the strip (8 per tile), the next block's operands requested
  // one iteration ahead.
  double acol[NG], anext[NG], ncol[4], nnext[4];
  load_cols(0, acol, ncol);
#pragma unroll 1
  for (int jb = 0; jb < ncb; ++jb) {
    load_cols(jb + 1 < ncb ? jb + 1 : jb, anext, nnext);
    d4 acc[2];
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      if constexpr (NORM_IN_C) acc[ib] = d4{nrow[ib] + ncol[0], nrow[ib] + ncol[1], nrow[ib] + ncol[2], nrow[ib] + ncol[3]};
      else acc[ib] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int g = 0; g < NG; ++g)
        acc[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(acol[g], brow[ib][g], acc[ib], 0, 0, 0);
    }
    // The two row blocks of a wave interleave (block ib holds rows 2 r16 + ib), so a lane's two results for a column are
    // NEIGHBOURS in memory: one 16-byte store per column instead of two 8-byte ones -- a store instruction writes 4 columns x
    // 256 contiguous bytes (r03: half the store instructions and address arithmetic of the build).
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#if GMB_KB_PROBE == 1 || GMB_KB_PROBE == 3  // probes: the store stream alone (no transcendental work); 3: neither
      const d2 v = d2{acc[0][r], acc[1][r]};
#else
      const d2 v = d2{stationary_interior<KIND>(acc[0][r], eta2), stationary_interior<KIND>(acc[1][r], eta2)};
#endif
      d2* dst = reinterpret_cast<d2*>(tile + (uint64_t)(4 * r) * col_bytes + lane_off);
#if GMB_KB_PROBE >= 2     // probe: the arithmetic alone (one store per lane and tile)
      probe_sum += v[0] + v[1];
      if (jb == ncb - 1 && r == 3) *dst = d2{probe_sum, probe_sum};
#else
      // streaming stores for matrices far larger than the caches (the factorisation that follows starts at the
      // other end of a 10 .. 80 GB buffer): +3 .. 7 % on the build at N = 50k / 100k
      if constexpr (NT) __builtin_nontemporal_store(v, dst);
      else *dst = v;
#endif
    }
    tile += 16 * col_bytes;
#pragma unroll
    for (int g = 0; g < NG; ++g) acol[g] = anext[g];
#pragma unroll
    for (int r = 0; r < 4; ++r) ncol[r] = nnext[r];
  }
}

// One tile through the direct loop: diagonal and boundary tiles, models with linear / coregion terms, later passes
// of additive models, interior tiles that hold a non-finite point (the differences carry NaN into the
// factorisation -> GMB_ENOTPD, and into predictions, exactly as the reference's arithmetic would), and every tile
// under -DGMB_KBUILD_DIRECT (the A/B switch for the matrix-pipe form).  Block-wide; ends with a barrier so that
// the caller may go on to the next tile of its strip.
template <int KIND, int NC>
__device__ __forceinline__ void cov_general_tile(const CovTileArgs& a, const int tix, const int tjx, const int part = 0, const int nparts = 1) {
  __shared__ double xj[NC][TILE];
  __shared__ double lj[MAX_LIN][TILE];
  __shared__ double li[MAX_LIN][TILE];
  __shared__ int32_t cj[MAX_TABS][TILE];
  __shared__ int32_t ci[MAX_TABS][TILE];
  const int64_t gi0 = a.i0 + (int64_t)tix * TILE;
  const int64_t gj0 = a.j0 + (int64_t)tjx * TILE;
  if (a.mode == COV_TRAIN && a.lower_only && gi0 + TILE - 1 < gj0) return;  // (block-uniform)
  const int tid = threadIdx.x;
  const CovParams& p = a.p;
  const bool full = gi0 + TILE <= a.rows.n && gj0 + TILE <= a.cols.n;
  const bool below = a.mode != COV_TRAIN || gj0 + TILE <= gi0;
  const int il = tid & (TILE - 1);
  const int jh = tid >> 7;  // which half of the tile's columns
  const int64_t gi = gi0 + il;

  // stage the tile's column points
  for (int idx = tid; idx < NC * TILE; idx += 256) {
    const int k = idx / TILE, j = idx - k * TILE;
    xj[k][j] = a.cols.xs[(int64_t)k * a.cols.npad + gj0 + j];
  }
  for (int idx = tid; idx < p.n_lin * TILE; idx += 256) {
    const int k = idx / TILE, j = idx - k * TILE;
    lj[k][j] = a.cols.xl[(int64_t)k * a.cols.npad + gj0 + j];
    li[k][j] = a.rows.xl[(int64_t)k * a.rows.npad + gi0 + j];
  }
  for (int idx = tid; idx < p.n_tab * TILE; idx += 256) {
    const int t = idx / TILE, j = idx - t * TILE;
    cj[t][j] = a.cols.cat[(int64_t)t * a.cols.npad + gj0 + j];
    ci[t][j] = a.rows.cat[(int64_t)t * a.rows.npad + gi0 + j];
  }
  // this thread's row point in registers
  double xi[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) xi[k] = a.rows.xs[(int64_t)k * a.rows.npad + gi];
  __syncthreads();

  const bool row_real = gi < a.rows.n;
  double* outp = a.out + cov_out_row(a, gi0) + il + (gj0 - a.j0 + jh * (TILE / 2)) * a.ldo;
  // this workgroup's share of the thread's 64 columns (gsplit)
  const int jj_lo = part * ((TILE / 2) / nparts), jj_hi = jj_lo + (TILE / 2) / nparts;
  double ndiag = 0.0;
  if (a.mode == COV_TRAIN && row_real) {
    ndiag = p.sigma2;
    if (p.noise_tab >= 0) ndiag *= p.noise_mult[ci[p.noise_tab][il]];
    ndiag += p.jitter;
  }

  if (full && below && p.n_lin == 0 && p.n_tab == 0) {  // no per-entry conditionals
    const double* xjp = &xj[0][jh * (TILE / 2)];
    auto entry = [&](int jj) {
      double r2 = 0.0;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        const double d = xi[k] - xjp[k * TILE + jj];
        r2 = fma(d, d, r2);
      }
      return p.eta2 * stationary<KIND>(r2);
    };
    if (a.accumulate) {  // (tested once per tile, not once per entry)
#pragma unroll 8
      for (int jj = jj_lo; jj < jj_hi; ++jj) outp[(int64_t)jj * a.ldo] += entry(jj);
    } else {
#pragma unroll 8
      for (int jj = jj_lo; jj < jj_hi; ++jj) outp[(int64_t)jj * a.ldo] = entry(jj);
    }
  } else {
    // U entries at a time: the distance / sqrt / exp chains of the entries are independent and interleave (one entry after
    // the other this loop was a single dependent chain per thread: ~35 us per diagonal / boundary tile whatever the size)
    constexpr int U = NC <= 4 ? 4 : 2;
    for (int jj0 = jj_lo; jj0 < jj_hi; jj0 += U) {
      double v_[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = jh * (TILE / 2) + jj0 + u;
        double r2 = 0.0;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          const double d = xi[k] - xj[k][j];
          r2 = fma(d, d, r2);
        }
        double v = p.eta2 * stationary<KIND>(r2);
        if (p.n_lin > 0) {
          double s = 0.0;
          for (int k = 0; k < p.n_lin; ++k) s = fma(li[k][il], lj[k][j], s);
          v = fma(p.tau, s, v);
        }
        for (int t = 0; t < p.n_tab; ++t)
          v *= p.tabs[p.tab_off[t] + ci[t][il] * p.tab_levels[t] + cj[t][j]];
        v_[u] = v;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int jj = jj0 + u;
        const int j = jh * (TILE / 2) + jj;
        const int64_t gj = gj0 + j;
        double v = v_[u];
        const bool col_real = gj < a.cols.n;
        if (a.accumulate) {
          if (row_real && col_real && (a.mode != COV_TRAIN || gi >= gj)) outp[(int64_t)jj * a.ldo] += v;
        } else if (a.mode == COV_TRAIN) {
          if (!col_real) {
            v = (gi == gj) ? 1.0 : 0.0;  // identity padding keeps the padded factor trivial
          } else if (!row_real) {
            v = (gi == a.rows.n) ? a.y[gj] : 0.0;  // appended y row: the factor's row n becomes L^-1 y
          } else if (gi == gj) {
            v += ndiag;
          }
          if (gi >= gj) outp[(int64_t)jj * a.ldo] = v;  // lower triangle only
        } else {
          outp[(int64_t)jj * a.ldo] = (row_real && col_real) ? v : 0.0;
        }
      }
    }
  }
  __syncthreads();  // the staging arrays are reused by the strip's next tile
}

// blocks of the rows 0 .. i-1 of a lower-triangular tile grid cut into strips of S tiles along the column index:
// row q (q + 1 tiles, at most tj) has floor(min(q, tj - 1) / S) + 1 strips
__host__ __device__ __forceinline__ long long cov_tri_blocks_before(int i, int tj, int S) {
  const int c = i < tj ? i : tj;
  const long long q = c / S, r = c - (c / S) * S;
  long long n = (long long)c + (long long)S * q * (q - 1) / 2 + q * r;  // sum_{q' < c} (floor(q' / S) + 1)
  if (i > tj) n += (long long)(i - tj) * ((tj - 1) / S + 1);
  return n;
}

// Block b of cov_tile_kernel's grid -> its strip: tile row *tix, tiles [*tj_lo, *tj_hi) along the column index;
// false for blocks with nothing to do.  Shared by the kernel and gmb_debug_cov_grid (tests/test_abi.py checks on
// the host that every tile of every enumeration comes out exactly once).  `sqrt_guess` differs between host and
// device only in rounding; the stepping loops make the result exact either way.
__host__ __device__ __forceinline__ bool cov_decode_block(int ti, int tj, int strip, int tri_grid, int row_first,
                                                          int row_stride, bool keep_order, long long b, int* tix,
                                                          int* tj_lo, int* tj_hi) {
  const int S = strip < 1 ? 1 : strip;
  int row, ts, ncols;  // tile row, strip index, tiles this row holds
  if (row_stride > 0) {
    const int nst = (tj + S - 1) / S;
    const int m = (int)(b / nst);
    ts = (int)(b - (long long)m * nst);
    row = row_first + m * row_stride;
    ncols = row + 1 < tj ? row + 1 : tj;
  } else if (tri_grid) {
    // blocks before row i ~ i + i^2 / (2 S): invert, then step to the exact row
    const double Sd = (double)S;
    int i = (int)(__builtin_sqrt(Sd * Sd + 2.0 * Sd * (double)b) - Sd);
    i = i < 0 ? 0 : (i > ti - 1 ? ti - 1 : i);
    while (cov_tri_blocks_before(i, tj, S) > b) --i;
    while (cov_tri_blocks_before(i + 1, tj, S) <= b) ++i;
    row = i;
    ts = (int)(b - cov_tri_blocks_before(i, tj, S));
    ncols = row + 1 < tj ? row + 1 : tj;
  } else {
    const int nst = (tj + S - 1) / S;
    const int wg = keep_order ? (int)b : xcd_remap((int)b, ti * nst);
    ts = wg / ti;
    row = wg - ts * ti;
    ncols = tj;
  }
  *tix = row;
  *tj_lo = ts * S;
  *tj_hi = *tj_lo + S < ncols ? *tj_lo + S : ncols;
  return row < ti && *tj_lo < *tj_hi;
}

// The grid.  A workgroup computes a STRIP of `strip` consecutive tiles of one tile row (same 128 row points,
// 128 * strip column points): its interior tiles in one pass of cov_interior_tile, whatever else it holds (the
// diagonal tile at the end of a row, boundary tiles) through cov_general_tile.  Strips amortise dispatch, decode,
// the finiteness check and the row operand's loads over more entries: worth +5 % at N = 100k (strip 4), nothing
// at N = 50k, and they cost balance on small grids -- launch_cov picks 1 / 2 / 4 by tile count.  Enumerations:
//   tri_grid   : rows 0 .. ti-1 of the lower triangle, row q with its floor(q / S) + 1 strips (closed-form decode);
//   row_stride : one rank's block rows row_first, row_first + row_stride, ...: ceil(tj / S) blocks per owned
//                row, those right of the diagonal exit at once;
//   otherwise  : rectangle ti x ceil(tj / S), dealt over the XCDs in equal runs (cross-covariance).
// Consecutive blocks go to consecutive XCDs (the hardware's own order), which balances the triangle.
// (__launch_bounds__(256, 2): with at most 256 registers per lane the compiler keeps the MFMA results in VGPRs;
// otherwise they land in AGPRs and every entry pays two v_accvgpr_read.)
template <int KIND, int NC>
__global__ __launch_bounds__(256, 2) void cov_tile_kernel(CovTileArgs a) {
  int tix, tj_lo, tj_hi;
  const int gs = a.gsplit > 1 ? a.gsplit : 1;  // (strip == 1 then: launch_cov)
  const int part = (int)(blockIdx.x % (unsigned)gs);
  // The triangle is walked from its END: the strips that hold direct-loop tiles -- a ragged last tile row, every row's diagonal
  // tile -- take ~30 us each whatever the matrix, and in ascending order the last of them STARTED when everything else was done.
  long long b = (long long)(blockIdx.x / (unsigned)gs);
  if (a.tri_grid) b = (long long)(gridDim.x / (unsigned)gs) - 1 - b;
  if (a.inline_prep && blockIdx.x == 0) {
#pragma unroll
    for (int z = 0; z < 2; ++z)
      if (a.prep.zero[z])
        for (int64_t w = threadIdx.x; w < a.prep.zero_words[z]; w += 256) a.prep.zero[z][w] = 0ull;
  }
  if (!cov_decode_block(a.ti, a.tj, a.strip, a.tri_grid, a.row_first, a.row_stride,
                        a.mode == COV_TRAIN && a.lower_only, b, &tix, &tj_lo, &tj_hi))
    return;
  if (a.inline_prep) {
    const int64_t gi0 = a.i0 + (int64_t)tix * TILE, gj0 = a.j0 + (int64_t)tj_lo * TILE;
    const int ncols = (tj_hi - tj_lo) * TILE;
    for (int q = threadIdx.x; q < TILE + ncols; q += 256) {
      const int64_t i = q < TILE ? gi0 + q : gj0 + (q - TILE);
      if (i < a.prep.npad) prep_one_point(a.prep, i);
    }
    __syncthreads();  // (workgroup-scope fence + barrier: the staging below reads what this workgroup has just written)
  }
  int tj_gen = tj_lo;  // first tile of the strip that goes through the direct loop
#ifndef GMB_KBUILD_DIRECT
  // Interior tiles: stationary term only, every row and column real and -- for the training matrix -- the tile
  // strictly below the diagonal; in a tile row they are the columns left of min(cols.n / 128, diagonal).
  // (Kinds with a kink at r = 0 -- Matern-1/2, Exponential -- keep the direct differences everywhere: an error e
  // of the expansion becomes e / (2 r) in exp(-r) near coinciding points, 5e-10 at r^2 + 1e-12 = 1e-12.)
  if constexpr (KIND <= 2) {
    const int64_t gi0 = a.i0 + (int64_t)tix * TILE;
    const CovParams& p = a.p;
    if (gi0 + TILE <= a.rows.n && p.n_lin == 0 && p.n_tab == 0 && !a.accumulate) {  // (additive models' later passes: direct loop)
      int64_t jend = (a.cols.n - a.j0) / TILE;                         // tiles whose 128 columns are all real
      if (a.mode == COV_TRAIN) jend = jend < (gi0 - a.j0) / TILE ? jend : (gi0 - a.j0) / TILE;  // ... and left of the diagonal
      const int tj_int = (int)(jend < tj_hi ? (jend > tj_lo ? jend : tj_lo) : tj_hi);
      if (tj_int > tj_lo) {
        const int tid = threadIdx.x;
        const int64_t gj0 = a.j0 + (int64_t)tj_lo * TILE + part * (TILE / gs);  // (gsplit: this workgroup's columns of the one tile)
        bool bad = tid < TILE && !(a.rows.xs[(int64_t)NC * a.rows.npad + gi0 + tid] < 1.0e150);
        for (int j = tid; j < (tj_int - tj_lo) * TILE / gs; j += 256)
          bad |= !(a.cols.xs[(int64_t)NC * a.cols.npad + gj0 + j] < 1.0e150);
        if (!__syncthreads_or(bad)) {  // all norms finite (and no coordinate absurdly far out)
          if (a.stream_stores) cov_interior_tile<KIND, NC, true>(a, gi0, gj0, 8 * (tj_int - tj_lo) / gs);
          else cov_interior_tile<KIND, NC, false>(a, gi0, gj0, 8 * (tj_int - tj_lo) / gs);
          tj_gen = tj_int;
        }
      }
    }
  }
#endif
  for (int tjx = tj_gen; tjx < tj_hi; ++tjx) cov_general_tile<KIND, NC>(a, tix, tjx, part, gs);
}

// ---------------------------------------------------------------------------------------------
// prior variance + noise at the test points:  kss[m] = diag K(x*_m, x*_m) (+ noise)
// ---------------------------------------------------------------------------------------------
struct KssArgs {
  CovParams p;
  PointSet pts;
  int32_t with_noise;
  double* kss;
  int32_t accumulate;  // additive models: later terms add to kss
};

__global__ void kss_kernel(KssArgs a) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= a.pts.npad) return;
  const CovParams& p = a.p;
  double v = p.eta2;  // stationary diag is exactly 1 (pm.gp.cov.Stationary.diag)
  if (p.n_lin > 0) {
    double s = 0.0;
    for (int k = 0; k < p.n_lin; ++k) {
      const double x = a.pts.xl[(int64_t)k * a.pts.npad + m];
      s = fma(x, x, s);
    }
    v = fma(p.tau, s, v);
  }
  for (int t = 0; t < p.n_tab; ++t) {
    const int c = a.pts.cat[(int64_t)t * a.pts.npad + m];
    v *= p.tabs[p.tab_off[t] + c * p.tab_levels[t] + c];
  }
  if (a.with_noise) {
    double nz = p.sigma2;
    if (p.noise_tab >= 0) nz *= p.noise_mult[a.pts.cat[(int64_t)p.noise_tab * a.pts.npad + m]];
    v += nz;
  }
  if (a.accumulate) {
    if (m < a.pts.n) a.kss[m] += v;
  } else {
    a.kss[m] = (m < a.pts.n) ? v : 0.0;
  }
}

// ---------------------------------------------------------------------------------------------
// posterior reductions over the solved cross-covariance V (m fast, ldz):
//   part_mu[c][m] = sum_{i in chunk c} V[m + i*ldz] * v[i],  part_s[c][m] = sum V^2
// then  mean = sum_c part_mu,  var = kss - sum_c part_s   (deterministic two-stage sum)
// ---------------------------------------------------------------------------------------------
constexpr int RED_CHUNK = 256;

__global__ __launch_bounds__(256) void predict_partial_kernel(const double* __restrict__ V,
                                                              int64_t ldz,
                                                              const double* __restrict__ v,
                                                              int64_t n, double* part_mu,
                                                              double* part_s, int64_t mpad) {
  __shared__ double vs[RED_CHUNK];
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.y * RED_CHUNK;
  const int64_t i1 = (i0 + RED_CHUNK < n) ? i0 + RED_CHUNK : n;
  if (i0 + threadIdx.x < n) vs[threadIdx.x] = v[i0 + threadIdx.x];
  __syncthreads();
  if (m >= mpad) return;  // mpad is a multiple of 128, the block covers 256
  double am = 0.0, as = 0.0;
  const double* col = V + m + i0 * ldz;
  for (int64_t i = i0; i < i1; ++i) {
    const double x = *col;
    am = fma(x, vs[i - i0], am);
    as = fma(x, x, as);
    col += ldz;
  }
  part_mu[(int64_t)blockIdx.y * mpad + m] = am;
  part_s[(int64_t)blockIdx.y * mpad + m] = as;
}

__global__ void predict_final_kernel(const double* part_mu, const double* part_s, int nchunk,
                                     int64_t mpad, const double* kss, int64_t M, double* mean,
                                     double* var) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  double am = 0.0, as = 0.0;
  for (int c = 0; c < nchunk; ++c) {
    am += part_mu[(int64_t)c * mpad + m];
    as += part_s[(int64_t)c * mpad + m];
  }
  mean[m] = am;
  var[m] = kss[m] - as;
}

// v[i] = L[n + i*ld] (the y row of the factor), and |v|^2 -- summed in ONE fixed order, the same in every kernel that forms it
// (this one behind a factorisation, eval_finish_kernel of eval_tiles.hpp inside a fused evaluation: same bits either way):
// chunk c = rows 128 c .. 128 c + 127 is summed by a fixed two-wave tree (v_chunk_sum); partial b = the chunk sums of the chunks
// c = b, b + 32, ... added in that order; |v|^2 = partials 0 .. 31 added in that order.  Workgroup b leaves its partial in
// scal[1 + b]; the workgroup that finishes LAST (a counter in scal[40], zeroed with the rest of the scalars at the start of
// every factorisation) adds them into scal[0].
// Launch with EXTRACT_V_BLOCKS workgroups of 128 threads; scal = the engine's scalar block + 1 ([0] |v|^2, [1..] partials).
constexpr int EXTRACT_V_BLOCKS = 32;
// every thread of a 128-thread workgroup calls it with its x^2; thread 0 returns the chunk's sum (ends with a barrier)
__device__ __forceinline__ double v_chunk_sum(double acc, double* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  const double s = red[0] + red[1];
  __syncthreads();
  return s;
}
__global__ __launch_bounds__(128) void extract_v_kernel(const double* L, int64_t ld, int64_t n,
                                                        double* v, double* scal) {
  __shared__ double red[2];
  __shared__ int last;
  double p = 0.0;
  const int64_t nchunks = (n + TILE - 1) / TILE;
  for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int64_t i = c * TILE + threadIdx.x;
    double x = 0.0;
    if (i < n) {
      x = L[n + i * ld];
      v[i] = x;
    }
    p += v_chunk_sum(x * x, red);
  }
  if (threadIdx.x == 0) {
    atomicExch((unsigned long long*)&scal[1 + blockIdx.x], (unsigned long long)__double_as_longlong(p));
    __threadfence();
    last = atomicAdd((unsigned int*)&scal[40], 1u) == gridDim.x - 1;
    if (last) {
      __threadfence();
      double s = 0.0;
      for (unsigned b = 0; b < gridDim.x; ++b)
        s += __longlong_as_double((long long)atomicAdd((unsigned long long*)&scal[1 + b], 0ull));  // read at the L2
      scal[0] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// pairwise-distance extrema for the lengthscale prior (gumbi/utils/gp_utils.py:34-46)
// ---------------------------------------------------------------------------------------------
// pts: SoA [ncols][npad]; group g covers columns [g*gw, (g+1)*gw).  Results are bit patterns of
// non-negative doubles, so unsigned-integer atomic min / max order them correctly.
__global__ __launch_bounds__(256) void ls_limits_kernel(const double* pts, int64_t n, int64_t npad,
                                                        int32_t gw, unsigned long long* mn,
                                                        unsigned long long* mx, int32_t tiles) {
  __shared__ double xj[16][TILE];
  const int g = blockIdx.y;
  // decode lower-triangle tile pair (ti >= tj) from blockIdx.x
  const double t2 = 2.0 * (double)tiles + 1.0;
  int tjx = (int)(0.5 * (t2 - __builtin_sqrt(t2 * t2 - 8.0 * (double)blockIdx.x)));
  auto before = [&](int q) -> long long { return (long long)q * tiles - (long long)q * (q - 1) / 2; };
  tjx = tjx < 0 ? 0 : (tjx > tiles ? tiles : tjx);
  while (before(tjx) > (long long)blockIdx.x) --tjx;
  while (before(tjx + 1) <= (long long)blockIdx.x) ++tjx;
  const int tix = tjx + (int)((long long)blockIdx.x - before(tjx));
  const int tid = threadIdx.x;
  const int il = tid & (TILE - 1), jh = tid >> 7;
  const int64_t gi = (int64_t)tix * TILE + il, gj0 = (int64_t)tjx * TILE;
  for (int idx = tid; idx < gw * TILE; idx += 256) {
    const int k = idx / TILE, j = idx - k * TILE;
    xj[k][j] = pts[(int64_t)(g * gw + k) * npad + gj0 + j];
  }
  __syncthreads();
  double lo = 1.0e300, hi = 0.0;
  if (gi < n) {
    for (int jj = 0; jj < TILE / 2; ++jj) {
      const int j = jh * (TILE / 2) + jj;
      const int64_t gj = gj0 + j;
      if (gj >= gi) continue;  // strictly lower pairs, each once
      double r2 = 0.0;
      for (int k = 0; k < gw; ++k) {
        const double d = pts[(int64_t)(g * gw + k) * npad + gi] - xj[k][j];
        r2 = fma(d, d, r2);
      }
      if (r2 > 0.0) {
        lo = fmin(lo, r2);
        hi = fmax(hi, r2);
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fmin(lo, __shfl_down(lo, off));
    hi = fmax(hi, __shfl_down(hi, off));
  }
  if ((tid & 63) == 0) {
    if (lo < 1.0e300) atomicMin(&mn[g], (unsigned long long)__double_as_longlong(lo));
    if (hi > 0.0) atomicMax(&mx[g], (unsigned long long)__double_as_longlong(hi));
  }
}

}  // namespace gmb
