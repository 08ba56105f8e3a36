// gradient.hpp -- fused trace reductions for the NLML gradient
//
//   d NLML / d theta = sum_ij M_ij * dSigma_ij/dtheta,   M = 1/2 (Sigma^-1 - alpha alpha^T)
//
// PyMC gets this by reverse-mode autodiff through its Cholesky op inside pm.find_MAP (call site
// gumbi/regression/pymc/GP.py:811); here Sigma^-1 (lower triangle, column-major, from the MFMA
// GEMMs in engine.hip) is streamed ONCE from HBM and every per-parameter partial is reduced in
// the same pass -- distances and kernel values are recomputed from the LDS-staged coordinates
// instead of being re-read.  Off-diagonal entries are visited once (i > j) with weight 2.
//
// Accumulator layout (doubles) in `acc`:
//   [0, NC)            d/d ls_k   (ARD)  or [0] d/d ls (shared)      -- natural-scale lengthscales
//   [NC]               d/d eta
//   [NC+1]             d/d tau
//   [NC+2, +n_lin)     d/d c_k
//   then per coregion table t: L_t*L_t entries G_t[a][b] = sum M_ij * K_ij / B_t[a,b]
// The diagonal-only terms (sigma, noise table) come from grad_diag_kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "covariance.hpp"

namespace gmb {

struct GradArgs {
  CovParams p;
  PointSet pts;
  const double* Z;  // Sigma^-1, lower triangle valid, column-major
  int64_t ldz;
  const double* alpha;
  int32_t tiles;    // tiles per side (N rounded up / 128)
  int32_t ard;
  int32_t nc_real;
  double inv_ls[16];
  double eta;
  // Deterministic two-stage reduction (no floating-point atomics whose order depends on scheduling): workgroup
  // b reduces the tiles [b * per, (b + 1) * per) of the launch's enumeration and WRITES its sums densely to
  // part[b * part_stride + q]; grad_sum_partials_kernel adds the workgroups' vectors in a fixed order.
  //   q in [0, NC + 2 + n_lin): ls.. | eta | tau | c..      then the tables with <= 8 levels, 64 slots each
  double* part;
  int32_t part_stride;
  int64_t total_tiles;  // tiles of this launch
  int32_t per;          // tiles per workgroup
  // tables with more than 8 levels do not fit the LDS copies: every WAVE adds into a private global copy
  // big[(b * 4 + wave) * big_stride + big_off[t] + ca * L + cb] (zeroed by the caller, summed afterwards)
  double* big;
  int32_t big_stride;
  int32_t big_off[MAX_TABS];
  // shard of the lower triangle this launch reduces: block rows row_first, row_first + row_stride, ...
  // (0, 1 = everything; a rank of the multi-GPU gradient passes (rank, world))
  int32_t row_first, row_stride;
  // z_packed: Z holds ONLY the owned block rows, packed (row block t of Z = block row row_first + t*row_stride)
  int32_t z_packed;
  // split: the interior tiles (full tiles strictly below the diagonal) are grad_interior_kernel's, the rest grad_tile_kernel's;
  // part_block0: first partial vector of this launch (the two launches of a split pass write grid vectors each)
  int32_t split, part_block0;
  int32_t n_owned_rows;  // split, MODE 0: block rows of this shard (its diagonal tiles lead the general-tile list)
  int32_t general_tiles; // ... and the length of that list = grid of the MODE 0 launch
  // gsplit (grad_tile_kernel; small launches): every run / list tile of the direct loops is shared by `gsplit` workgroups, each
  // with its share of the thread's 64 columns (a direct-loop tile is a serial chain of 16 entry groups per thread, ~27 us
  // whatever the matrix).  Workgroup b = run (b / gsplit), share (b % gsplit); one partial vector per workgroup.  1 = off.
  int32_t gsplit;
};

constexpr int GRAD_SMALL = 16 + 2 + MAX_LIN;                 // ls.. | eta | tau | c..
constexpr int GRAD_MAX_LDS_ACC = GRAD_SMALL + MAX_TABS * 64;  // + tables up to 8 levels
// dynamic LDS of grad_tile_kernel<KIND, NC> for a model with n_lin linear dims and n_tab coregion tables
inline size_t grad_lds_bytes(int nc, int n_lin, int n_tab) {
  return sizeof(double) * ((size_t)nc * TILE + TILE + 4 * GRAD_SMALL + 2 * (size_t)n_lin * TILE + 4 * (size_t)n_tab * 64) +
         sizeof(int32_t) * 2 * (size_t)n_tab * TILE;
}

// ---- interior tiles: the per-dimension sums on the matrix pipe --------------------------------------------------
// For a full 128 x 128 tile strictly below the diagonal of a stationary model every entry stands for (i,j) and (j,i):
//     g_eta   += mm_ij * 2 eta k_ij,                       mm = Sigma^-1_ij - alpha_i alpha_j  (= 2 M_ij)
//     g_ls[k] += -2/ls_k * mm_ij eta^2 k'_ij * ((x_ik - x_jk)/ls_k)^2
// The round-3 loop spent ~83 double-precision instructions per entry on this (differences, squares and two multiply-adds per
// dimension, at 233 registers and two waves per SIMD): 6.9 ms for the 10 GB of C3's Sigma^-1, 0.18 of the HBM roofline.
// Here, as in cov_interior_tile, r^2 comes off the matrix pipe (PyMC's own expansion |x|^2 + |x'|^2 - 2 x.x'), ONE value
//     G_ij = mm_ij eta^2 k'(r^2_ij)
// is formed per entry, and the per-dimension sums are taken apart as
//     sum_ij G_ij (x_ik - x_jk)^2 = sum_i [ x_ik^2 R_i - 2 x_ik P_ik + S_ik ],
//     R_i = sum_j G_ij,   P_ik = sum_j G_ij x_jk,   S_ik = sum_j G_ij x_jk^2:
// P and S are a skinny product G . [X | X^2] -- the G values sit in the accumulator layout of the r^2 contraction, which IS
// the B-operand layout of v_mfma_f64_16x16x4_f64 with the column index as contraction index, so four MFMAs per 16 x 16
// entries (one per accumulator register, A operand = the 2 NC features of four column points) add them into an accumulator
// that belongs to the wave's 32 ROWS and stays in registers along the whole tile row; R is one add per entry.  The rows'
// own coordinates enter once per tile row (grad_rows_flush).  ~60 issue-slot equivalents per entry for Matern-5/2, d = 8.
// The expansion carries ~eps |x/ls|^2 of absolute rounding per term (as the covariance build's interior tiles do).
template <int KIND>
__device__ __forceinline__ void stationary_pair_interior(const double q, const double eta2, double& ks, double& dk) {
  if constexpr (KIND == 0) {  // q = -r^2 / 2
    const double e = eta2 * exp_interior(fmin(q, 0.0));
    ks = e;
    dk = -0.5 * e;
  } else {
    const double p = fmax(q, 1e-12);  // q = r^2 + 1e-12 (see stationary_interior)
    const double r = sqrt_interior(p);
    if constexpr (KIND == 1) {
      const double s5 = 2.23606797749978969641;
      const double e = exp_interior(-s5 * r);
      const double lin = fma(eta2 * s5, r, eta2);  // eta^2 (1 + sqrt5 r)
      ks = fma(eta2 * (5.0 / 3.0), p, lin) * e;
      dk = (-5.0 / 6.0) * lin * e;
    } else {
      const double s3 = 1.73205080756887729353;
      const double e = eta2 * exp_interior(-s3 * r);
      ks = fma(s3, r, 1.0) * e;
      dk = -1.5 * e;
    }
  }
}

template <int NC>
struct GradRowAcc {  // what a wave has summed for its 32 rows since the last flush
  static constexpr int ND = (2 * NC + 15) / 16;  // accumulators of 16 features x 16 rows
  d4 D[2][ND];
  double R[2];
};

template <int NC>
__device__ __forceinline__ void grad_rows_reset(GradRowAcc<NC>& s) {
#pragma unroll
  for (int ib = 0; ib < 2; ++ib) {
    s.R[ib] = 0.0;
#pragma unroll
    for (int n = 0; n < GradRowAcc<NC>::ND; ++n) s.D[ib][n] = d4{0.0, 0.0, 0.0, 0.0};
  }
}

// the rows' side of the sums: g[k] += x_ik^2 R_i - 2 x_ik P_ik + S_ik for this lane's pieces (feature kq + 4 r of row slot r16)
template <int NC>
__device__ __forceinline__ void grad_rows_flush(const GradArgs& a, const int64_t gi0, GradRowAcc<NC>& s, double (&g)[NC]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int ib = 0; ib < 2; ++ib) {
    const double* src = a.pts.xs + gi0 + 32 * wave + 2 * r16 + ib;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const double x = src[(int64_t)k * a.pts.npad];
      double add = (x * x) * s.R[ib];  // (this lane's share of R_i: its column groups)
#pragma unroll
      for (int n = 0; n < GradRowAcc<NC>::ND; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 16 * n + kq + 4 * r;  // feature index: f < NC -> P_ik (k = f), NC <= f < 2 NC -> S_ik (k = f - NC)
          add += (f == k) ? -2.0 * x * s.D[ib][n][r] : (f == NC + k) ? s.D[ib][n][r] : 0.0;
        }
      g[k] += add;
    }
  }
  grad_rows_reset(s);
}

// A run of interior tiles of ONE tile row: rows gi0 = 128 tix .. (this wave: 32 of them), column tiles tj0 .. tj1-1 (all left of
// the diagonal, all rows and columns real).  What the columns contribute -- coordinates, their squares, norms, alpha: 10 KB per
// tile -- is staged in LDS once per tile by the whole workgroup (requested while the previous tile computes) in the two layouts
// the two contractions read: F[k][col] (k-major, pitch 144: the A operand of the r^2 contraction) and FT[col][feature]
// (pitch 16 ND + 1: the A operand of G . [X | X^2]); as per-lane global loads these 14 gathers per 16-column block were half of
// the kernel (probe: 2.1 of 3.7 ms at C3 with neither the Sigma^-1 stream nor the transcendental work).  Sigma^-1 itself streams
// from HBM straight into registers, three 16-column blocks ahead and continuously across the tiles of the run.
#ifndef GMB_GRI_WAVES
#define GMB_GRI_WAVES 2  // waves per SIMD grad_interior_kernel is compiled for
#endif
template <int NC>
struct GradInteriorLds {
  static constexpr int ND = GradRowAcc<NC>::ND;
  static constexpr int FP = TILE + 16;    // pitch of F: 144 doubles == 32 banks mod 64 (the two k rows of a half-wave on disjoint banks)
  static constexpr int TP = 16 * ND + 1;  // pitch of FT: odd, so that the staging writes of 32 consecutive columns hit 32 different banks
  static constexpr int DOUBLES = NC * FP + TILE * TP + 2 * TILE;
};

template <int KIND, int NC>
__device__ __forceinline__ void grad_interior_segment(const GradArgs& a, const int tix, const int tj0, const int tj1, double* lds,
                                                      double (&g_ls2)[NC], double& g_eta2) {
  using L = GradInteriorLds<NC>;
  constexpr bool NORM_IN_C = (NC + 2 + 3) / 4 > (NC + 3) / 4;
  constexpr int KA = NORM_IN_C ? NC : NC + 2;
  constexpr int NG = (KA + 3) / 4;
  constexpr int ND = L::ND, FP = L::FP, TP = L::TP;
  constexpr double sc = KIND == 0 ? 1.0 : -2.0;
  constexpr double sn = KIND == 0 ? -0.5 : 1.0;
  constexpr double off12 = KIND == 0 ? 0.0 : 1e-12;
  double* const F = lds;
  double* const FT = F + NC * FP;
  d2* const NA = reinterpret_cast<d2*>(FT + TILE * TP);  // {sn |x_j|^2, alpha_j}
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int64_t npad = a.pts.npad;
  const int64_t gi0 = (int64_t)tix * TILE;
  // row side (slot r16 of block ib <-> row 2 r16 + ib: the two rows of a lane are neighbours in memory)
  double brow[2][NG], nrow[2], ai[2];
#pragma unroll
  for (int ib = 0; ib < 2; ++ib) {
    const double* src = a.pts.xs + gi0 + 32 * wave + 2 * r16 + ib;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int k = 4 * g + kq;
      double v = k < (NORM_IN_C ? NC : NC + 1) ? src[(int64_t)k * npad] : 0.0;
      if (!NORM_IN_C && k == NC) v = fma(v, sn, off12);
      if (!NORM_IN_C && k == NC + 1) v = 1.0;
      brow[ib][g] = v;
    }
    nrow[ib] = NORM_IN_C ? fma(sn, src[(int64_t)NC * npad], off12) : 0.0;
    ai[ib] = a.alpha[gi0 + 32 * wave + 2 * r16 + ib];
  }
  GradRowAcc<NC> s;
  grad_rows_reset(s);
  const int64_t zrow0 = a.z_packed ? (int64_t)((tix - a.row_first) / a.row_stride) * TILE : gi0;
  const double* zsrc = a.Z + zrow0 + 32 * wave + 2 * r16 + ((int64_t)tj0 * TILE + kq) * a.ldz;
  struct ZBlk {
    d2 z[4];
  };
  constexpr int NB = TILE / 16;
  const int nblk = NB * (tj1 - tj0);
  auto load_z = [&](int B, ZBlk& zb) {
    B = B < nblk ? B : nblk - 1;
#if defined(GMB_GR_PROBE) && (GMB_GR_PROBE & 1)  // measurement probe: no Sigma^-1 stream (arithmetic alone)
#pragma unroll
    for (int r = 0; r < 4; ++r) zb.z[r] = d2{1.0 + B, 2.0};
#else
#pragma unroll
    for (int r = 0; r < 4; ++r) zb.z[r] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(zsrc + (int64_t)(16 * B + 4 * r) * a.ldz));
#endif
  };
  // staging: thread (column j = tid & 127, half h = tid >> 7) carries the coordinates k = h KH .. of its column, and the
  // norm (h = 0) or alpha (h = 1)
  constexpr int KH = (NC + 1) / 2;
  const int sj = tid & (TILE - 1), sh = tid >> 7;
  double sx[KH], sna;
  auto stage_load = [&](const int t) {
    const int64_t gj = (int64_t)t * TILE + sj;
#pragma unroll
    for (int q = 0; q < KH; ++q) {
      const int k = sh * KH + q;
      sx[q] = k < NC ? a.pts.xs[(int64_t)k * npad + gj] : 0.0;
    }
    sna = sh == 0 ? sn * a.pts.xs[(int64_t)NC * npad + gj] : a.alpha[gj];
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int q = 0; q < KH; ++q) {
      const int k = sh * KH + q;
      if (k < NC) {
        F[k * FP + sj] = sx[q];
        FT[sj * TP + k] = sx[q];
        FT[sj * TP + NC + k] = sx[q] * sx[q];
      }
    }
    reinterpret_cast<double*>(NA)[2 * sj + sh] = sna;
  };
  __syncthreads();  // (whoever used the staging area before is done with it)
  if (2 * NC < 16 * ND)  // features beyond 2 NC: zero, once
    for (int idx = tid; idx < TILE * (16 * ND - 2 * NC); idx += 256) {
      const int j = idx / (16 * ND - 2 * NC), f = 2 * NC + idx - j * (16 * ND - 2 * NC);
      FT[j * TP + f] = 0.0;
    }
  stage_load(tj0);
  stage_store();
  ZBlk z0, z1, z2, z3;
  load_z(0, z0);
  load_z(1, z1);
  load_z(2, z2);
  __syncthreads();
  const double eta2 = a.p.eta2;
  for (int t = tj0; t < tj1; ++t) {
    if (t + 1 < tj1) stage_load(t + 1);
#pragma unroll 1
    for (int jb = 0; jb < NB; ++jb) {
      load_z(NB * (t - tj0) + jb + 3, z3);
      // column side of this 16-column block from LDS
      double acol[NG], feat[ND][4];
      d2 na[4];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int k = 4 * g + kq;
        double v;
        if constexpr (NORM_IN_C) {
          v = k < NC ? sc * F[k * FP + 16 * jb + r16] : 0.0;
        } else {
          v = k < NC ? sc * F[(k < NC ? k : 0) * FP + 16 * jb + r16] : (k == NC ? 1.0 : (k == NC + 1 ? NA[16 * jb + r16][0] : 0.0));
        }
        acol[g] = v;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        na[r] = NA[16 * jb + kq + 4 * r];
#pragma unroll
        for (int n = 0; n < ND; ++n) feat[n][r] = FT[(16 * jb + kq + 4 * r) * TP + 16 * n + r16];
      }
      d4 acc[2];
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) {
        if constexpr (NORM_IN_C) acc[ib] = d4{nrow[ib] + na[0][0], nrow[ib] + na[1][0], nrow[ib] + na[2][0], nrow[ib] + na[3][0]};
        else acc[ib] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(acol[g], brow[ib][g], acc[ib], 0, 0, 0);
      }
      double G[2][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ib = 0; ib < 2; ++ib) {
          double ks, dk;
#if defined(GMB_GR_PROBE) && (GMB_GR_PROBE & 2)  // measurement probe: no transcendental work (the stream + the MFMAs alone)
          ks = acc[ib][r];
          dk = -0.5 * ks;
#else
          stationary_pair_interior<KIND>(acc[ib][r], eta2, ks, dk);
#endif
          const double mm = fma(-ai[ib], na[r][1], z0.z[r][ib]);
          g_eta2 = fma(mm, ks, g_eta2);
          G[ib][r] = mm * dk;
          s.R[ib] += G[ib][r];
        }
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int n = 0; n < ND; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) s.D[ib][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(feat[n][r], G[ib][r], s.D[ib][n], 0, 0, 0);
      z0 = z1;
      z1 = z2;
      z2 = z3;
    }
    __syncthreads();  // every wave is done with this tile's columns
    if (t + 1 < tj1) {
      stage_store();
      __syncthreads();
    }
  }
  grad_rows_flush<NC>(a, gi0, s, g_ls2);
}

// MODE 0: every tile the interior kernel does not take (all of them when a.split == 0); MODE 1: the interior tiles only.  Two
// kernels over the same enumeration and the same partition into workgroup runs, so that the matrix-pipe path keeps its own
// register budget (together the two paths asked for 299 registers: one wave per SIMD); each leaves its own partial vectors
// (a.part_block0 + blockIdx.x), summed in a fixed order by grad_sum_partials_kernel.
template <int KIND, int NC, int MODE>
__device__ __forceinline__ void grad_tile_body(const GradArgs& a) {
  // LDS sized by the model (grad_lds_bytes): the plain stationary case keeps 8 workgroups per compute unit
  // (10 KB each at d = 8); static arrays for the largest model cost 41 KB and two thirds of the occupancy
  extern __shared__ double grad_dsm[];
  const CovParams& p = a.p;
  double* const xj = grad_dsm;                             // [NC][TILE]
  double* const aj = xj + NC * TILE;                       // [TILE]
  double* const swave = aj + TILE;                         // [4][GRAD_SMALL]  per-wave sums of the register accumulators
  double* const lj = swave + 4 * GRAD_SMALL;               // [n_lin][TILE]
  double* const li = lj + p.n_lin * TILE;                  // [n_lin][TILE]
  double* const stab = li + p.n_lin * TILE;                // [4][n_tab * 64]  per-wave copies of the small coregion tables
  int32_t* const cj = reinterpret_cast<int32_t*>(stab + 4 * p.n_tab * 64);  // [n_tab][TILE]
  int32_t* const ci = cj + p.n_tab * TILE;                 // [n_tab][TILE]

  const int tid = threadIdx.x, wave = tid >> 6;
  const int il = tid & (TILE - 1), jh = tid >> 7;
  const int n_acc_small = NC + 2 + p.n_lin;
  const int tabw = p.n_tab * 64;
  for (int idx = tid; idx < 4 * tabw; idx += 256) stab[idx] = 0.0;

  double g_ls[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) g_ls[k] = 0.0;
  double g_eta = 0.0, g_tau = 0.0;
  double g_c[MAX_LIN];
#pragma unroll
  for (int k = 0; k < MAX_LIN; ++k) g_c[k] = 0.0;

  // this workgroup's run of lower-triangle tile pairs (ti >= tj), enumerated block row by owned block row
  // (closed form: owned row m = (tix - row_first) / row_stride is preceded by m (row_first + 1) + row_stride m (m - 1) / 2 tiles)
  const int gs = a.gsplit > 1 ? a.gsplit : 1;
  const int bidx = (int)(blockIdx.x / (unsigned)gs), share = (int)(blockIdx.x % (unsigned)gs);
  const int jj_lo = share * ((TILE / 2) / gs), jj_hi = jj_lo + (TILE / 2) / gs;
  long long t_begin = (long long)bidx * a.per;
  long long t_end = t_begin + a.per;
  t_end = t_end < a.total_tiles ? t_end : a.total_tiles;
  // MODE 0 beside an interior launch: ONE tile per workgroup from the short list of the tiles that launch leaves -- the
  // diagonal tile of every owned block row, then (a ragged last block row that is owned) its tiles left of the diagonal.
  // (As runs of the full enumeration the ~390 tiles of C3's last row fell to eight workgroups: 2.7 ms for 0.5 % of the work.)
  int list_tix = -1, list_tjx = -1;
  if (MODE == 0 && a.split) {
    const int b = bidx;
    if (b < a.n_owned_rows) {
      list_tix = list_tjx = a.row_first + b * a.row_stride;
    } else {
      list_tix = a.tiles - 1;
      list_tjx = b - a.n_owned_rows;
    }
    t_begin = 0;
    t_end = 1;
  }
  auto before = [&](int q) -> long long { return (long long)q * (a.row_first + 1) + (long long)a.row_stride * q * (q - 1) / 2; };
  int m = 0;
  if (t_begin < t_end) {
    const double sd = (double)a.row_stride, f1 = (double)a.row_first + 1.0 - 0.5 * sd;
    m = (int)((__builtin_sqrt(f1 * f1 + 2.0 * sd * (double)t_begin) - f1) / sd);
    m = m < 0 ? 0 : m;
    while (before(m) > t_begin) --m;
    while (before(m + 1) <= t_begin) ++m;
  }
  int tix = a.row_first + m * a.row_stride;
  int tjx = (int)(t_begin - before(m));
  if (MODE == 0 && a.split) {
    tix = list_tix;
    tjx = list_tjx;
  }

  for (long long tile = t_begin; tile < t_end; ++tile) {
    const int64_t gi0 = (int64_t)tix * TILE, gj0 = (int64_t)tjx * TILE;
    const int64_t gi = gi0 + il;
    // interior tile of a smooth stationary model (block-uniform test): the matrix-pipe form, nothing staged in LDS
    const bool interior = a.split && tix > tjx && gi0 + TILE <= a.pts.n;
    (void)interior;  // (MODE 0 beside an interior launch walks the general-tile list: none of its tiles is interior)
    __syncthreads();  // the previous tile's readers are done with the staged coordinates
    for (int idx = tid; idx < NC * TILE; idx += 256) {
      const int k = idx / TILE, j = idx - k * TILE;
      xj[k * TILE + j] = a.pts.xs[(int64_t)k * a.pts.npad + gj0 + j];
    }
    for (int idx = tid; idx < p.n_lin * TILE; idx += 256) {
      const int k = idx / TILE, j = idx - k * TILE;
      lj[k * TILE + j] = a.pts.xl[(int64_t)k * a.pts.npad + gj0 + j];
      li[k * TILE + j] = a.pts.xl[(int64_t)k * a.pts.npad + gi0 + j];
    }
    for (int idx = tid; idx < p.n_tab * TILE; idx += 256) {
      const int t = idx / TILE, j = idx - t * TILE;
      cj[t * TILE + j] = a.pts.cat[(int64_t)t * a.pts.npad + gj0 + j];
      ci[t * TILE + j] = a.pts.cat[(int64_t)t * a.pts.npad + gi0 + j];
    }
    if (tid < TILE) aj[tid] = (gj0 + tid < a.pts.n) ? a.alpha[gj0 + tid] : 0.0;
    double xi[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) xi[k] = a.pts.xs[(int64_t)k * a.pts.npad + gi];
    const bool row_real = gi < a.pts.n;
    const double ai = row_real ? a.alpha[gi] : 0.0;
    __syncthreads();

    const int64_t zrow = a.z_packed ? (int64_t)((tix - a.row_first) / a.row_stride) * TILE + il : gi;
    const double* zp = a.Z + zrow + (gj0 + jh * (TILE / 2)) * a.ldz;
    // Fast path (almost every tile): stationary term only, tile strictly below the diagonal, every
    // row and column real -- each entry stands for (i,j) and (j,i), no per-entry conditionals.
    const bool fast = p.n_lin == 0 && p.n_tab == 0 && tix > tjx && gi0 + TILE <= a.pts.n && gj0 + TILE <= a.pts.n;
    if (fast) {
      const int jb = jh * (TILE / 2);
#pragma unroll 4
      for (int jj = jj_lo; jj < jj_hi; ++jj) {
        const double mm = zp[(int64_t)jj * a.ldz] - ai * aj[jb + jj];  // 2 * M_ij
        double d2[NC];
        double r2 = 0.0;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          const double d = xi[k] - xj[k * TILE + jb + jj];
          d2[k] = d * d;
          r2 += d2[k];
        }
        const double ks = stationary<KIND>(r2);
        const double mdk = mm * (p.eta2 * stationary_dr2<KIND>(r2));
#pragma unroll
        for (int k = 0; k < NC; ++k) g_ls[k] = fma(mdk, -2.0 * d2[k] * a.inv_ls[k], g_ls[k]);
        g_eta = fma(mm, 2.0 * a.eta * ks, g_eta);
      }
    }
    // U entries at a time: their distance / exp / sqrt chains are independent and interleave (one entry after the other the
    // loop was a single dependent chain per thread: 32 - 39 us per tile whatever the size); entries that do not count (padding,
    // above the diagonal) carry M = 0 instead of being skipped.  The sums are still added in jj order: same bits as before.
    constexpr int U = NC <= 4 ? 4 : 2;
    // (the Sigma^-1 values of the NEXT group are requested before this group is computed: one exposed memory latency per
    // group of entries was most of a small tile's time)
    double znext[U];
    auto load_z_group = [&](const int jj0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t gj = gj0 + jh * (TILE / 2) + jj0 + u;
        znext[u] = (jj0 < jj_hi && row_real && gj < a.pts.n && gj <= gi) ? zp[(int64_t)(jj0 + u) * a.ldz] : 0.0;
      }
    };
    load_z_group(fast ? jj_hi : jj_lo);
    for (int jj0 = fast ? jj_hi : jj_lo; jj0 < jj_hi; jj0 += U) {
      double mfull_[U], ks_[U], dk_[U], lin_[U], F_[U], d2_[U][NC], zcur[U];
      bool valid_[U];
#pragma unroll
      for (int u = 0; u < U; ++u) zcur[u] = znext[u];
      load_z_group(jj0 + U);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int jj = jj0 + u;
        const int j = jh * (TILE / 2) + jj;
        const int64_t gj = gj0 + j;
        valid_[u] = row_real && gj < a.pts.n && gj <= gi;
        const double z = zcur[u];
        mfull_[u] = valid_[u] ? 0.5 * (z - ai * aj[j]) : 0.0;  // M_ij
        double r2 = 0.0;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          const double d = xi[k] - xj[k * TILE + j];
          d2_[u][k] = d * d;
          r2 += d2_[u][k];
        }
        ks_[u] = stationary<KIND>(r2);
        dk_[u] = p.eta2 * stationary_dr2<KIND>(r2);
        double lin = 0.0;
        for (int k = 0; k < p.n_lin; ++k) lin = fma(li[k * TILE + il], lj[k * TILE + j], lin);
        lin_[u] = lin;
        double F = 1.0;
        for (int t = 0; t < p.n_tab; ++t)
          F *= p.tabs[p.tab_off[t] + ci[t * TILE + il] * p.tab_levels[t] + cj[t * TILE + j]];
        F_[u] = F;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int j = jh * (TILE / 2) + jj0 + u;
        const int64_t gj = gj0 + j;
        const double mfull = mfull_[u], ks = ks_[u], dk = dk_[u], lin = lin_[u];
        const double mm = (gi == gj) ? mfull : 2.0 * mfull;  // (i,j) and (j,i)
        const double mF = mm * F_[u];
        // d r2 / d ls_k = -2 d2_k / ls_k   (d2 already in scaled units)
#pragma unroll
        for (int k = 0; k < NC; ++k) g_ls[k] = fma(mF * dk, -2.0 * d2_[u][k] * a.inv_ls[k], g_ls[k]);
        g_eta = fma(mF, 2.0 * a.eta * ks, g_eta);
        if (p.n_lin > 0) {
          g_tau = fma(mF, lin, g_tau);
#pragma unroll
          for (int k = 0; k < MAX_LIN; ++k)
            if (k < p.n_lin) g_c[k] = fma(-mF * p.tau, li[k * TILE + il] + lj[k * TILE + j], g_c[k]);
        }
        if (p.n_tab > 0 && valid_[u]) {
          const double base = p.eta2 * ks + p.tau * lin;
          for (int t = 0; t < p.n_tab; ++t) {
            double others = 1.0;
            for (int t2 = 0; t2 < p.n_tab; ++t2)
              if (t2 != t) others *= p.tabs[p.tab_off[t2] + ci[t2 * TILE + il] * p.tab_levels[t2] + cj[t2 * TILE + j]];
            const int L = p.tab_levels[t];
            const double val = mfull * base * others;
            // ordered pair (i,j) feeds G[a][b]; its mirror (j,i) feeds G[b][a] (off-diagonal only).  The copies
            // are private to this wave: lanes of one instruction that hit the same entry are served in the
            // hardware's fixed lane order, and no other wave ever adds into them.
            const int ca = ci[t * TILE + il], cb = cj[t * TILE + j];
            const bool off = gi != gj;
            if (L <= 8) {
              atomicAdd(&stab[wave * tabw + t * 64 + ca * L + cb], val);
              if (off) atomicAdd(&stab[wave * tabw + t * 64 + cb * L + ca], val);
            } else {
              double* bt = a.big + ((int64_t)blockIdx.x * 4 + wave) * a.big_stride + a.big_off[t];
              atomicAdd(&bt[ca * L + cb], val);
              if (off) atomicAdd(&bt[cb * L + ca], val);
            }
          }
        }
      }
    }
    // next tile of the enumeration
    if (++tjx > tix) {
      tix += a.row_stride;
      tjx = 0;
    }
  }

  // wave sums of the register accumulators (fixed shuffle tree), then the four waves in wave order
  auto wave_sum = [&](double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
  };
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const double v = wave_sum(g_ls[k]);
    if ((tid & 63) == 0) swave[wave * GRAD_SMALL + k] = v;
  }
  {
    const double v = wave_sum(g_eta), u = wave_sum(g_tau);
    if ((tid & 63) == 0) {
      swave[wave * GRAD_SMALL + NC] = v;
      swave[wave * GRAD_SMALL + NC + 1] = u;
    }
  }
#pragma unroll
  for (int k = 0; k < MAX_LIN; ++k) {
    if (k < p.n_lin) {
      const double v = wave_sum(g_c[k]);
      if ((tid & 63) == 0) swave[wave * GRAD_SMALL + NC + 2 + k] = v;
    }
  }
  __syncthreads();
  double* out = a.part + ((int64_t)a.part_block0 + blockIdx.x) * a.part_stride;
  if (tid < n_acc_small) out[tid] = (swave[tid] + swave[GRAD_SMALL + tid]) + (swave[2 * GRAD_SMALL + tid] + swave[3 * GRAD_SMALL + tid]);
  for (int idx = tid; idx < tabw; idx += 256)
    out[n_acc_small + idx] = (stab[idx] + stab[tabw + idx]) + (stab[2 * tabw + idx] + stab[3 * tabw + idx]);
}

template <int KIND, int NC>
__global__ __launch_bounds__(256) void grad_tile_kernel(GradArgs a) {
  grad_tile_body<KIND, NC, 0>(a);
}
// The interior tiles of the launch's enumeration, run by run (the same partition into workgroup runs as the direct kernel's
// full enumeration): every run is cut into its tile rows, every row piece is one grad_interior_segment.
template <int KIND, int NC>
__global__ __launch_bounds__(256, GMB_GRI_WAVES) void grad_interior_kernel(GradArgs a) {
  __shared__ __attribute__((aligned(16))) double lds[GradInteriorLds<NC>::DOUBLES];
  __shared__ double swave[4][GRAD_SMALL];
  const int tid = threadIdx.x, wave = tid >> 6;
  double g_ls2[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) g_ls2[k] = 0.0;
  double g_eta2 = 0.0;
  long long tile = (long long)blockIdx.x * a.per;
  long long t_end = tile + a.per;
  t_end = t_end < a.total_tiles ? t_end : a.total_tiles;
  auto before = [&](int q) -> long long { return (long long)q * (a.row_first + 1) + (long long)a.row_stride * q * (q - 1) / 2; };
  int m = 0;
  if (tile < t_end) {
    const double sd = (double)a.row_stride, f1 = (double)a.row_first + 1.0 - 0.5 * sd;
    m = (int)((__builtin_sqrt(f1 * f1 + 2.0 * sd * (double)tile) - f1) / sd);
    m = m < 0 ? 0 : m;
    while (before(m) > tile) --m;
    while (before(m + 1) <= tile) ++m;
  }
  int tix = a.row_first + m * a.row_stride;
  int tjx = (int)(tile - before(m));
  while (tile < t_end) {
    const long long left_in_row = (long long)(tix - tjx + 1);  // tiles of this row from tjx on, the diagonal one included
    const long long nseg = left_in_row < t_end - tile ? left_in_row : t_end - tile;
    const int tj1 = (int)(tjx + nseg) < tix ? (int)(tjx + nseg) : tix;  // interior tiles end at the diagonal
    if (tjx < tj1 && (int64_t)(tix + 1) * TILE <= a.pts.n) grad_interior_segment<KIND, NC>(a, tix, tjx, tj1, lds, g_ls2, g_eta2);
    tile += nseg;
    tjx += (int)nseg;
    if (tjx > tix) {
      tix += a.row_stride;
      tjx = 0;
    }
  }
  // units of the direct loops: d r2 / d ls_k = -2 d2_k / ls_k; eta^2 k -> 2 eta k.  Wave sums (fixed shuffle tree), waves in order.
  auto wave_sum = [&](double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
  };
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const double v = wave_sum(-2.0 * a.inv_ls[k] * g_ls2[k]);
    if ((tid & 63) == 0) swave[wave][k] = v;
  }
  {
    const double v = wave_sum((2.0 / a.eta) * g_eta2);
    if ((tid & 63) == 0) {
      swave[wave][NC] = v;
      swave[wave][NC + 1] = 0.0;  // tau: no linear term on this path
    }
  }
  __syncthreads();
  double* out = a.part + ((int64_t)a.part_block0 + blockIdx.x) * a.part_stride;
  if (tid < NC + 2) out[tid] = (swave[0][tid] + swave[1][tid]) + (swave[2][tid] + swave[3][tid]);
}

// Second stage: out[dst(q)] = sum over the nparts partial vectors of slot q, in a FIXED order (thread t adds parts
// t, t + 256, ...; then a fixed LDS tree) -- run to run, the same bits.  Dense slot q of the partial vectors maps
// to out[dst[r] + (q - dense[r])] for the range r that contains it; slots outside every range are dropped (the
// zero-padded coordinates of the compile-time dimension count).
struct GradRanges {
  int32_t n;
  int32_t dense[MAX_TABS + 2], count[MAX_TABS + 2], dst[MAX_TABS + 2];
};
__global__ __launch_bounds__(256) void grad_sum_partials_kernel(const double* __restrict__ part, int nparts, int64_t stride,
                                                                GradRanges r, double* __restrict__ acc) {
  __shared__ double red[256];
  const int q = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nparts; b += 256) s += part[(int64_t)b * stride + q];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  for (int i = 0; i < r.n; ++i)
    if (q >= r.dense[i] && q < r.dense[i] + r.count[i]) {
      acc[r.dst[i] + (q - r.dense[i])] = red[0];
      return;
    }
}
// shared lengthscale: every dimension's partial belongs to the one parameter -- acc[0] = sum_k tmp[k], fixed order
__global__ void grad_fold_ls_kernel(double* acc, int n, const double* tmp) {
  if (threadIdx.x || blockIdx.x) return;
  double s = 0.0;
  for (int k = 0; k < n; ++k) s += tmp[k];
  acc[0] = s;
}

// Diagonal-only terms: d/d sigma and the noise-table partials.
//   out[0] = sum_i M_ii * 2 sigma * nmult_i ;  out[1 + a] = sum_{i: out(i) = a} M_ii * sigma^2
// One workgroup, fixed order: thread t takes rows t, t + 1024, ...; the noise-table partials go through per-wave
// LDS copies (lanes of one instruction are served in lane order), summed wave by wave.
__global__ __launch_bounds__(1024) void grad_diag_kernel(const double* Z, int64_t ldz,
                                                         const double* alpha, PointSet pts,
                                                         CovParams p, double sigma, double* out, int row_first,
                                                         int row_stride, int z_packed = 0) {
  __shared__ double red[16];
  __shared__ double tab[16][32];
  const int wave = threadIdx.x >> 6;
  for (int idx = threadIdx.x; idx < 16 * 32; idx += 1024) (&tab[0][0])[idx] = 0.0;
  __syncthreads();
  double gs = 0.0;
  for (int64_t i = threadIdx.x; i < pts.n; i += 1024) {
    if ((int)((i >> 7) % row_stride) != row_first) continue;  // block rows of this shard only
    const double a = alpha[i];
    const int64_t zr = z_packed ? (((i >> 7) - row_first) / row_stride) * TILE + (i & 127) : i;
    const double m = 0.5 * (Z[zr + i * ldz] - a * a);
    double mult = 1.0;
    if (p.noise_tab >= 0) {
      const int c = pts.cat[(int64_t)p.noise_tab * pts.npad + i];
      mult = p.noise_mult[c];
      atomicAdd(&tab[wave][c], m * sigma * sigma);
    }
    gs = fma(m, 2.0 * sigma * mult, gs);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) gs += __shfl_down(gs, off);
  if ((threadIdx.x & 63) == 0) red[wave] = gs;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < 16; ++w) s += red[w];
    out[0] = s;
  }
  if (p.noise_tab >= 0 && threadIdx.x < 32) {
    double s = 0.0;
    for (int w = 0; w < 16; ++w) s += tab[w][threadIdx.x];
    out[1 + threadIdx.x] = s;
  }
}

// The factor buffer carries y (later v) in row N, so the block inversion leaves -v^T L^-1 in the
// padding rows [n, npad) of W; reset them to identity before forming W^T W.
__global__ void reset_pad_rows_kernel(double* W, int64_t ld, int64_t n, int64_t npad) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= npad) return;
  for (int64_t r = n; r < npad; ++r) W[r + c * ld] = (r == c) ? 1.0 : 0.0;
}

// dst[c + r*ldd] = src[r + c*lds] for an (rows x cols) block (both column-major); 32 x 32 tiles
// through LDS so that reads and writes are both contiguous along the fast index.
__global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ src, int64_t lds_,
                                                        double* __restrict__ dst, int64_t ldd, int rows,
                                                        int cols) {
  __shared__ double tile[32][33];
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8)
    if (r0 + tx < rows && c0 + j < cols) tile[j][tx] = src[(r0 + tx) + (int64_t)(c0 + j) * lds_];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (c0 + tx < cols && r0 + j < rows) dst[(c0 + tx) + (int64_t)(r0 + j) * ldd] = tile[tx][j];
}

// Multi-GPU inverse: a rank's block rows of the identity (block t of V is global block row
// first + t*stride), the right-hand side of V <- V L^-T that yields its rows of U = L^-T ...
__global__ void identity_rows_kernel(double* V, int64_t ldv, int first, int stride) {
  const int t = blockIdx.x, i = threadIdx.x;
  const int64_t col = ((int64_t)first + (int64_t)t * stride) * TILE + i;
  V[(int64_t)t * TILE + i + col * ldv] = 1.0;
}
// ... and its share of alpha = U v: 64 rows x 4 k-phases per workgroup, fixed summation order
__global__ __launch_bounds__(256) void urows_v_kernel(const double* __restrict__ V, int64_t ldv,
                                                      const double* __restrict__ v, int64_t n,
                                                      double* __restrict__ alpha_rows) {
  __shared__ double part[4][64];
  const int r = threadIdx.x & 63, ph = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 64 + r;
  double s = 0.0;
  for (int64_t k = ph; k < n; k += 4) s = fma(V[row + k * ldv], v[k], s);
  part[ph][r] = s;
  __syncthreads();
  if (ph == 0) alpha_rows[row] = (part[0][r] + part[1][r]) + (part[2][r] + part[3][r]);
}

// Batched transposes (one level of the inverse tree): blockIdx.z selects the descriptor.
struct TransposeJob {
  const double* src;
  double* dst;
  int32_t rows, cols;
};
__global__ __launch_bounds__(256) void transpose_batched_kernel(const TransposeJob* __restrict__ jobs, int64_t lds_,
                                                                int64_t ldd) {
  __shared__ double tile[32][33];
  const TransposeJob jb = jobs[blockIdx.z];
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  if (r0 >= jb.rows || c0 >= jb.cols) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8)
    if (r0 + tx < jb.rows && c0 + j < jb.cols) tile[j][tx] = jb.src[(r0 + tx) + (int64_t)(c0 + j) * lds_];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (c0 + tx < jb.cols && r0 + j < jb.rows) jb.dst[(c0 + tx) + (int64_t)(r0 + j) * ldd] = tile[tx][j];
}

// U = L^-T lives in the upper triangle of the factor buffer; its padding COLUMNS [n, npad) pick up
// the y row's pollution through the transposes -- reset them to identity.
__global__ void reset_pad_cols_kernel(double* U, int64_t ld, int64_t n, int64_t npad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npad) return;
  for (int64_t k = n; k < npad; ++k) U[i + k * ld] = (i == k) ? 1.0 : 0.0;
}

// Gather (to_packed) / scatter block rows between a column-major matrix and a packed buffer -- the send and
// receive sides of the all-gathers of the multi-GPU path (dist_driver.hpp), where rank q owns the block rows
// b = q (mod G):
//   packed[seg*seg_elems + (t*128 + r) + c*ldp]  <->  mat[(first(seg) + t*stride)*128 + r + c*ld],
//   t < count(seg), r < 128, c < ncols.
// One segment (by_rank = 0: this rank packing its own rows; first / count given) or, for the receive side,
// by_rank = 1 with nseg = G segments in one launch: segment q holds the block rows of [lo, hi) that rank q owns, i.e.
// first(q) = lo + ((q - lo) mod G), count(q) = ceil((hi - first(q)) / G).  16 bytes per lane.
struct PackArgs {
  double* mat;
  int64_t ld;
  double* packed;
  int64_t ldp;        // leading dimension of one packed segment (rows)
  int64_t seg_elems;  // doubles between consecutive segments
  int32_t ncols;
  int32_t to_packed;
  int32_t nseg;       // segments in the packed buffer (grid z)
  int32_t by_rank;    // 0: one segment, (first, count) below; 1: segment q = rank q's block rows of [lo, hi)
  int32_t first, count, stride;
  int32_t lo, hi;
};
__global__ __launch_bounds__(256) void pack_rows_kernel(PackArgs a) {
  const int seg = blockIdx.z;
  int first = a.first, count = a.count;
  if (a.by_rank) {
    const int G = a.stride;
    first = a.lo + (((seg - a.lo) % G) + G) % G;
    count = first < a.hi ? (a.hi - first + G - 1) / G : 0;
  }
  const int t = blockIdx.x;
  if (t >= count) return;
  const int c = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (c >= a.ncols) return;
  const int r = (threadIdx.x & 63) * 2;
  double* m = a.mat + ((int64_t)first + (int64_t)t * a.stride) * 128 + r + (int64_t)c * a.ld;
  double* q = a.packed + (int64_t)seg * a.seg_elems + (int64_t)t * 128 + r + (int64_t)c * a.ldp;
  typedef double pk_d2 __attribute__((ext_vector_type(2)));
  if (a.to_packed) *reinterpret_cast<pk_d2*>(q) = *reinterpret_cast<const pk_d2*>(m);
  else *reinterpret_cast<pk_d2*>(m) = *reinterpret_cast<const pk_d2*>(q);
}

// Save (to_save) / restore the 128 x 128 diagonal blocks of the factor buffer: the gradient keeps U = L^-T in the
// upper triangle AND the diagonal tiles of that buffer, so putting the diagonal blocks of L back afterwards
// leaves the factorisation intact (the strictly lower blocks are never touched) -- predict() after a
// gradient evaluation needs no re-factorisation.
__global__ __launch_bounds__(256) void diag_blocks_copy_kernel(double* __restrict__ A, int64_t ld,
                                                               double* __restrict__ save, int to_save) {
  double* blk = A + (int64_t)blockIdx.x * 128 * (ld + 1);
  double* sv = save + (int64_t)blockIdx.x * 128 * 128;
  typedef double dg_d2 __attribute__((ext_vector_type(2)));
  const int r = (threadIdx.x & 63) * 2, c0 = threadIdx.x >> 6;
  for (int c = c0; c < 128; c += 4) {
    dg_d2* m = reinterpret_cast<dg_d2*>(blk + r + (int64_t)c * ld);
    dg_d2* q = reinterpret_cast<dg_d2*>(sv + r + c * 128);
    if (to_save) *q = *m;
    else *m = *q;
  }
}

// alpha = W^T v with W = L^-1 lower triangular, column-major: alpha_j = sum_{k>=j} W[k + j*ld] v_k
__global__ __launch_bounds__(256) void wt_v_kernel(const double* W, int64_t ld, const double* v,
                                                   int64_t n, double* alpha) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t j = (int64_t)blockIdx.x * 4 + wave;
  if (j >= n) return;
  const double* col = W + j * ld;
  double s = 0.0;
  for (int64_t k = j + lane; k < n; k += 64) s = fma(col[k], v[k], s);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if (lane == 0) alpha[j] = s;
}

}  // namespace gmb
