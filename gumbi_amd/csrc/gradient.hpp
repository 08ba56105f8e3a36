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
  double* acc;      // global accumulators (atomicAdd)
  int32_t tab_acc_off[MAX_TABS];
  // shard of the lower triangle this launch reduces: block rows row_first, row_first + row_stride, ...
  // (0, 1 = everything; a rank of the multi-GPU gradient passes (rank, world))
  int32_t row_first, row_stride;
  // z_packed: Z holds ONLY the owned block rows, packed (row block t of Z = block row row_first + t*row_stride)
  int32_t z_packed;
};

constexpr int GRAD_MAX_LDS_ACC = 16 + 2 + MAX_LIN + MAX_TABS * 64;  // tables up to 8 levels in LDS

template <int KIND, int NC>
__global__ __launch_bounds__(256) void grad_tile_kernel(GradArgs a) {
  __shared__ double xj[NC][TILE];
  __shared__ double lj[MAX_LIN][TILE];
  __shared__ double li[MAX_LIN][TILE];
  __shared__ int32_t cj[MAX_TABS][TILE];
  __shared__ int32_t ci[MAX_TABS][TILE];
  __shared__ double aj[TILE];
  __shared__ double sacc[GRAD_MAX_LDS_ACC];

  // lower-triangle tile pair (ti >= tj), enumerated block row by owned block row
  // (closed form: owned row m = (tix - row_first) / row_stride is preceded by m (row_first + 1) + row_stride m (m - 1) / 2
  // tiles; a linear search cost the late rows of a large matrix thousands of scalar cycles per workgroup)
  const double sd = (double)a.row_stride, f1 = (double)a.row_first + 1.0 - 0.5 * sd;
  int m = (int)((__builtin_sqrt(f1 * f1 + 2.0 * sd * (double)blockIdx.x) - f1) / sd);
  auto before = [&](int q) -> long long { return (long long)q * (a.row_first + 1) + (long long)a.row_stride * q * (q - 1) / 2; };
  m = m < 0 ? 0 : m;
  while (before(m) > (long long)blockIdx.x) --m;
  while (before(m + 1) <= (long long)blockIdx.x) ++m;
  const int tix = a.row_first + m * a.row_stride;
  if (tix >= a.tiles) return;
  const int tjx = (int)((long long)blockIdx.x - before(m));
  const int64_t gi0 = (int64_t)tix * TILE, gj0 = (int64_t)tjx * TILE;
  const int tid = threadIdx.x;
  const int il = tid & (TILE - 1), jh = tid >> 7;
  const int64_t gi = gi0 + il;
  const CovParams& p = a.p;
  const int n_acc_small = NC + 2 + p.n_lin;

  for (int idx = tid; idx < GRAD_MAX_LDS_ACC; idx += 256) sacc[idx] = 0.0;
  for (int idx = tid; idx < NC * TILE; idx += 256) {
    const int k = idx / TILE, j = idx - k * TILE;
    xj[k][j] = a.pts.xs[(int64_t)k * a.pts.npad + gj0 + j];
  }
  for (int idx = tid; idx < p.n_lin * TILE; idx += 256) {
    const int k = idx / TILE, j = idx - k * TILE;
    lj[k][j] = a.pts.xl[(int64_t)k * a.pts.npad + gj0 + j];
    li[k][j] = a.pts.xl[(int64_t)k * a.pts.npad + gi0 + j];
  }
  for (int idx = tid; idx < p.n_tab * TILE; idx += 256) {
    const int t = idx / TILE, j = idx - t * TILE;
    cj[t][j] = a.pts.cat[(int64_t)t * a.pts.npad + gj0 + j];
    ci[t][j] = a.pts.cat[(int64_t)t * a.pts.npad + gi0 + j];
  }
  if (tid < TILE) aj[tid] = (gj0 + tid < a.pts.n) ? a.alpha[gj0 + tid] : 0.0;
  double xi[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) xi[k] = a.pts.xs[(int64_t)k * a.pts.npad + gi];
  const bool row_real = gi < a.pts.n;
  const double ai = row_real ? a.alpha[gi] : 0.0;
  __syncthreads();

  double g_ls[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) g_ls[k] = 0.0;
  double g_eta = 0.0, g_tau = 0.0;

  const int64_t zrow = a.z_packed ? (int64_t)((tix - a.row_first) / a.row_stride) * TILE + il : gi;
  const double* zp = a.Z + zrow + (gj0 + jh * (TILE / 2)) * a.ldz;
  // Fast path (almost every tile): stationary term only, tile strictly below the diagonal, every
  // row and column real -- each entry stands for (i,j) and (j,i), no per-entry conditionals.
  const bool fast = p.n_lin == 0 && p.n_tab == 0 && tix > tjx && gi0 + TILE <= a.pts.n && gj0 + TILE <= a.pts.n;
  if (fast) {
    const int jb = jh * (TILE / 2);
#pragma unroll 4
    for (int jj = 0; jj < TILE / 2; ++jj) {
      const double m = zp[(int64_t)jj * a.ldz] - ai * aj[jb + jj];  // 2 * M_ij
      double d2[NC];
      double r2 = 0.0;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        const double d = xi[k] - xj[k][jb + jj];
        d2[k] = d * d;
        r2 += d2[k];
      }
      const double ks = stationary<KIND>(r2);
      const double mdk = m * (p.eta2 * stationary_dr2<KIND>(r2));
#pragma unroll
      for (int k = 0; k < NC; ++k) g_ls[k] = fma(mdk, -2.0 * d2[k] * a.inv_ls[k], g_ls[k]);
      g_eta = fma(m, 2.0 * a.eta * ks, g_eta);
    }
  }
  for (int jj = fast ? TILE / 2 : 0; jj < TILE / 2; ++jj) {
    const int j = jh * (TILE / 2) + jj;
    const int64_t gj = gj0 + j;
    if (!row_real || gj >= a.pts.n || gj > gi) continue;
    const double mfull = 0.5 * (zp[(int64_t)jj * a.ldz] - ai * aj[j]);  // M_ij
    const double m = (gi == gj) ? mfull : 2.0 * mfull;                   // (i,j) and (j,i)
    double d2[NC];
    double r2 = 0.0;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const double d = xi[k] - xj[k][j];
      d2[k] = d * d;
      r2 += d2[k];
    }
    const double ks = stationary<KIND>(r2);
    const double dk = p.eta2 * stationary_dr2<KIND>(r2);
    double lin = 0.0;
    for (int k = 0; k < p.n_lin; ++k) lin = fma(li[k][il], lj[k][j], lin);
    double F = 1.0;
    for (int t = 0; t < p.n_tab; ++t)
      F *= p.tabs[p.tab_off[t] + ci[t][il] * p.tab_levels[t] + cj[t][j]];
    const double mF = m * F;
    // d r2 / d ls_k = -2 d2_k / ls_k   (d2 already in scaled units)
#pragma unroll
    for (int k = 0; k < NC; ++k) g_ls[k] = fma(mF * dk, -2.0 * d2[k] * a.inv_ls[k], g_ls[k]);
    g_eta = fma(mF, 2.0 * a.eta * ks, g_eta);
    if (p.n_lin > 0) {
      g_tau = fma(mF, lin, g_tau);
      for (int k = 0; k < p.n_lin; ++k)
        atomicAdd(&sacc[NC + 2 + k], -mF * p.tau * (li[k][il] + lj[k][j]));
    }
    if (p.n_tab > 0) {
      const double base = p.eta2 * ks + p.tau * lin;
      for (int t = 0; t < p.n_tab; ++t) {
        double others = 1.0;
        for (int t2 = 0; t2 < p.n_tab; ++t2)
          if (t2 != t) others *= p.tabs[p.tab_off[t2] + ci[t2][il] * p.tab_levels[t2] + cj[t2][j]];
        const int L = p.tab_levels[t];
        const double val = mfull * base * others;
        // ordered pair (i,j) feeds G[a][b]; its mirror (j,i) feeds G[b][a] (off-diagonal only)
        const int ca = ci[t][il], cb = cj[t][j];
        const bool off = gi != gj;
        if (L <= 8) {
          atomicAdd(&sacc[n_acc_small + t * 64 + ca * L + cb], val);
          if (off) atomicAdd(&sacc[n_acc_small + t * 64 + cb * L + ca], val);
        } else {
          atomicAdd(&a.acc[a.tab_acc_off[t] + ca * L + cb], val);
          if (off) atomicAdd(&a.acc[a.tab_acc_off[t] + cb * L + ca], val);
        }
      }
    }
  }
  // block reduction of the register accumulators
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    double v = g_ls[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((tid & 63) == 0) atomicAdd(&sacc[k], v);
  }
  {
    double v = g_eta, u = g_tau;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      v += __shfl_down(v, off);
      u += __shfl_down(u, off);
    }
    if ((tid & 63) == 0) {
      atomicAdd(&sacc[NC], v);
      atomicAdd(&sacc[NC + 1], u);
    }
  }
  __syncthreads();
  // flush: ARD keeps one slot per dim; shared lengthscale sums all dims into slot 0
  if (tid < NC) {
    if (a.ard) {
      if (tid < a.nc_real) atomicAdd(&a.acc[tid], sacc[tid]);
    } else if (tid == 0) {
      double s = 0.0;
      for (int k = 0; k < NC; ++k) s += sacc[k];
      atomicAdd(&a.acc[0], s);
    }
  }
  const int n_ls_out = a.ard ? a.nc_real : 1;
  if (tid == 0) {
    atomicAdd(&a.acc[n_ls_out], sacc[NC]);
    atomicAdd(&a.acc[n_ls_out + 1], sacc[NC + 1]);
  }
  if (tid < p.n_lin) atomicAdd(&a.acc[n_ls_out + 2 + tid], sacc[NC + 2 + tid]);
  for (int t = 0; t < p.n_tab; ++t) {
    const int L = p.tab_levels[t];
    if (L <= 8 && tid < L * L) {
      const double v = sacc[n_acc_small + t * 64 + tid];
      if (v != 0.0) atomicAdd(&a.acc[a.tab_acc_off[t] + tid], v);
    }
  }
}

// Diagonal-only terms: d/d sigma and the noise-table partials.
//   out[0] += sum_i M_ii * 2 sigma * nmult_i ;  out[1 + a] += sum_{i: out(i) = a} M_ii * sigma^2
__global__ __launch_bounds__(256) void grad_diag_kernel(const double* Z, int64_t ldz,
                                                        const double* alpha, PointSet pts,
                                                        CovParams p, double sigma, double* out, int row_first,
                                                        int row_stride, int z_packed = 0) {
  __shared__ double red[4];
  double gs = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pts.n; i += (int64_t)gridDim.x * 256) {
    if ((int)((i >> 7) % row_stride) != row_first) continue;  // block rows of this shard only
    const double a = alpha[i];
    const int64_t zr = z_packed ? (((i >> 7) - row_first) / row_stride) * TILE + (i & 127) : i;
    const double m = 0.5 * (Z[zr + i * ldz] - a * a);
    double mult = 1.0;
    if (p.noise_tab >= 0) {
      const int c = pts.cat[(int64_t)p.noise_tab * pts.npad + i];
      mult = p.noise_mult[c];
      atomicAdd(&out[1 + c], m * sigma * sigma);
    }
    gs = fma(m, 2.0 * sigma * mult, gs);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) gs += __shfl_down(gs, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gs;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&out[0], red[0] + red[1] + red[2] + red[3]);
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
