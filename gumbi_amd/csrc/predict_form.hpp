// predict_form.hpp -- the posterior solves as ONE plain product when the inverse factor is already there.
//
// pm.gp.Marginal.predict (gumbi/regression/pymc/GP.py:845-847) forms A = L^-1 K(X, X*) by a triangular solve.  Behind a MAP
// fit the last objective + gradient evaluation has left U = L^-T (eval_tiles.hpp: by rows, strictly upper tiles in the
// factor buffer's upper triangle, diagonal tiles in a side buffer), so the same A^T = K(X*, X) L^-T is a GEMM with a
// triangular operand -- the launch that runs at 0.93 of the f64 matrix peak instead of the tile solve's 0.73 (N = 10k,
// M = 10^4).  Both operands of the engine's GEMM are k-major (gemm_f64.hpp), and U keeps the CONTRACTION index fastest, so
// the inverse is transposed once per fit into the lower triangle of the gradient's Sigma^-1 buffer (dead by then):
//
//   Linv(c, i) = U(i, c)            linv_from_u_kernel: tile (c, r) of Linv = tile (r, c) of U, transposed
//   Vt(c, m)   = sum_{i <= c} Linv(c, i) K(x*_m, x_i)      gemm_f64: n = c, m = test point, k = i, k_hi per n-tile (khi_n)
//   mean_m = sum_c Vt(c, m) v_c,   var_m = k** - sum_c Vt(c, m)^2        predict_rows_kernel: one workgroup per test point,
//                                                                         contiguous reads, fixed-order sums
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_f64.hpp"

namespace gmb {

struct LinvArgs {
  const double* U;      // factor buffer: tile (r, c), r < c, at U + r * 128 + c * 128 * ldu
  int64_t ldu;
  const double* udiag;  // tile (r, r) at udiag + r * 128 * 128, leading dimension 128
  double* W;            // out: tile (c, r) at W + c * 128 + r * 128 * ldw
  int64_t ldw;
  int32_t nct;
};

// grid: nct (nct + 1) / 2 workgroups of 256 threads; workgroup b -> (r, c), c >= r, enumerated row by row
__global__ __launch_bounds__(256) void linv_from_u_kernel(LinvArgs a) {
  __shared__ double tile[32][33];
  // b = r * nct - r (r - 1) / 2 + (c - r)
  const int b = blockIdx.x;
  int r = (int)((2.0 * a.nct + 1.0 - sqrt((2.0 * a.nct + 1.0) * (2.0 * a.nct + 1.0) - 8.0 * (double)b)) * 0.5);
  if (r < 0) r = 0;
  if (r > a.nct - 1) r = a.nct - 1;
  auto row_start = [&](int q) { return q * a.nct - q * (q - 1) / 2; };
  while (r > 0 && row_start(r) > b) --r;
  while (r + 1 < a.nct && row_start(r + 1) <= b) ++r;
  const int c = r + (b - row_start(r));
  const double* src = r == c ? a.udiag + (int64_t)r * TILE * TILE : a.U + (int64_t)r * TILE + (int64_t)c * TILE * a.ldu;
  const int64_t lds_ = r == c ? (int64_t)TILE : a.ldu;
  double* dst = a.W + (int64_t)c * TILE + (int64_t)r * TILE * a.ldw;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int sr = 0; sr < TILE; sr += 32)
    for (int sc = 0; sc < TILE; sc += 32) {
      for (int j = ty; j < 32; j += 8) tile[j][tx] = src[(sr + tx) + (int64_t)(sc + j) * lds_];
      __syncthreads();
      for (int j = ty; j < 32; j += 8) dst[(sc + tx) + (int64_t)(sr + j) * a.ldw] = tile[tx][j];
      __syncthreads();
    }
}

// mean / variance of test point m = blockIdx.x from column m of Vt (n valid rows, contiguous): 256 threads take the rows
// t, t + 256, ... in order, then a fixed tree -- the same bits run to run
__global__ __launch_bounds__(256) void predict_rows_kernel(const double* __restrict__ Vt, int64_t ldv, const double* __restrict__ v, int64_t n,
                                                           const double* __restrict__ kss, double* __restrict__ mean,
                                                           double* __restrict__ var) {
  __shared__ double red[2][4];
  const int64_t m = blockIdx.x;
  const double* col = Vt + m * ldv;
  double am = 0.0, as = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const double x = col[i];
    am = fma(x, v[i], am);
    as = fma(x, x, as);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    am += __shfl_down(am, off);
    as += __shfl_down(as, off);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = am;
    red[1][threadIdx.x >> 6] = as;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    mean[m] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    var[m] = kss[m] - ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
  }
}

}  // namespace gmb
