// Triangular solve of a row strip against one factored 128 x 128 diagonal block, on the matrix
// pipe and without shared memory:
//
//     B  <-  B inv(L_kk)^T          B: nrows x 128 (column-major, rows contiguous)
//
// This is the step between two consecutive diagonal-block factorisations of the Cholesky
// (panel rows below the block), the leaf of the predict solve V <- K_* L^-T, and the packed
// row solve of the multi-GPU driver.  Each wavefront owns 16 rows.  Written for rows as columns
// (Y = B^T, 128 x 16), the solve L_kk X = Y is a blocked forward substitution over the eight
// 16 x 16 sub-blocks:
//
//     X_s = inv(L_ss) (Y_s - sum_{t<s} L_st X_t)
//
// Every X_t lives in the v_mfma_f64_16x16x4 accumulator layout (lane = row of B, register =
// sub-block row), which is exactly the B-operand layout of the next product, so the eight tiles
// never leave registers: 144 MFMAs per wavefront, operands L_st / inv(L_ss) read straight from
// global memory (the block is shared by every wavefront of the launch and sits in L2).
// inv(L_ss) are the sub-block inverses the leaf factorisation already has (potrf_leaf.hpp); the
// 128 x 128 inverse is not needed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gmb {

struct TrsmArgs {
  double* B;             // in/out
  int64_t ldb;
  int64_t nrows;         // multiple of 16
  const double* L;       // factored diagonal block, column-major
  int64_t ldl;
  const double* dinv16;  // 8 x (16 x 16) column-major sub-block inverses (identity-padded)
  int32_t nvalid;        // rows/columns >= nvalid of the block are identity padding (whatever
                         // the buffer holds there -- the y row of the factor buffer -- is ignored)
};

typedef double strip_d4 __attribute__((ext_vector_type(4)));

// The slab's eight 16 x 16 tiles in accumulator layout (independent of the diagonal block: a fused
// kernel issues these loads before it waits for the leaf)
__device__ __forceinline__ void trsm_strip_load(const TrsmArgs& g, const int64_t r0, strip_d4 (&X)[8]) {
  const int lane = threadIdx.x & 63;
  const int r16 = lane & 15, kq = lane >> 4;
  const double* Bp = g.B + r0 + r16;
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) X[s][q] = Bp[(int64_t)(16 * s + kq + 4 * q) * g.ldb];
}

// WT: write-through (sc1) stores of the solved slab -- a result another workgroup of the same launch will read
// (chol_tiles.hpp; see leaf_store in potrf_leaf.hpp)
template <bool WT = false>
__device__ __forceinline__ void trsm_strip_solve_store(const TrsmArgs& g, const int64_t r0, strip_d4 (&X)[8]) {
  typedef strip_d4 d4;
  const int lane = threadIdx.x & 63;
  const int r16 = lane & 15, kq = lane >> 4;
  double* Bp = g.B + r0 + r16;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const bool live = 16 * s + r16 < g.nvalid;
    const double* Lrow = g.L + 16 * s + r16;
    d4 y = X[s];
#pragma unroll
    for (int t = 0; t < s; ++t)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        double a = Lrow[(int64_t)(16 * t + 4 * kk + kq) * g.ldl];
        a = live ? -a : 0.0;
        y = __builtin_amdgcn_mfma_f64_16x16x4f64(a, X[t][kk], y, 0, 0, 0);
      }
    d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const double a = g.dinv16[s * 256 + (4 * kk + kq) * 16 + r16];
      x = __builtin_amdgcn_mfma_f64_16x16x4f64(a, y[kk], x, 0, 0, 0);
    }
    X[s] = x;
  }
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if constexpr (WT) __hip_atomic_store(&Bp[(int64_t)(16 * s + kq + 4 * q) * g.ldb], X[s][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else Bp[(int64_t)(16 * s + kq + 4 * q) * g.ldb] = X[s][q];
    }
}

// The same solve with the operand loads written out ahead of their use: step s + 1's L_st / inv(L_ss) values are
// requested before step s's MFMAs (two register sets, <= 32 values each).  Inside the persistent tile Cholesky's task
// functions (chol_tiles.hpp) the compiler kept the source order of trsm_strip_solve_store -- load, wait, MFMA, load,
// wait, ... 144 exposed L2 latencies, 40 us per slab instead of 9 -- although it batches the loads of the stand-alone
// kernel by itself.
template <bool WT = false>
__device__ __forceinline__ void trsm_strip_solve_store_pf(const TrsmArgs& g, const int64_t r0, strip_d4 (&X)[8]) {
  typedef strip_d4 d4;
  const int lane = threadIdx.x & 63;
  const int r16 = lane & 15, kq = lane >> 4;
  double* Bp = g.B + r0 + r16;
  double ops[2][32];
  auto load_step = [&](const int s, double (&o)[32]) {
    const double* Lrow = g.L + 16 * s + r16;
#pragma unroll
    for (int t = 0; t < s; ++t)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) o[4 * t + kk] = Lrow[(int64_t)(16 * t + 4 * kk + kq) * g.ldl];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) o[28 + kk] = g.dinv16[s * 256 + (4 * kk + kq) * 16 + r16];
  };
  // sub-blocks that are identity padding altogether (a ragged last diagonal block: 16 s >= nvalid) leave their X as it is --
  // the products below would add exact zeros and multiply by exact ones: same bits, none of the loads and MFMAs
  const int smax = (g.nvalid + 15) / 16;
  load_step(0, ops[0]);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (s >= smax) break;
    if (s < 7) load_step(s + 1, ops[(s + 1) & 1]);
    asm volatile("" ::: "memory");  // keep the requests above the MFMAs below
    const double(&o)[32] = ops[s & 1];
    const bool live = 16 * s + r16 < g.nvalid;
    d4 y = X[s];
#pragma unroll
    for (int t = 0; t < s; ++t)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) y = __builtin_amdgcn_mfma_f64_16x16x4f64(live ? -o[4 * t + kk] : 0.0, X[t][kk], y, 0, 0, 0);
    d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) x = __builtin_amdgcn_mfma_f64_16x16x4f64(o[28 + kk], y[kk], x, 0, 0, 0);
    X[s] = x;
  }
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if constexpr (WT) __hip_atomic_store(&Bp[(int64_t)(16 * s + kq + 4 * q) * g.ldb], X[s][q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else Bp[(int64_t)(16 * s + kq + 4 * q) * g.ldb] = X[s][q];
    }
}

// one 16-row slab (rows r0 .. r0+15) by the calling wavefront
__device__ __forceinline__ void trsm_strip_slab(const TrsmArgs& g, const int64_t r0) {
  strip_d4 X[8];
  trsm_strip_load(g, r0, X);
  trsm_strip_solve_store(g, r0, X);
}

__global__ __launch_bounds__(256) void trsm_strip_kernel(TrsmArgs g) {
  __builtin_amdgcn_s_setprio(3);  // part of the latency-bound chain: go first on a shared compute unit
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
  if (r0 >= g.nrows) return;
  trsm_strip_slab(g, r0);
}

}  // namespace gmb
