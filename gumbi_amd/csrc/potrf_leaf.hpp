// potrf_leaf.hpp -- diagonal-block leaf of the blocked Cholesky: factor one 128 x 128 block in
// LDS and produce the inverse of its triangular factor, so that every panel / predict
// triangular solve against this block becomes a plain MFMA GEMM (gemm_f64.hpp).
//
// Replaces the unblocked dpotf2 + dtrtri work LAPACK does inside dpotrf / dtrsm for
// pm.gp.Marginal (call sites gumbi/regression/pymc/GP.py:580, 845-847).
//
// One workgroup of 256 threads (4 waves) owns the block.  The block lives column-major in LDS
// (pitch 130 doubles).  Factorisation is right-looking over 16-column sub-panels:
//   1. wave 0 factors the 16 x 16 diagonal sub-block in registers (row per lane, pivots and
//      multipliers moved with v_readlane -- no LDS round trips on the dependent chain);
//   2. waves 1-3 solve the sub-panel rows by forward substitution (row per thread, the 16 x 16
//      factor is read with LDS broadcasts);
//   3. all waves apply the rank-16 update with v_mfma_f64_16x16x4_f64, one 16 x 16 tile per
//      wave at a time, operands read straight from the column-major block.
// Rows >= nvalid of the block are "panel rows" (the appended y row that carries v = L^-1 y, and
// zero padding): they are solved but never used as pivots; columns >= nvalid are treated as
// identity and never written back.
// Inversion: X = L^-1 by 16 x 16 blocks, block-diagonal by block-diagonal,
//   X_ij = -X_ii * (sum_{k=j..i-1} L_ik X_kj), both products on MFMA.  X is kept transposed in
// the (free) upper triangle of the LDS block, X_ii additionally as dense 16 x 16 tiles.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_f64.hpp"

namespace gmb {

constexpr int LB = 128;  // leaf block edge
constexpr int LP = 130;  // LDS pitch (doubles)
constexpr int SB = 16;   // sub-panel width

// Broadcast one lane's value to every lane.  ds_bpermute keeps the result in a VGPR: the
// v_readlane form returns SGPR pairs, and the ~250 multipliers of a 16 x 16 block overflowed the
// scalar file (521 SGPR spills, each a v_writelane + v_readlane round trip).  `idx4` is the source
// lane times 4, held in a VGPR the optimiser cannot see through (it would fold a constant index
// straight back into v_readlane).
__device__ __forceinline__ double bcast_f64(double v, int idx4) {
  union {
    double d;
    int i[2];
  } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_ds_bpermute(idx4, u.i[0]);
  u.i[1] = __builtin_amdgcn_ds_bpermute(idx4, u.i[1]);
  return u.d;
}

// acc(16x16) += sum_{e<K} Aop[m][e] * Bop[n][e]; operands k-major in LDS:
// Aop(m,e) at a[e*lda + m], Bop(n,e) at b[e*ldb + n].  Result: lane holds n = lane&15,
// m = (lane>>4) + 4*reg.
__device__ __forceinline__ d4 mfma_tile_k16(const double* a, int lda, const double* b, int ldb,
                                            d4 acc) {
  const int lane = threadIdx.x & 63;
  const int r16 = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int e = 0; e < SB; e += 4) {
    const double av = a[(e + kq) * lda + r16];
    const double bv = b[(e + kq) * ldb + r16];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
  }
  return acc;
}

struct LeafArgs {
  double* A;        // diagonal block (0,0), column-major
  int64_t lda;
  int32_t nvalid;   // columns < nvalid are real, the rest identity padding
  double* invL;     // 128 x 128 column-major out: inverse of the (identity-padded) factor
  double* logdet;   // += sum_{c<nvalid} log L_cc
  int32_t* info;    // set to row0 + c + 1 of the first non-positive pivot (0 = ok)
  int64_t row0;     // global index of the block's first row (for info)
  double* dbg;      // optional: 8 wall-clock stamps (100 MHz) at the phase boundaries
};

#define LEAF_STAMP(i) do { if (g.dbg && threadIdx.x == 0) g.dbg[i] = (double)wall_clock64(); } while (0)

// sqrt(p) and 1/sqrt(p) from the hardware rsq estimate + two Newton steps and one residual
// correction (the dependent chain is ~10 FMAs instead of a full sqrt followed by a division)
__device__ __forceinline__ void sqrt_rsqrt(double p, double& l, double& rl) {
  double y = __builtin_amdgcn_rsq(p);
  const double h = 0.5 * p;
  y = y * fma(-h * y, y, 1.5);
  y = y * fma(-h * y, y, 1.5);
  double s = p * y;
  s = fma(0.5 * y, fma(-s, s, p), s);
  y = y * fma(-h * y, y, 1.5);
  l = s;
  rl = y;
}

// Wave 0: factor the 16 x 16 diagonal sub-block at (c0, c0) held in S (lower triangle), leave
// L_ss in S, 1/diag in rdiag, X_ss = L_ss^-1 dense in dinv_s (column-major) and its strict lower
// part transposed into the upper triangle of S_ss.  Everything stays in registers; pivots,
// multipliers and the entries of L are moved between lanes with v_readlane.
__device__ __forceinline__ void factor_diag16(double* S, double* dinv_s, double* rdiag, int c0, int nv,
                                              int32_t* info, int64_t row0) {
  const int lane = threadIdx.x & 63;
  const int rr = lane < SB ? lane : SB - 1;
  int lidx[SB];  // 4 * source lane, opaque (see bcast_f64)
#pragma unroll
  for (int c = 0; c < SB; ++c) {
    lidx[c] = c << 2;
    asm volatile("" : "+v"(lidx[c]));
  }
  double d[SB];
#pragma unroll
  for (int c = 0; c < SB; ++c) d[c] = S[(c0 + c) * LP + c0 + rr];
  double rl[SB];
  double x[SB];  // X = L^-1, column `rr` per lane:  x[a] = ((a == j) - sum_{k<a} L[a][k] x[k]) / L[a][a]
  int badcol = -1;
  // The pivot chain is the critical path: keep it free of LDS-crossbar round trips.  Every lane
  // tracks its OWN diagonal entry (dg -= L[r][c]^2 needs no other lane), the next pivot is one
  // v_readlane of dg away, and the multiplier broadcasts / rank-1 updates trail one column behind.
  double dg = 0.0;
#pragma unroll
  for (int c = 0; c < SB; ++c) dg = (rr == c) ? d[c] : dg;
#pragma unroll
  for (int c = 0; c < SB; ++c) {
    double piv;
    {
      union {
        double dd;
        int i[2];
      } u;
      u.dd = dg;
      u.i[0] = __builtin_amdgcn_readlane(u.i[0], c);
      u.i[1] = __builtin_amdgcn_readlane(u.i[1], c);
      piv = u.dd;
    }
    const bool bad = !(piv > 0.0);  // also catches NaN; identical in every lane
    badcol = (bad && badcol < 0) ? c : badcol;
    piv = bad ? 1.0 : piv;
    // row c of L is final here (its entries sit in lane c, registers 0..c-1): start row c of the
    // inverse -- independent of the pivot's sqrt chain, so it fills that latency
    double t = (c == rr) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < c; ++k) t = fma(-bcast_f64(d[k], lidx[c]), x[k], t);
    double l;
    sqrt_rsqrt(piv, l, rl[c]);
    x[c] = t * rl[c];
    d[c] = (rr == c) ? l : d[c] * rl[c];
    dg = fma(-d[c], d[c], dg);
#pragma unroll
    for (int c2 = c + 1; c2 < SB; ++c2) {
      const double m = bcast_f64(d[c], lidx[c2]);  // L[c2][c]
      d[c2] = fma(-d[c], m, d[c2]);
    }
  }
  if (badcol >= 0 && lane == 0 && c0 + badcol < nv) atomicCAS(info, 0, (int)(row0 + c0 + badcol + 1));
#pragma unroll
  for (int c = 0; c < SB; ++c)
    if (lane == c) rdiag[c0 + c] = rl[c];
  if (lane < SB) {
#pragma unroll
    for (int c = 0; c < SB; ++c)
      if (c <= lane) S[(c0 + c) * LP + c0 + lane] = d[c];   // L_ss, row `lane`
#pragma unroll
    for (int a = 0; a < SB; ++a) {
      dinv_s[lane * SB + a] = x[a];                          // X[a][lane], column-major
      if (a > lane) S[(c0 + a) * LP + c0 + lane] = x[a];     // strict lower part, transposed
    }
  }
}

__global__ __launch_bounds__(256) void potrf_leaf_kernel(LeafArgs g) {
  __shared__ double S[LB * LP];        // 133,120 B
  __shared__ double dinv[8][SB * SB];  //  16,384 B  dense X_ii, column-major
  __shared__ double rdiag[LB];         //   1,024 B  1 / L_cc  (= X_cc)
  __shared__ double red[4];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int nv = g.nvalid;
  LEAF_STAMP(0);

  // ---- load: lower triangle of real columns, identity elsewhere, zeros above the diagonal ----
  // (8 independent loads in flight per thread: the block is read once, latency-bound otherwise)
  {
    // all 32 16-byte loads of a thread are issued before the first use (one latency, not 64)
    d2 buf[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int p = tid + 256 * i;            // pair index: column p / 64, rows 2 (p % 64), +1
      const int c = p >> 6, r = (p & 63) * 2;
      buf[i] = *reinterpret_cast<const d2*>(g.A + r + (int64_t)c * g.lda);
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int p = tid + 256 * i;
      const int c = p >> 6, r = (p & 63) * 2;
      d2 v = buf[i];
      if (c >= nv) v = d2{0.0, 0.0};
      if (r < c) v[0] = 0.0;
      if (r + 1 < c) v[1] = 0.0;
      if (c >= nv && r == c) v[0] = 1.0;
      if (c >= nv && r + 1 == c) v[1] = 1.0;
      S[c * LP + r] = v[0];
      S[c * LP + r + 1] = v[1];
    }
  }
  __syncthreads();
  LEAF_STAMP(1);

  // ---- factorisation: right-looking over 16-column sub-panels with one-step look-ahead -------
  if (wave == 0) factor_diag16(S, dinv[0], rdiag, 0, nv, g.info, g.row0);
  __syncthreads();
  for (int s = 0; s < LB / SB - 1; ++s) {
    const int c0 = s * SB;
    // (B) sub-panel rows below the diagonal sub-block:  P <- P X_ss^T  (MFMA, in place per tile)
    for (int tr = s + 1 + wave; tr < LB / SB; tr += 4) {
      const int rw = tr * SB;
      d4 acc = d4{0.0, 0.0, 0.0, 0.0};
      acc = mfma_tile_k16(dinv[s], SB, &S[c0 * LP + rw], LP, acc);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) S[(c0 + kq + 4 * r) * LP + rw + r16] = acc[r];
    }
    LEAF_STAMP(8 + 3 * s);
    __syncthreads();
    LEAF_STAMP(9 + 3 * s);
    // (C) rank-16 update of the remaining lower triangle.  Wave 0 takes the next diagonal tile
    //     first and factors it straight away, hiding that dependent chain behind the other
    //     waves' MFMA tiles.
    {
      const int n = LB / SB - 1 - s;  // remaining block rows
      if (wave == 0) {
        const int cc = (s + 1) * SB;
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
        acc = mfma_tile_k16(&S[c0 * LP + cc], LP, &S[c0 * LP + cc], LP, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(cc + kq + 4 * r) * LP + cc + r16] -= acc[r];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        factor_diag16(S, dinv[s + 1], rdiag, cc, nv, g.info, g.row0);
      } else {
        const int ntile = n * (n + 1) / 2;
        for (int t = wave; t < ntile; t += 3) {  // t = 0 is the diagonal tile wave 0 owns
          int tc = 0, rem = t;
          while (rem >= n - tc) {
            rem -= n - tc;
            ++tc;
          }
          const int tr = tc + rem;
          const int cc = (s + 1 + tc) * SB, rw = (s + 1 + tr) * SB;
          d4 acc = d4{0.0, 0.0, 0.0, 0.0};
          acc = mfma_tile_k16(&S[c0 * LP + cc], LP, &S[c0 * LP + rw], LP, acc);
#pragma unroll
          for (int r = 0; r < 4; ++r) S[(cc + kq + 4 * r) * LP + rw + r16] -= acc[r];
        }
      }
    }
    LEAF_STAMP(10 + 3 * s);
    __syncthreads();
  }

  LEAF_STAMP(2);
  // ---- write the factor back (real columns only), accumulate log-det ------------------------
#pragma unroll 8
  for (int idx = tid; idx < LB * LB; idx += 256) {
    const int c = idx >> 7, r = idx & 127;
    if (c < nv && r >= c) g.A[r + (int64_t)c * g.lda] = S[c * LP + r];
  }
  {
    double lg = 0.0;
    if (tid < nv) lg = log(S[tid * LP + tid]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lg += __shfl_down(lg, off);
    if (lane == 0) red[wave] = lg;
    __syncthreads();
    if (tid == 0 && g.logdet) atomicAdd(g.logdet, red[0] + red[1] + red[2] + red[3]);
  }
  if (g.invL == nullptr) return;
  LEAF_STAMP(3);

  // ---- inversion ----------------------------------------------------------------------------
  if (nv < LB) {
    // Only the last diagonal block of a matrix is ragged.  Its panel rows (the y row, padding)
    // are not part of the triangular factor and the padding is identity: rebuild the affected
    // state the slow way.
    __syncthreads();
    for (int idx = tid; idx < LB * LB; idx += 256) {
      const int c = idx >> 7, r = idx & 127;
      if (c < nv && r >= nv) S[c * LP + r] = 0.0;
      if (c >= nv && r >= c) S[c * LP + r] = (r == c) ? 1.0 : 0.0;
      if (r < c) S[c * LP + r] = 0.0;  // drop the X_ss transposes stored during factorisation
    }
    if (tid >= nv && tid < LB) rdiag[tid] = 1.0;
    __syncthreads();
    if (tid < LB) {
      const int s = tid >> 4, j = tid & 15, c0 = s * SB;
      double xc[SB];
#pragma unroll
      for (int a = 0; a < SB; ++a) {
        double t = (a == j) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < a; ++k) t -= S[(c0 + k) * LP + c0 + a] * xc[k];
        xc[a] = t * rdiag[c0 + a];
      }
#pragma unroll
      for (int a = 0; a < SB; ++a) dinv[s][j * SB + a] = xc[a];
    }
    __syncthreads();
    if (tid < LB) {
      const int s = tid >> 4, j = tid & 15, c0 = s * SB;
#pragma unroll
      for (int a = 0; a < SB; ++a)
        if (a > j) S[(c0 + a) * LP + c0 + j] = dinv[s][j * SB + a];
    }
  }
  __syncthreads();
  for (int dist = 1; dist < LB / SB; ++dist) {
    const int nblk = LB / SB - dist;
    for (int bj = wave; bj < nblk; bj += 4) {
      const int bi = bj + dist;
      d4 acc = d4{0.0, 0.0, 0.0, 0.0};
      // T[c][b] = sum_{k=bj..bi-1} sum_e L[16bi+c][16k+e] * X[16k+e][16bj+b]
      for (int k = bj; k < bi; ++k) {
        const double* ap = &S[(k * SB) * LP + bi * SB];  // L_ik, k-major
        const double* bp = &S[(k * SB) * LP + bj * SB];  // X_kj stored at S[col=16k+e][row=16bj+b]
#pragma unroll
        for (int e = 0; e < SB; e += 4) {
          const int ee = e + kq;
          const double av = ap[ee * LP + r16];
          double bv = bp[ee * LP + r16];
          if (k == bj) bv = (ee > r16) ? bv : ((ee == r16) ? rdiag[bj * SB + r16] : 0.0);
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
      }
      double* dst = &S[(bi * SB) * LP + bj * SB];  // destination block, also T staging
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(kq + 4 * r) * LP + r16] = acc[r];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      d4 acc2 = d4{0.0, 0.0, 0.0, 0.0};
      acc2 = mfma_tile_k16(dinv[bi], SB, dst, LP, acc2);  // X_ii * T
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(kq + 4 * r) * LP + r16] = -acc2[r];
    }
    __syncthreads();
  }
  LEAF_STAMP(4);
  // X[p][q] (p > q) sits at S[p*LP + q]; diagonal in rdiag
#pragma unroll 8
  for (int idx = tid; idx < LB * LB; idx += 256) {
    const int q = idx >> 7, p = idx & 127;
    double v = 0.0;
    if (p == q) v = rdiag[p];
    else if (p > q) v = S[p * LP + q];
    g.invL[p + q * LB] = v;
  }
  LEAF_STAMP(5);
}

// Plain reference leaf (one column at a time, no MFMA) -- selected with GMB_LEAF_NAIVE=1 to
// cross-check the blocked leaf on hardware.  Same contract as potrf_leaf_kernel.
__global__ __launch_bounds__(256) void potrf_leaf_naive_kernel(LeafArgs g) {
  __shared__ double S[LB * (LB + 1)];
  __shared__ double X[LB * 16];  // inverse computed in 8 column strips of 16
  const int tid = threadIdx.x;
  const int nv = g.nvalid;
  const int P = LB + 1;
  for (int idx = tid; idx < LB * LB; idx += 256) {
    const int c = idx >> 7, r = idx & 127;
    double v = (r == c) ? 1.0 : 0.0;
    if (c < nv && r >= c) v = g.A[r + (int64_t)c * g.lda];
    S[c * P + r] = v;
  }
  __syncthreads();
  for (int c = 0; c < LB; ++c) {
    double piv = S[c * P + c];
    if (!(piv > 0.0)) {
      if (tid == 0 && c < nv) atomicCAS(g.info, 0, (int)(g.row0 + c + 1));
      piv = 1.0;
    }
    const double l = sqrt(piv);
    __syncthreads();
    if (tid < LB) {
      if (tid == c) S[c * P + c] = l;
      else if (tid > c) S[c * P + tid] /= l;
    }
    __syncthreads();
    for (int idx = tid; idx < LB * LB; idx += 256) {
      const int c2 = idx >> 7, r = idx & 127;
      if (c2 > c && r >= c2) S[c2 * P + r] -= S[c * P + r] * S[c * P + c2];
    }
    __syncthreads();
  }
  for (int idx = tid; idx < LB * LB; idx += 256) {
    const int c = idx >> 7, r = idx & 127;
    if (c < nv && r >= c) g.A[r + (int64_t)c * g.lda] = S[c * P + r];
  }
  if (tid == 0 && g.logdet) {
    double lg = 0.0;
    for (int c = 0; c < nv; ++c) lg += log(S[c * P + c]);
    atomicAdd(g.logdet, lg);
  }
  if (g.invL == nullptr) return;
  __syncthreads();
  if (nv < LB) {
    for (int idx = tid; idx < LB * LB; idx += 256) {
      const int c = idx >> 7, r = idx & 127;
      if (c < nv && r >= nv) S[c * P + r] = 0.0;
      if (c >= nv && r >= c) S[c * P + r] = (r == c) ? 1.0 : 0.0;
    }
  }
  __syncthreads();
  for (int strip = 0; strip < 8; ++strip) {
    if (tid < 16) {
      const int q = strip * 16 + tid;
      for (int p = 0; p < LB; ++p) {
        double t = (p == q) ? 1.0 : 0.0;
        for (int k = q; k < p; ++k) t -= S[k * P + p] * X[k * 16 + tid];
        X[p * 16 + tid] = (p < q) ? 0.0 : t / S[p * P + p];
      }
    }
    __syncthreads();
    for (int idx = tid; idx < LB * 16; idx += 256) {
      const int p = idx >> 4, qq = idx & 15;
      g.invL[p + (strip * 16 + qq) * LB] = X[p * 16 + qq];
    }
    __syncthreads();
  }
}

}  // namespace gmb
