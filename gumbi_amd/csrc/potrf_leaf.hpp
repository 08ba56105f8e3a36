// potrf_leaf.hpp -- diagonal-block leaf of the blocked Cholesky: factor one 128 x 128 block in
// LDS and produce the inverses of its eight 16 x 16 diagonal sub-blocks, the only "inverse" the
// strip solves (trsm_strip.hpp) need; plus the batched inversion of whole diagonal blocks that
// the hyper-parameter gradient uses as the leaves of L^-1.
//
// Replaces the unblocked dpotf2 + dtrtri work LAPACK does inside dpotrf / dtrsm for
// pm.gp.Marginal (call sites gumbi/regression/pymc/GP.py:580, 845-847).
//
// potrf_leaf_kernel: one workgroup of 256 threads (4 waves) owns the block.  Its lower block
// triangle lives column-major in LDS in a packed layout (block column s keeps rows 16 s .. 127,
// 76 KB).  Factorisation is right-looking over 16-column sub-panels:
//   1. wave 0 factors the 16 x 16 diagonal sub-block on the matrix pipe (factor_diag16_mfma: the
//      symmetric block sits in one accumulator quad that doubles as a 16 x 4 panel operand; per
//      4-column step ten v_readlane, a 4 x 4 factor + inverse in uniform arithmetic, four lane
//      gathers and one rank-4 MFMA) and forms its inverse by forward substitution on the identity;
//   2. all waves solve the sub-panel rows, P <- P X_ss^T, one 16 x 16 MFMA tile at a time;
//   3. all waves apply the rank-16 update (MFMA tiles); wave 0 takes the next diagonal tile first
//      and factors it straight away (one-step look-ahead), the others also write the finished
//      block column back to global memory.
// Rows >= nvalid of the block are "panel rows" (the appended y row that carries v = L^-1 y, and
// zero padding): they are solved but never used as pivots; columns >= nvalid are treated as
// identity and never written back.
//
// leaf_invert_kernel: X = L^-1 by 16 x 16 blocks, block-diagonal by block-diagonal,
//   X_ij = -X_ii * (sum_{k=j..i-1} L_ik X_kj), both products on MFMA; X is kept transposed in the
// (free) upper triangle of the full LDS square while it is built.  One workgroup per diagonal
// block, all blocks in one launch, off the factorisation's critical path.
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
  double* dinv16;   // 8 x (16 x 16) column-major out: inverses of the diagonal 16 x 16 sub-blocks of the
                    // identity-padded factor (what trsm_strip_kernel and leaf_invert_kernel consume)
  double* logdet;   // += sum_{c<nvalid} log L_cc
  int32_t* info;    // set to row0 + c + 1 of the first non-positive pivot (0 = ok)
  int64_t row0;     // global index of the block's first row (for info)
  double* dbg;      // optional: 8 wall-clock stamps (100 MHz) at the phase boundaries
                    // (software XCD partition probe, GMB_PROBE_XCD)
  uint32_t* prog;   // optional (persistent tile Cholesky, full blocks only): progress word another workgroup polls --
                    // value k + 1 = block columns 0 .. k-1 of the factored block and the sub-block inverses 0 .. k
                    // are in memory (stored write-through).  The strip solve of the tile below starts its step k then
                    // instead of waiting for the whole block.
};

#define LEAF_STAMP(i) do { if (g.dbg && threadIdx.x == 0) g.dbg[i] = (double)wall_clock64(); } while (0)

// sqrt(p) and 1/sqrt(p) from the hardware rsq estimate (~26 bits) by one coupled (Goldschmidt)
// step on g ~ sqrt(p), h ~ 1/(2 sqrt(p)) and one residual correction of each.  The pivot chain of
// the factorisation waits for 1/sqrt(p): five dependent FMAs after v_rsq_f64 (each costs ~20
// cycles of latency on this chain; three plain Newton steps were nine).
__device__ __forceinline__ void sqrt_rsqrt(double p, double& l, double& rl) {
  const double y = __builtin_amdgcn_rsq(p);
  double g = p * y;
  double h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  const double d = fma(-g, g, p);   // residual of the square root
  const double r2 = fma(-h, g, 0.5);  // residual of the half reciprocal
  const double hh = h + h;
  l = fma(d, h, g);
  rl = fma(hh, r2, hh);
}

// ---------------------------------------------------------------------------------------------
// 16 x 16 diagonal sub-block on the matrix pipe (wave 0).
//
// The block D is held, fully symmetric, in ONE accumulator quad in v_mfma_f64_16x16x4_f64's C/D
// layout (lane: n = lane & 15, rows m = (lane >> 4) + 4 r in acc[r]).  Because D is symmetric,
// acc[q] of lane (k = lane >> 4, i = lane & 15) is D[i][4q + k] -- exactly the A/B operand layout
// of a 16 x 4 column panel.  So each 4-column step is: read the 4 x 4 diagonal piece (10 uniform
// values, v_readlane), factor + invert it in uniform arithmetic, form the solved panel
// P = D[:, 4q..4q+3] X44^T with 4 lane gathers, and apply the rank-4 update D -= P P^T with ONE
// MFMA whose A and B operands are the same register.  The inverse of the 16 x 16 factor is built
// the same way (forward substitution on the identity, one MFMA per 4-row block).  ~600 issued
// instructions instead of ~3000 for the row-per-lane formulation, and the pivot chain carries no
// LDS-crossbar round trips.
__device__ __forceinline__ double rl_sgpr(double v, int lane_const) {
  union {
    double d;
    int i[2];
  } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane_const);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane_const);
  return u.d;
}

// Global store of a result another workgroup of the SAME launch will read (persistent tile Cholesky, chol_tiles.hpp):
// WT = write-through (sc1) -- the bytes leave this XCD's L2 at once, so the publisher needs no agent-scope release fence
// (a fence writes back every dirty line of the L2: ~8 us behind a freshly written 128 KB tile, 3 us with write-through
// stores); plain otherwise.
template <bool WT>
__device__ __forceinline__ void leaf_store(double* p, double v) {
  if constexpr (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// `Sd` / `pitch`: the block column that holds the sub-block (its first row is the sub-block's
// first row, see the packed layout below).
// `gdinv` (may be null): global copy of the dense inverse, for the solves of later kernels.
#define D16_STAMP(i) do { if (dbg && threadIdx.x == 0) dbg[i] = (double)__builtin_readcyclecounter(); } while (0)
template <bool WT = false>
__device__ __forceinline__ void factor_diag16_mfma(double* Sd, int pitch, double* dinv_s, double* gdinv,
                                                   double* rdiag, int c0, int nv, int32_t* info, int64_t row0,
                                                   double* dbg = nullptr) {
  D16_STAMP(0);
  const int lane = threadIdx.x & 63;
  const int r16 = lane & 15, kq = lane >> 4;
  // symmetric load of D
  d4 acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = kq + 4 * r;
    const int hi = m > r16 ? m : r16, lo = m > r16 ? r16 : m;
    acc[r] = Sd[lo * pitch + hi];
  }
  double Lcol[4], ax[4];
  // position masks of this lane: mx[] over the lower triangle (i, k <= i) of a 4 x 4 piece held at
  // lanes (r16 = i, kq = k); keep[q] over the rows / columns that take part in step q
  double mx[10], keep[4];
  {
    int n = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k <= i; ++k) mx[n++] = (r16 == i && kq == k) ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int a = r16 - 4 * q;
      keep[q] = (a < 0 || (a < 4 && kq > a)) ? 0.0 : 1.0;
    }
  }
  unsigned badbits = 0;  // bit c: pivot c was not positive (one OR per pivot; the column is decoded at the end)
  // X = L^-1 (16 x 16) by forward substitution on the identity, 4 rows per step, INTERLEAVED with the
  // factor steps: step q of the substitution needs only X44_q and the solved panel of factor step q,
  // and its two MFMAs run on the matrix pipe while the vector pipe works through the pivot chain of
  // factor step q + 1 (as a separate loop after the factorisation it cost 1460 cycles of 8400).
  d4 e;
#pragma unroll
  for (int r = 0; r < 4; ++r) e[r] = (kq + 4 * r == r16) ? 1.0 : 0.0;
  double Xrow[4];
  D16_STAMP(1);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int j0 = 4 * q;
    // 4 x 4 diagonal piece: T[i][j] sits in lane 16 i + j0 + j, register acc[q]
    const double t00 = rl_sgpr(acc[q], 0 * 16 + j0 + 0);
    const double t10 = rl_sgpr(acc[q], 1 * 16 + j0 + 0), t11 = rl_sgpr(acc[q], 1 * 16 + j0 + 1);
    const double t20 = rl_sgpr(acc[q], 2 * 16 + j0 + 0), t21 = rl_sgpr(acc[q], 2 * 16 + j0 + 1),
                 t22 = rl_sgpr(acc[q], 2 * 16 + j0 + 2);
    const double t30 = rl_sgpr(acc[q], 3 * 16 + j0 + 0), t31 = rl_sgpr(acc[q], 3 * 16 + j0 + 1),
                 t32 = rl_sgpr(acc[q], 3 * 16 + j0 + 2), t33 = rl_sgpr(acc[q], 3 * 16 + j0 + 3);
    double l00, l11, l22, l33, r0, r1, r2, r3;
    double p = t00;
    {
      const bool bad = !(p > 0.0);
      badbits |= bad ? (1u << (j0 + 0)) : 0u;
      p = bad ? 1.0 : p;
    }
    sqrt_rsqrt(p, l00, r0);
    const double l10 = t10 * r0, l20 = t20 * r0, l30 = t30 * r0;
    p = fma(-l10, l10, t11);
    {
      const bool bad = !(p > 0.0);
      badbits |= bad ? (1u << (j0 + 1)) : 0u;
      p = bad ? 1.0 : p;
    }
    sqrt_rsqrt(p, l11, r1);
    const double l21 = fma(-l20, l10, t21) * r1, l31 = fma(-l30, l10, t31) * r1;
    p = fma(-l21, l21, fma(-l20, l20, t22));
    {
      const bool bad = !(p > 0.0);
      badbits |= bad ? (1u << (j0 + 2)) : 0u;
      p = bad ? 1.0 : p;
    }
    sqrt_rsqrt(p, l22, r2);
    const double l32 = fma(-l31, l21, fma(-l30, l20, t32)) * r2;
    p = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, t33)));
    {
      const bool bad = !(p > 0.0);
      badbits |= bad ? (1u << (j0 + 3)) : 0u;
      p = bad ? 1.0 : p;
    }
    sqrt_rsqrt(p, l33, r3);
    // X44 = L44^-1
    const double x10 = -(l10 * r0) * r1;
    const double x21 = -(l21 * r1) * r2;
    const double x32 = -(l32 * r2) * r3;
    const double x20 = -fma(l21, x10, l20 * r0) * r2;
    const double x31 = -fma(l32, x21, l31 * r1) * r3;
    const double x30 = -fma(l32, x20, fma(l31, x10, l30 * r0)) * r3;
    // X44 as an MFMA A operand: lane (i = r16, k = kq) holds X44[r16][kq] for r16 < 4, else 0 -- a sum of
    // ten products with the lane's 0/1 position masks (one of them is 1 at most: exact, and branch-free;
    // nested selects compiled to divergent branches that cost a third of the step)
    ax[q] = fma(mx[9], r3, fma(mx[8], x32, fma(mx[7], x31, fma(mx[6], x30, mx[5] * r2)))) +
            fma(mx[4], x21, fma(mx[3], x20, fma(mx[2], r1, fma(mx[1], x10, mx[0] * r0))));
    // solved panel in operand layout, P[i][k] = sum_k' D[i][j0+k'] X44[k][k'] (i = r16, k = kq), from ONE
    // MFMA: C[m][n] = sum_e X44[m][e] D[n][j0+e]; acc[q] already is the B operand (D is symmetric),
    // and C's register 0 (m = kq, n = r16) is P in the operand layout the rank-4 update needs.
    const d4 pt = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[q], acc[q], d4{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
    // finished rows take no part in the update and the piece's own rows are L44, whose entries above
    // the diagonal are exact zeros (keep[q] = 0 there, 1 elsewhere); L44 itself is the MFMA's
    // T X44^T = L44 to rounding
    const double pn = pt[0] * keep[q];
    Lcol[q] = pn;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-pn, pn, acc, 0, 0, 0);
    {
      // X[4q + kq][r16] = sum_k' X44_q[kq][k'] E[4q + k'][r16]: e[q] is the B operand as it stands
      const d4 xt = __builtin_amdgcn_mfma_f64_16x16x4f64(ax[q], e[q], d4{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
      const double xq = xt[0];
      Xrow[q] = xq;
      if (q < 3) e = __builtin_amdgcn_mfma_f64_16x16x4f64(-pn, xq, e, 0, 0, 0);
    }
    D16_STAMP(2 + q);
  }
  if (badbits) {
    const int badcol = __builtin_ctz(badbits);
    if (lane == 0 && c0 + badcol < nv) atomicCAS(info, 0, (int)(row0 + c0 + badcol + 1));
  }
  D16_STAMP(6); D16_STAMP(7); D16_STAMP(8); D16_STAMP(9);
  // results: L_ss (lower) into the block, dense X_ss into dinv_s, 1/diag
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int col = 4 * q + kq;          // Lcol[q]: L[r16][col];  Xrow[q]: X[col][r16]
    // unconditional: Lcol is exactly zero above the diagonal (keep[]), and nothing reads the upper
    // triangle of a diagonal sub-block in LDS anyway
    Sd[col * pitch + r16] = Lcol[q];
    dinv_s[r16 * SB + col] = Xrow[q];    // column-major X: X[a = col][j = r16] at j*16 + a
    // 1 / L_cc is the diagonal of X: exactly the pivot's reciprocal (the substitution multiplies it
    // by the identity's 1 and adds exact zeros).  The lanes off the diagonal write to a spare slot
    // instead of being branched around.
    rdiag[r16 == col ? c0 + col : LB] = Xrow[q];
  }
  if (gdinv) {
#pragma unroll
    for (int q = 0; q < 4; ++q) leaf_store<WT>(&gdinv[r16 * SB + 4 * q + kq], Xrow[q]);
  }
  D16_STAMP(10);
}

// Packed LDS layout of the lower block triangle: block column s (columns 16s .. 16s+15) keeps
// rows 16s .. 127 only, column pitch 130 - 16s.  76 KB instead of 133 KB for the full square, which
// (with a single 2 KB buffer for the current sub-block inverse) brings the kernel to 79 KB of LDS:
// it then fits on a compute unit NEXT TO one resident GEMM workgroup (74 KB) and no longer has to
// wait for a compute unit to drain completely while a trailing update occupies the chip (that
// wait was the whole duration of the update).
__host__ __device__ constexpr int pk_off(int s) { return 16 * s * (138 - 8 * s); }
__host__ __device__ constexpr int pk_pitch(int s) { return LP - 16 * s; }
constexpr int PK_SIZE = pk_off(8);  // 9472 doubles
__device__ __forceinline__ int pk(int r, int c) {
  const int s = c >> 4;
  return pk_off(s) + (c & 15) * pk_pitch(s) + r - 16 * s;
}

// NW = wavefronts of the calling workgroup (4: the stand-alone kernel; 8: the whole-CU workgroups of
// panel_chain_kernel -- the extra waves take update tiles and write-back).  Every thread of the
// workgroup must call it (workgroup barriers inside).
// LDS of the leaf, in doubles: S[PK_SIZE] (75,776 B) | dinv[256] (dense X_ss of the current sub-panel, column-major)
// | rdiag[LB + 1] (1 / L_cc = X_cc; [LB] is a write-only spare slot) | red[8]
constexpr int LEAF_LDS_DOUBLES = PK_SIZE + SB * SB + (LB + 1) + 8;  // 9873 doubles = 78,984 B

// The leaf on LDS the caller provides (`lds`: LEAF_LDS_DOUBLES doubles, 16-byte aligned): the stand-alone kernel
// below declares its own; the persistent tile Cholesky (chol_tiles.hpp) hands in the region its GEMM staging uses.
// PRE: the caller has already put the block into S (packed layout: (r, c) at pk(r, c) for r >= (c & ~15), zeros above
// the diagonal inside the diagonal sub-blocks; full blocks only, nvalid == 128) -- the persistent tile Cholesky writes the
// epilogue of the diagonal tile's contraction straight into LDS instead of sending it through global memory.
template <int NW, bool WT = false, bool PRE = false>
__device__ __forceinline__ void potrf_leaf_core(const LeafArgs& g, double* __restrict__ lds) {
  const bool pipe = WT && g.prog != nullptr && g.nvalid == LB;  // publish block column by block column
  constexpr int NTH = 64 * NW;
  double* const S = lds;
  double* const dinv = lds + PK_SIZE;
  double* const rdiag = dinv + SB * SB;
  double* const red = rdiag + (LB + 1);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int nv = g.nvalid;
  // sub-block inverses go straight to global memory as they are produced (a ragged block
  // patches them at the end)
  double* gd = g.dinv16;
  // the block column chain waits for this kernel: its waves go first on a compute unit it shares
  __builtin_amdgcn_s_setprio(3);
  LEAF_STAMP(0);

  // ---- load: lower block triangle of real columns, identity elsewhere, zeros above the diagonal.
  //      Wave 0 fetches only the first 16 x 16 diagonal sub-block and factors it at once; the other
  //      waves bring in the rest meanwhile (the first factorisation, 2.3 us, used to start after the
  //      whole 64 KB burst had landed).
  auto put_pair = [&](int r, int c, d2 v) {
    if (c >= nv) v = d2{0.0, 0.0};
    if (r < c) v[0] = 0.0;
    if (r + 1 < c) v[1] = 0.0;
    if (c >= nv && r == c) v[0] = 1.0;
    if (c >= nv && r + 1 == c) v[1] = 1.0;
    const int a = pk(r, c);
    S[a] = v[0];
    S[a + 1] = v[1];
  };
  if constexpr (PRE) {
    if (wave == 0) factor_diag16_mfma<WT>(&S[0], pk_pitch(0), dinv, gd, rdiag, 0, nv, g.info, g.row0, nullptr);
  } else if (wave == 0) {
    d2 buf[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = lane + 64 * i;  // 16 columns x 8 row pairs
      const int c = p >> 3, r = (p & 7) * 2;
      buf[i] = *reinterpret_cast<const d2*>(g.A + r + (int64_t)c * g.lda);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = lane + 64 * i;
      put_pair((p & 7) * 2, p >> 3, buf[i]);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    factor_diag16_mfma<WT>(&S[0], pk_pitch(0), dinv, gd, rdiag, 0, nv, g.info, g.row0, g.dbg ? g.dbg + 40 : nullptr);
  } else {
    // all 16-byte loads of a thread are issued before the first use (one latency, not 19)
    constexpr int NO = NTH - 64;
    constexpr int NL = (8192 + NO - 1) / NO;
    d2 buf[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int p = tid - 64 + NO * i;        // pair index: column p / 64, rows 2 (p % 64), +1
      const int c = p >> 6, r = (p & 63) * 2;
      if (p < 8192 && r >= (c & ~15) && !(c < SB && r < SB))
        buf[i] = *reinterpret_cast<const d2*>(g.A + r + (int64_t)c * g.lda);
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int p = tid - 64 + NO * i;
      const int c = p >> 6, r = (p & 63) * 2;
      if (p < 8192 && r >= (c & ~15) && !(c < SB && r < SB)) put_pair(r, c, buf[i]);
    }
  }
  __syncthreads();
  LEAF_STAMP(1);
  if (pipe && wave == 0) {  // the first sub-block inverse is out (wave 0 stored it above)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(g.prog, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // finished block column s goes back to global memory while the factorisation continues
  auto write_back = [&](int s, int first, int nthreads) {
    const int c0 = s * SB, h = LB - c0;
    const double* Sd = &S[pk_off(s)];
    const int pitch = pk_pitch(s);
    for (int idx = first; idx < SB * h; idx += nthreads) {
      const int cl = idx / h, rl = idx - cl * h;
      if (c0 + cl < nv && rl >= cl) leaf_store<WT>(&g.A[c0 + rl + (int64_t)(c0 + cl) * g.lda], Sd[cl * pitch + rl]);
    }
  };

  // ---- factorisation: right-looking over 16-column sub-panels with one-step look-ahead -------
  //      (sub-block 0 was factored by wave 0 during the load)
  // A ragged block (the last diagonal block of a matrix; never published piecewise) is the identity from column nv on: step s
  // is needed while sub-panel s holds real columns (its panel rows -- the y row, whatever else the caller put below -- are
  // solved against it); the sub-panels behind are identity and take no step.  (N = 392, eight real columns: one step, not seven.)
  const int nsteps = (nv + SB - 1) / SB < LB / SB - 1 ? (nv + SB - 1) / SB : LB / SB - 1;
  for (int s = 0; s < nsteps; ++s) {
    const int c0 = s * SB;
    double* Sd = &S[pk_off(s)];      // block column s, first row c0
    const int pitch = pk_pitch(s);
    // (B) sub-panel rows below the diagonal sub-block:  P <- P X_ss^T  (MFMA, in place per tile)
    for (int tr = s + 1 + wave; tr < LB / SB; tr += NW) {
      const int rw = tr * SB - c0;
      d4 acc = d4{0.0, 0.0, 0.0, 0.0};
      acc = mfma_tile_k16(dinv, SB, &Sd[rw], pitch, acc);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) Sd[(kq + 4 * r) * pitch + rw + r16] = acc[r];
    }
    LEAF_STAMP(8 + 3 * s);
    __syncthreads();
    LEAF_STAMP(9 + 3 * s);
    // (C) rank-16 update of the remaining lower triangle.  Wave 0 takes the next diagonal tile
    //     first and factors it straight away, hiding that dependent chain behind the other
    //     waves' MFMA tiles (and their write-back of the finished block column).
    {
      const int n = LB / SB - 1 - s;  // remaining block rows
      if (wave == 0) {
        const int cc = SB;            // offset of block row s + 1 inside block column s
        double* Sn = &S[pk_off(s + 1)];
        const int pn = pk_pitch(s + 1);
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
        acc = mfma_tile_k16(&Sd[cc], pitch, &Sd[cc], pitch, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) Sn[(kq + 4 * r) * pn + r16] -= acc[r];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // (a ragged block's sub-blocks without a real column are the identity as far as anybody else is concerned: not stored)
        factor_diag16_mfma<WT>(Sn, pn, dinv, gd && c0 + SB < nv ? gd + (s + 1) * SB * SB : nullptr, rdiag, c0 + SB, nv, g.info, g.row0);
      } else {
        // published column by column: block column s (final since the barrier above) goes out FIRST, so that its
        // write-through stores land under the update tiles below and the drain before the next barrier is free
        if (pipe) write_back(s, tid - 64, NTH - 64);
        const int ntile = n * (n + 1) / 2;
        // Tiles 1 .. ntile-1 (t = 0 is the diagonal tile wave 0 owns) are dealt in rounds of 2 (NW - 2) + 1
        // slots: two per wave, but only ONE for wave NW / 2 -- it shares its SIMD with wave 0, whose
        // dependent chain takes half of that SIMD's issue slots, so it works at about half speed (with an
        // equal share it reached the barrier ~1 us after everybody else in the first two steps).
        constexpr int SLOTS = 2 * (NW - 2) + 1;
        const int mate = NW / 2;
        const int rank = wave < mate ? wave - 1 : wave - 2;  // 0 .. NW-3 among the full-speed waves
        for (int i = 0;; ++i) {
          const int t = (wave == mate) ? 1 + i * SLOTS + (SLOTS - 1) : 1 + (i >> 1) * SLOTS + rank + (i & 1) * (NW - 2);
          if (t >= ntile) break;
          int tc = 0, rem = t;
          while (rem >= n - tc) {
            rem -= n - tc;
            ++tc;
          }
          const int tr = tc + rem;
          const int sc = s + 1 + tc;                         // destination block column
          const int cc = (1 + tc) * SB, rw = (1 + tr) * SB;  // offsets inside block column s
          d4 acc = d4{0.0, 0.0, 0.0, 0.0};
          acc = mfma_tile_k16(&Sd[cc], pitch, &Sd[rw], pitch, acc);
          double* dst = &S[pk_off(sc)] + (tr - tc) * SB + r16;
          const int pd = pk_pitch(sc);
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[(kq + 4 * r) * pd] -= acc[r];
        }
        // write-back is deferred to the steps where these waves run out of update tiles -- unless the block is
        // published column by column: then block column s goes out now and must have landed before the barrier
        if (!pipe && s >= 3) write_back(s - 3, tid - 64, NTH - 64);
      }
      if (pipe) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (wave 0: the next sub-block inverse)
    }
    LEAF_STAMP(10 + 3 * s);
    __syncthreads();
    if (pipe && wave == 0) __hip_atomic_store(g.prog, (uint32_t)(s + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  LEAF_STAMP(2);
  // ---- remaining block columns back, accumulate log-det --------------------------------------
  for (int s = pipe ? LB / SB - 1 : (nsteps >= 3 ? nsteps - 3 : 0); s < LB / SB; ++s) write_back(s, tid, NTH);
  {
    double lg = 0.0;
    if (tid < nv) lg = -log(rdiag[tid]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lg += __shfl_down(lg, off);
    if (lane == 0) red[wave] = lg;
    __syncthreads();
    if (tid == 0 && g.logdet) {
      double tot = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) tot += red[w];
      atomicAdd(g.logdet, tot);
    }
  }
  LEAF_STAMP(3);
  if (g.dinv16 && nv < LB) {
    // Only the last diagonal block of a matrix is ragged.  What the solves want are the sub-block inverses of the
    // IDENTITY-PADDED factor: its panel rows (the y row, padding) are not part of the triangle, while the factorisation above
    // treated them as rows of L (and let its updates run over the padding's diagonal -- discarded, never written back).  Of
    // the sub-blocks with real columns only the ONE that also holds panel rows differs: it came out as
    // inv([L_mm 0; P C]) = [L_mm^-1 0; * C^-1] where the padded factor has [L_mm 0; 0 I] -- its rows from m on are reset to
    // the identity's; the sub-blocks without a real column are the identity.  (Until round 4 all eight were rebuilt by a
    // 16-step substitution per thread: ~10 us on every ragged matrix.)
    const int ss = nv / SB, m = nv - ss * SB;
    const int first_id = m > 0 ? ss + 1 : ss;  // sub-blocks from here on hold no real column
    for (int idx = tid; idx < (LB / SB - first_id) * SB * SB; idx += NTH) {
      const int e = idx % (SB * SB);
      leaf_store<WT>(&g.dinv16[first_id * SB * SB + idx], (e / SB == e % SB) ? 1.0 : 0.0);
    }
    if (m > 0 && wave == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's own stores of that sub-block (factor_diag16_mfma) first
      for (int idx = lane; idx < SB * SB; idx += 64) {
        const int j = idx / SB, a = idx % SB;  // column-major X: X[a][j] at j * 16 + a
        if (a >= m) leaf_store<WT>(&g.dinv16[ss * SB * SB + idx], a == j ? 1.0 : 0.0);
      }
    }
  }
  LEAF_STAMP(4);
  __syncthreads();
}

template <int NW>
__device__ __forceinline__ void potrf_leaf_body(const LeafArgs& g) {
  __shared__ __attribute__((aligned(16))) double lds[LEAF_LDS_DOUBLES];
  potrf_leaf_core<NW>(g, lds);
}

// 8 wavefronts: wave 0 carries the dependent diagonal chain, the other seven share the rank-16
// update tiles and the write-back (3.5 us faster per block than with four).
__global__ __launch_bounds__(512) void potrf_leaf_kernel(LeafArgs g) {
  potrf_leaf_body<8>(g);
}

// ---- inverse of a factored diagonal block (off the factorisation's critical path) ----------
// One workgroup per diagonal block: X = inv(L_kk) (identity-padded) from L_kk and its eight
// 16 x 16 sub-block inverses, block diagonal by block diagonal on the matrix pipe:
//   X_ij = -X_ii (sum_{j<=k<i} L_ik X_kj).
// Only the hyper-parameter gradient needs these (W = inv(L) is assembled from them).
struct InvArgs {
  const double* L;       // first diagonal block of the factor (column-major, leading dimension lda)
  int64_t lda;
  int64_t blk_stride;    // elements between consecutive diagonal blocks (128 * (lda + 1) in the factor buffer)
  const double* dinv16;  // nblk x 8 x 256
  double* invL;          // optional: nblk x 128 x 128 out, column-major, identity-padded
  int64_t n;             // real rows of the whole matrix from the first block on (ragged last block)
  double* W;             // optional: X_b into diagonal block b of a lower-triangular matrix ...
  int64_t ldw;
  double* U;             // ... and X_b^T over diagonal block b of an upper-triangular one (may be the
  int64_t ldu;           // factor buffer itself: a workgroup reads its whole block before it writes)
};

__global__ __launch_bounds__(256) void leaf_invert_kernel(InvArgs g) {
  __shared__ double S[LB * LP];
  __shared__ double dinv[8][SB * SB];
  __shared__ double rdiag[LB];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int b = blockIdx.x;
  const int64_t left = g.n - (int64_t)b * LB;
  const int nv = left < LB ? (int)left : LB;
  const double* A = g.L + (int64_t)b * g.blk_stride;
  const double* dsrc = g.dinv16 + (int64_t)b * 8 * SB * SB;
  {
    d2 buf[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int p = tid + 256 * i;
      const int c = p >> 6, r = (p & 63) * 2;
      buf[i] = *reinterpret_cast<const d2*>(A + r + (int64_t)c * g.lda);
    }
    for (int idx = tid; idx < 8 * SB * SB; idx += 256) dinv[idx >> 8][idx & 255] = dsrc[idx];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int p = tid + 256 * i;
      const int c = p >> 6, r = (p & 63) * 2;
      d2 v = buf[i];
      if (c >= nv || r >= nv) v[0] = 0.0;
      if (c >= nv || r + 1 >= nv) v[1] = 0.0;
      if (r < c) v[0] = 0.0;
      if (r + 1 < c) v[1] = 0.0;
      S[c * LP + r] = v[0];
      S[c * LP + r + 1] = v[1];
    }
  }
  __syncthreads();
  // the sub-block inverses go (transposed) into the empty upper triangles of the diagonal
  // sub-blocks, the layout the sweep below reads X_kj in; the diagonal itself into rdiag
  if (tid < LB) {
    const int s = tid >> 4, j = tid & 15, c0 = s * SB;
    rdiag[tid] = dinv[s][j * SB + j];
#pragma unroll
    for (int a = 0; a < SB; ++a)
      if (a > j) S[(c0 + a) * LP + c0 + j] = dinv[s][j * SB + a];
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
  // X[p][q] (p > q) sits at S[p*LP + q]; diagonal in rdiag
#pragma unroll 8
  for (int idx = tid; idx < LB * LB; idx += 256) {
    const int q = idx >> 7, p = idx & 127;
    double v = 0.0;
    if (p == q) v = rdiag[p];
    else if (p > q) v = S[p * LP + q];
    if (g.invL) g.invL[(int64_t)b * LB * LB + p + q * LB] = v;
    if (g.W) g.W[(int64_t)b * LB * (g.ldw + 1) + p + (int64_t)q * g.ldw] = v;
  }
  if (g.U) {
#pragma unroll 8
    for (int idx = tid; idx < LB * LB; idx += 256) {
      const int q = idx >> 7, p = idx & 127;  // U[p][q] = X[q][p]
      double v = 0.0;
      if (p == q) v = rdiag[p];
      else if (q > p) v = S[q * LP + p];
      g.U[(int64_t)b * LB * (g.ldu + 1) + p + (int64_t)q * g.ldu] = v;
    }
  }
}

// Plain reference leaf (one column at a time, no MFMA) -- selected with GMB_LEAF_NAIVE=1 to
// cross-check the blocked leaf on hardware.  Same contract as potrf_leaf_kernel.
__global__ __launch_bounds__(256) void potrf_leaf_naive_kernel(LeafArgs g) {
  __shared__ double S[LB * (LB + 1)];
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
  if (g.dinv16 == nullptr) return;
  __syncthreads();
  if (nv < LB) {
    for (int idx = tid; idx < LB * LB; idx += 256) {
      const int c = idx >> 7, r = idx & 127;
      if (c < nv && r >= nv) S[c * P + r] = 0.0;
      if (c >= nv && r >= c) S[c * P + r] = (r == c) ? 1.0 : 0.0;
    }
  }
  __syncthreads();
  if (tid < LB) {  // column j of the inverse of diagonal sub-block s, by forward substitution
    const int s = tid >> 4, j = tid & 15, c0 = s * 16;
    double xc[16];
    for (int a = 0; a < 16; ++a) {
      double t = (a == j) ? 1.0 : 0.0;
      for (int k = j; k < a; ++k) t -= S[(c0 + k) * P + c0 + a] * xc[k];
      xc[a] = (a < j) ? 0.0 : t / S[(c0 + a) * P + c0 + a];
    }
    for (int a = 0; a < 16; ++a) g.dinv16[s * 256 + j * 16 + a] = xc[a];
  }
}

}  // namespace gmb
