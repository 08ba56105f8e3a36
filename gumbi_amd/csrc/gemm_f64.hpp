// gemm_f64.hpp -- the one dense contraction of the GP hot path, on gfx950 f64 MFMA.
//
//   C[n + m*ldc] = beta * C[n + m*ldc] + alpha * sum_{k in [k_lo, k_hi)} A[m + k*lda] * B[n + k*ldb]
//
// with every matrix column-major and both operands "k-major" (consecutive m / n contiguous in
// memory for a fixed k), so every global load is a full-line coalesced read.  Users:
//   * Cholesky trailing update (SYRK/GEMM):  A = B = factored panel, C = trailing block,
//     alpha = -1, beta = 1, tri = 1 (tiles strictly above the diagonal are skipped);
//   * panel / predict triangular solves:     A = inv(L_kk) (128 x 128), C aliases B,
//     alpha = 1, beta = 0;
//   * predict forward-substitution update:   A = L block row, B = solved V columns;
//   * NLML gradient: L^-1 / L^-T by recursive block inversion and Sigma^-1 = L^-T L^-1, with
//     per-tile k ranges that skip the structurally zero part of the triangular operands.
//
// Replaces the LAPACK dpotrf/dtrsm/dpotri calls PyTensor's Cholesky / SolveTriangular ops (and
// their gradients) make under pm.gp.Marginal (call sites gumbi/regression/pymc/GP.py:580, 811,
// 845-847).
//
// Tiling (MI355X, wave64): 256 threads = 4 waves as 2 x 2; a wave owns WTM x WTN
// v_mfma_f64_16x16x4_f64 tiles, so the block tile is (32 WTM) x (32 WTN): 128 x 128 for the
// large updates (16 accumulators x 4 f64 per lane), 64 x 64 / 128 x 64 / 128 x 32 when a launch
// would otherwise leave most of the 256 CUs idle (the bottom of the recursion, the 128-column
// in-place solves).  Operands are staged through LDS in [k][row] order with a row pitch of
// (tile + 16) doubles so that the two k-slices a 32-lane group reads (ds_read_b64, 64 banks x
// 4 B) fall on disjoint banks.  Global loads are 16 B per lane along the contiguous index, one
// k-tile ahead of the MFMAs (register-staged double buffer, one barrier per k-tile).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>

namespace gmb {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int TILE = 128;         // block tile edge (both m and n)
constexpr int KT = 16;            // k extent of one LDS stage
constexpr int PITCH = TILE + 16;  // LDS row pitch in doubles: PITCH % 32 == 16 -> conflict-free

struct GemmArgs {
  double* C;
  int64_t ldc;
  const double* A;
  int64_t lda;
  const double* B;
  int64_t ldb;
  int32_t mt, nt;  // tile counts along m and n (in the launch's own block-tile units)
  int32_t k;       // multiple of KT
  double alpha, beta;
  // tri: skip tile (tm, tn) when it lies strictly above the diagonal, i.e. when its last row
  // noff(tn) + BN - 1 + tri_off is smaller than its first column tm*BM (rows / columns measured
  // from the origins of C's row and column ranges; tri_off shifts the row origin).
  int32_t tri;
  int32_t tri_off;     // elements
  // n-side block stride: the n index walks 128-row blocks that are `nblk_stride` blocks apart in
  // memory (1 = contiguous).  This is how one rank of the block-cyclic row partition updates
  // only the block rows it owns: noff(tn) = ((tn*BN)/128)*nblk_stride*128 + (tn*BN)%128.
  int32_t nblk_stride;
  // block stride of the n index in C when it differs from the one in B (0 = the same): one rank's block
  // rows of Sigma^-1 = U U^T read the strided rows of U out of the full-size factor buffer and write a
  // PACKED result (cblk_stride = 1)
  int32_t cblk_stride;
  // per-tile contraction range: k_lo = klo_m*tm*BM + klo_n*R(tn), k_hi = khi_n ? min(k, R(tn)+BN) : k,
  // R(tn) = noff(tn) + krow_off = the tile's first row in the coordinates of the contraction index
  // (krow_off = 0 and noff = tn*BN for a contiguous n range; a rank of the block-cyclic partition
  // passes its first block row's offset)
  int32_t klo_m, klo_n, khi_n;
  int32_t krow_off;
  // logical block stride of the n range for R(tn) when it differs from the stride in memory (0 =
  // the same): a rank's packed rows of U = L^-T are contiguous in its buffer but stand for every
  // G-th block row of the matrix, and their structural zeros end at that LOGICAL row
  int32_t krow_stride;
  // XCD-balanced schedule (filled by gemm_schedule): the computed tiles, enumerated row-major
  // (tm outer, tn inner), are cut into 8 contiguous runs of equal WORK; block b serves run b % 8.
  int32_t xstart[9];
  // Tile order.  0: the XCD runs above.  1 / 2: block b computes tile b of the list enumerated
  // n-tile by n-tile (1: tn ascending, 2: tn descending; tm inner).  With a triangular OPERAND the
  // contraction length depends on tn only (klo_n: longest at tn = 0, khi_n: longest at the last
  // tn), so these orders dispatch the longest tiles first: a full-k tile of the N = 10k inverse
  // runs 1.1 ms of a 2.8 ms launch and must not start late.  Consecutive blocks land on
  // consecutive XCDs, which deals every length class evenly over the 8 XCDs.
  int32_t order;
  // order == 0 only.  strip > 0: L2-aware rasterisation -- the tile list is enumerated strip by strip of
  // `strip` consecutive m-tiles, inside a strip n-tile by n-tile (tm fastest).  Blocks are dispatched in list
  // order, so the ~64 tiles resident on one XCD (32 compute units x 2 workgroups) form a strip x 64/strip
  // patch that shares `strip` A panels and 64/strip B panels through that XCD's 4 MiB L2, instead of one A
  // panel and 64 different B panels in the row-major order (strip == 0).  PMC at C3 (r02d): the row-major
  // 128 x 128 launches fetched 2.5 TB/s, 13x the algorithmic operand + C traffic.
  int32_t strip;
  // strip > 0 (filled by gemm_schedule): the strip that holds the first tile of XCD run x starts at m-tile
  // xs_t0[x] and at position xs_base[x] of the list -- a block starts its search for its strip there instead of
  // at strip 0 (one rank's update at N = 100k has ~90 strips; the search cost ~60 us of a 1.4 ms tile)
  int32_t xs_t0[8], xs_base[8];
};

// XCD-aware tile order.  The dispatcher places block b on XCD b % 8 (observed, speed only), and
// each XCD has its own L2.  Handing XCD x one CONTIGUOUS run of the row-major tile list keeps
// tiles that share an A panel on one L2; cutting the runs by accumulated work (triangular
// updates skip tiles, triangular operands shorten k ranges) keeps the 8 XCDs equally busy --
// equal tile COUNTS left the last XCD idle for half of a SYRK (23 vs 44 TFLOP/s measured).
__host__ __device__ __forceinline__ int xcd_remap(int bid, int nwg) {  // equal-count variant (K-build)
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// row offset of n-tile tn, and the first n-tile whose rows reach element T (see GemmArgs)
__host__ __device__ __forceinline__ int64_t gemm_noff(int tn, int bn, int stride) {
  const int64_t e = (int64_t)tn * bn;
  return (e >> 7) * stride * 128 + (e & 127);
}
__host__ __device__ __forceinline__ int gemm_first_tn(int64_t T, int bn, int stride) {
  if (T <= 0) return 0;
  const int per = 128 / bn;
  if (T < 0x7fffffff) {  // (every matrix the engine can hold: 32-bit division is several times cheaper on the device)
    const uint32_t S = (uint32_t)stride * 128u, t = (uint32_t)T;
    const uint32_t q = t / S, r = t - q * S;
    if (r == 0) return (int)(q * per);
    if (r > (uint32_t)(128 - bn)) return (int)((q + 1) * per);
    return (int)(q * per + (r + bn - 1) / (uint32_t)bn);
  }
  const int64_t S = (int64_t)stride * 128;
  const int64_t q = T / S, r = T % S;
  if (r == 0) return (int)(q * per);
  if (r > 128 - bn) return (int)((q + 1) * per);
  return (int)(q * per + (r + bn - 1) / bn);
}

// Block index -> tile (tm, tn) of the launch's tile list; false when the block has nothing to do.  Shared by
// the kernels and by the host-side enumeration the tests use (gmb_debug_tile_list): every computed tile must
// come out exactly once over the grid gemm_schedule returns.
__host__ __device__ inline bool gemm_decode_tile(const GemmArgs& g, const int BM, const int BN, const int vbid, int& tm,
                                                 int& tn) {
  const int xcd = vbid & 7;
  int ci = g.xstart[xcd] + (vbid >> 3);
  if (g.order == 0 && g.strip > 0) {
    if (ci >= g.xstart[xcd + 1]) return false;
    // strip-major list: strip s holds m-tiles [s*strip, min(mt, (s+1)*strip)); for n-tile tn it has the
    // m-tiles of the strip whose first computed n-tile is <= tn (first() is non-decreasing in tm)
    auto first_of = [&](int t) {
      if (!g.tri) return 0;
      const int f = gemm_first_tn((int64_t)t * BM - g.tri_off - (BN - 1), BN, g.nblk_stride);
      return f > g.nt ? g.nt : f;
    };
    int t0 = g.xs_t0[xcd];
    ci -= g.xs_base[xcd];
    for (;;) {  // find the strip
      const int t1 = t0 + g.strip < g.mt ? t0 + g.strip : g.mt;
      int cnt = 0;
      for (int t = t0; t < t1; ++t) cnt += g.nt - first_of(t);
      if (ci < cnt) break;
      ci -= cnt;
      t0 = t1;
      if (t0 >= g.mt) return false;
    }
    const int t1 = t0 + g.strip < g.mt ? t0 + g.strip : g.mt, sz = t1 - t0;
    const int f_hi = first_of(t1 - 1);
    tn = first_of(t0);
    int have = 1;  // m-tiles of the strip that reach n-tile tn
    while (tn < f_hi) {  // ragged head (triangular corner of the strip)
      while (have < sz && first_of(t0 + have) <= tn) ++have;
      if (ci < have) break;
      ci -= have;
      ++tn;
    }
    if (tn >= f_hi) {  // rectangular body: every m-tile of the strip
      tn += ci / sz;
      ci -= (ci / sz) * sz;
    }
    tm = t0 + ci;
    return true;
  }
  if (g.order == 0) {
    if (ci >= g.xstart[xcd + 1]) return false;
    if (!g.tri) {
      tm = ci / g.nt;
      tn = ci - tm * g.nt;
      return true;
    }
    tm = 0;
    for (;;) {
      int first = gemm_first_tn((int64_t)tm * BM - g.tri_off - (BN - 1), BN, g.nblk_stride);
      first = first > g.nt ? g.nt : first;
      const int cnt = g.nt - first;
      if (ci < cnt) {
        tn = first + ci;
        return true;
      }
      ci -= cnt;
      ++tm;
    }
  }
  // n-major list (nblk_stride == 1): n-tile tn holds the m-tiles [0, cnt(tn))
  ci = vbid;
  const int step = g.order == 1 ? 1 : -1;
  tn = g.order == 1 ? 0 : g.nt - 1;
  for (;;) {
    if (tn < 0 || tn >= g.nt) return false;
    int cnt = g.mt;
    if (g.tri) {
      const int lim = (tn * BN + BN - 1 + g.tri_off) / BM + 1;
      cnt = lim < cnt ? lim : cnt;
    }
    if (ci < cnt) {
      tm = ci;
      return true;
    }
    ci -= cnt;
    tn += step;
  }
}

// __launch_bounds__(256, 2): two workgroups (= two waves per SIMD) per CU.  The 128 x 128 variant
// then keeps its 128 accumulator registers + staging in 207 VGPRs (no AGPRs, no spills) and the
// second workgroup's MFMAs fill the first one's barrier / staging bubbles: 39 -> 52 TF/s on the
// whole N = 30k Cholesky, 44 -> 59 TF/s on the predict GEMMs (measured A/B on MI355X).
// `vbid`: the block index this call stands for (blockIdx.x in the plain kernels; a fused kernel
// that walks a tile list with fewer workgroups passes its own counter).
template <int WGM, int WGN, int WTM, int WTN, bool PFC = false>
__device__ __forceinline__ void gemm_f64_body(const GemmArgs& g, const int vbid) {
  constexpr int NT = 64 * WGM * WGN;                      // threads: WGM x WGN waves
  constexpr int BM = 16 * WTM * WGM, BN = 16 * WTN * WGN; // block tile; wave tile 16*WTM x 16*WTN
  constexpr int PA = BM + 16, PB = BN + 16;     // LDS pitches, % 32 == 16 -> conflict-free ds_read_b64
  constexpr int LA = BM / 2, RA = NT / LA, NA = KT / RA;  // staging: lanes per k-row, rows per pass, passes
  constexpr int LB_ = BN / 2, RB = NT / LB_, NB = KT / RB;
  static_assert(NA >= 1 && NB >= 1 && NA * RA == KT && NB * RB == KT, "staging must tile the k-tile");
  __shared__ double lds[2][KT * (PA + PB)];

  int tm, tn;
  if (!gemm_decode_tile(g, BM, BN, vbid, tm, tn)) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int r16 = lane & 15, kq = lane >> 4;

  const int krow = (int)gemm_noff(tn, BN, g.krow_stride > 0 ? g.krow_stride : g.nblk_stride) + g.krow_off;
  int k_lo = g.klo_m * tm * BM + g.klo_n * krow;
  int k_hi = g.k;
  if (g.khi_n) k_hi = min(k_hi, krow + BN);
  if (k_lo < 0) k_lo = 0;
  if (k_lo > k_hi) k_lo = k_hi;

  const int a_row = tid / LA, a_col = 2 * (tid % LA);
  const int b_row = tid / LB_, b_col = 2 * (tid % LB_);
  const double* __restrict__ Ag = g.A + (int64_t)tm * BM + a_col;
  const int64_t noff = gemm_noff(tn, BN, g.nblk_stride);
  const double* __restrict__ Bg = g.B + noff + b_col;

  d4 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

  d2 ra[NA], rb[NB];
  const int kt0 = k_lo / KT, kt1 = k_hi / KT;

  // PFC instantiations (small tiles, launches with a contraction of at most 1024: the latency-bound
  // updates of the panel chain): request the C tile BEFORE the contraction, so its memory latency
  // runs under the k loop instead of after it -- N = 10k factorisation 9.92 -> 9.65 ms.  Separate
  // instantiations because the extra registers cost the long launches of the same tile shapes ~1 %.
  // C never aliases A or B when beta != 0.
  constexpr bool PREFETCH_C = PFC && WTM * WTN <= 8;
  double cpre[PREFETCH_C ? WTM * 4 * WTN : 1];
  const bool prefetch_c = PREFETCH_C && g.beta != 0.0;
  if constexpr (PREFETCH_C) {
    if (prefetch_c) {
      const double* __restrict__ Cp = g.C + (g.cblk_stride > 0 ? gemm_noff(tn, BN, g.cblk_stride) : noff) + wn * (16 * WTN) + r16;
      const int64_t mp = (int64_t)tm * BM + wm * (16 * WTM) + kq;
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int j = 0; j < WTN; ++j) cpre[(i * 4 + r) * WTN + j] = Cp[(mp + i * 16 + 4 * r) * g.ldc + j * 16];
    }
  }

  auto gload = [&](int kt) {
#pragma unroll
    for (int p = 0; p < NA; ++p)
      ra[p] = *reinterpret_cast<const d2*>(Ag + ((int64_t)kt * KT + a_row + RA * p) * g.lda);
#pragma unroll
    for (int p = 0; p < NB; ++p)
      rb[p] = *reinterpret_cast<const d2*>(Bg + ((int64_t)kt * KT + b_row + RB * p) * g.ldb);
  };
  auto lstore = [&](int st) {
    double* As = &lds[st][0];
    double* Bs = &lds[st][KT * PA];
#pragma unroll
    for (int p = 0; p < NA; ++p) *reinterpret_cast<d2*>(&As[(a_row + RA * p) * PA + a_col]) = ra[p];
#pragma unroll
    for (int p = 0; p < NB; ++p) *reinterpret_cast<d2*>(&Bs[(b_row + RB * p) * PB + b_col]) = rb[p];
  };

  // one k-tile of MFMAs from LDS stage st
  auto compute = [&](int st) {
    const double* As = &lds[st][0];
    const double* Bs = &lds[st][KT * PA];
#pragma unroll
    for (int k4 = 0; k4 < KT; k4 += 4) {
      double a[WTM], b[WTN];
#pragma unroll
      for (int i = 0; i < WTM; ++i) a[i] = As[(k4 + kq) * PA + wm * (16 * WTM) + i * 16 + r16];
#pragma unroll
      for (int j = 0; j < WTN; ++j) b[j] = Bs[(k4 + kq) * PB + wn * (16 * WTN) + j * 16 + r16];
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };

  if (kt0 < kt1) {
    gload(kt0);
    lstore(0);
    __syncthreads();
    int st = 0;
    // Steady state.  The last k-tile is peeled so that the body is ONE basic block, and the instruction order
    // inside it is pinned with sched_group_barrier: the next tile's global loads go in among the first MFMAs
    // of this tile and its LDS stores among the last ones, one memory instruction per SLOT MFMAs, instead of
    // a cluster of loads at the top and a cluster of ds_writes before the barrier.  Those clusters were
    // where the matrix pipe idled (PMC r02f: pipe busy 88 % with both resident waves of a SIMD issuing
    // non-MFMA work at the k-tile seams); left to itself the compiler hoists every fragment read, then waits
    // for the global loads right after issuing them (63 TF/s).  Measured on one box, 8192^3 / 16384^2 x 3072:
    // 69.1 / 68.5 TF/s before, 73.4 / 73.0 TF/s with this order; N = 30k factorisation 152.3 -> 146.7 ms,
    // gradient 272.7 -> 263.5 ms (round-2 A/B; s_setprio around the MFMAs: -5 %; explicit fragment
    // prefetch: no change; loop rotated by one k4 group so that the last 16 MFMAs of a tile sit behind the
    // barrier and cover the next tile's first LDS reads: -1.5 %).
    constexpr int NMFMA = WTM * WTN * (KT / 4), NMEM = NA + NB;
    constexpr int SLOT = NMFMA / (4 * NMEM) > 0 ? NMFMA / (4 * NMEM) : 1;
    static_assert(2 * SLOT * NMEM <= NMFMA, "not enough MFMAs to interleave the staging with");
    for (int kt = kt0; kt + 1 < kt1; ++kt) {
      gload(kt + 1);
      compute(st);
      lstore(st ^ 1);
#pragma unroll
      for (int q = 0; q < NMEM; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, SLOT, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - 2 * SLOT * NMEM, 0);
#pragma unroll
      for (int q = 0; q < NMEM; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, SLOT, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);     // DS write
      }
      __syncthreads();
      st ^= 1;
    }
    compute(st);
  }


  // epilogue.  v_mfma_f64_16x16x4_f64 D layout: n = lane & 15, m = (lane >> 4) + 4 * reg.
  double* __restrict__ Cg = g.C + (g.cblk_stride > 0 ? gemm_noff(tn, BN, g.cblk_stride) : noff) + wn * (16 * WTN) + r16;
  const int64_t m0 = (int64_t)tm * BM + wm * (16 * WTM) + kq;
  if (g.beta == 0.0) {
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* row = Cg + (m0 + i * 16 + 4 * r) * g.ldc;
#pragma unroll
        for (int j = 0; j < WTN; ++j) row[j * 16] = g.alpha * acc[i][j][r];
      }
  } else {
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* row = Cg + (m0 + i * 16 + 4 * r) * g.ldc;
        double c[WTN];
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
          if constexpr (PREFETCH_C) c[j] = prefetch_c ? cpre[(i * 4 + r) * WTN + j] : row[j * 16];
          else c[j] = row[j * 16];
        }
#pragma unroll
        for (int j = 0; j < WTN; ++j) row[j * 16] = g.beta * c[j] + g.alpha * acc[i][j][r];
      }
  }
}

// ---- the 128 x 128 tile with LDS-DMA staging (round 4, VERDICT r03 item 4: one attempt) ------------------------------------
// The register-staged loop above leaves the matrix pipe idle ~7 % of the time at its k-tile seams even with two workgroups
// per compute unit (eight global loads + eight ds_write_b128 per thread and tile, one barrier at the END of a tile, the first
// fragment reads behind it).  Here the operands go global -> LDS directly (global_load_lds_dwordx4: one wave instruction per
// 128-double k-row = one row of the padded [k][row] image), into a ring of FOUR 8-deep stages in the same 73,728 B:
//     tile kt:  step 0 (16 MFMAs; reads the second half's fragments)
//               s_waitcnt lgkmcnt(0) vmcnt(8): this wave's reads of tile kt are back, its four DMAs of tile kt + 1 have landed
//               s_barrier:                     ... everybody's; the stage of tile kt may be overwritten
//               DMA of tile kt + 4 into it
//               step 1 (16 MFMAs; reads the first fragments of tile kt + 1)
// No staging registers, no ds_writes, the barrier inside the MFMA stream, three tiles in flight across it.  A call past the
// end of the contraction re-reads the last tile (the ring always has the same number of DMAs outstanding; the garbage lands
// in stages nobody reads).
template <bool DUMMY = false>
__device__ __forceinline__ void gemm_f64_dma_body(const GemmArgs& g, const int vbid) {
  constexpr int WGN = 2, WTM = 4, WTN = 4;  // 2 x 2 waves of 4 x 4 MFMA tiles
  constexpr int BM = 128, BN = 128, PA = BM + 16, PB = BN + 16;
  constexpr int KD = 8, NS = 4;             // k-tile depth and stages of the ring
  constexpr int STAGE = KD * (PA + PB);     // doubles
  typedef __attribute__((address_space(3))) double lds_double;
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void g_void;
  __shared__ __attribute__((aligned(16))) double lds_[NS * STAGE];
  lds_double* const lds = (lds_double*)lds_;

  int tm, tn;
  if (!gemm_decode_tile(g, BM, BN, vbid, tm, tn)) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int r16 = lane & 15, kq = lane >> 4;

  const int krow = (int)gemm_noff(tn, BN, g.krow_stride > 0 ? g.krow_stride : g.nblk_stride) + g.krow_off;
  int k_lo = g.klo_m * tm * BM + g.klo_n * krow;
  int k_hi = g.k;
  if (g.khi_n) k_hi = min(k_hi, krow + BN);
  if (k_lo < 0) k_lo = 0;
  if (k_lo > k_hi) k_lo = k_hi;
  const int64_t noff = gemm_noff(tn, BN, g.nblk_stride);
  const int kt0 = k_lo / KD, kt1 = k_hi / KD;

  d4 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

  // this wave's share of a tile: k-rows wave and wave + 4 of both operands; running pointers (scalar), + KD k-rows per tile
  const double* pa = g.A + (int64_t)tm * BM + ((int64_t)kt0 * KD + wave) * g.lda;
  const double* pb = g.B + noff + ((int64_t)kt0 * KD + wave) * g.ldb;
  const int lane2 = 2 * lane;
  int issued = kt0;  // tile the running pointers stand at
  auto dma = [&](const int stage) {
    lds_double* As = lds + stage * STAGE + wave * PA;
    lds_double* Bs = lds + stage * STAGE + KD * PA + wave * PB;
    __builtin_amdgcn_global_load_lds((g_void*)(pa + lane2), (lds_void*)As, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((g_void*)(pb + lane2), (lds_void*)Bs, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((g_void*)(pa + 4 * g.lda + lane2), (lds_void*)(As + 4 * PA), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((g_void*)(pb + 4 * g.ldb + lane2), (lds_void*)(Bs + 4 * PB), 16, 0, 0);
    if (issued + 1 < kt1) {  // (past the end: stay on the last tile)
      pa += (int64_t)KD * g.lda;
      pb += (int64_t)KD * g.ldb;
      ++issued;
    }
  };
  double fa[2][WTM], fb[2][WTN];
  auto frags = [&](const int kt, const int k4, const int buf) {
    const lds_double* As = lds + (kt & (NS - 1)) * STAGE + (4 * k4 + kq) * PA + r16;
    const lds_double* Bs = lds + (kt & (NS - 1)) * STAGE + KD * PA + (4 * k4 + kq) * PB + r16;
#pragma unroll
    for (int i = 0; i < WTM; ++i) fa[buf][i] = As[wm * (16 * WTM) + i * 16];
#pragma unroll
    for (int j = 0; j < WTN; ++j) fb[buf][j] = Bs[wn * (16 * WTN) + j * 16];
  };
  auto mfmas = [&](const int buf) {
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[buf][i], fb[buf][j], acc[i][j], 0, 0, 0);
  };
  auto pin_step = [&]() {  // the next step's four fragment reads (ds_read2_b64) go out behind the first MFMAs of this one
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
    }
    __builtin_amdgcn_sched_group_barrier(0x008, WTM * WTN - 8, 0);
  };

  if (kt0 < kt1) {
    dma(kt0 & (NS - 1));
    dma((kt0 + 1) & (NS - 1));
    dma((kt0 + 2) & (NS - 1));
    dma((kt0 + 3) & (NS - 1));
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    frags(kt0, 0, 0);
    for (int kt = kt0; kt < kt1; ++kt) {
      frags(kt, 1, 1);
      mfmas(0);
      pin_step();
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      dma(kt & (NS - 1));
      frags(kt + 1, 0, 0);  // (behind the last tile: a stage of garbage, never used)
      mfmas(1);
      pin_step();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of the ring may land after this workgroup has left
  }

  double* __restrict__ Cg = g.C + (g.cblk_stride > 0 ? gemm_noff(tn, BN, g.cblk_stride) : noff) + wn * (16 * WTN) + r16;
  const int64_t m0 = (int64_t)tm * BM + wm * (16 * WTM) + kq;
  if (g.beta == 0.0) {
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* row = Cg + (m0 + i * 16 + 4 * r) * g.ldc;
#pragma unroll
        for (int j = 0; j < WTN; ++j) row[j * 16] = g.alpha * acc[i][j][r];
      }
  } else {
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* row = Cg + (m0 + i * 16 + 4 * r) * g.ldc;
        double c[WTN];
#pragma unroll
        for (int j = 0; j < WTN; ++j) c[j] = row[j * 16];
#pragma unroll
        for (int j = 0; j < WTN; ++j) row[j * 16] = g.beta * c[j] + g.alpha * acc[i][j][r];
      }
  }
}

__global__ __launch_bounds__(256, 2) void gemm_f64_dma_kernel(GemmArgs g) {
  gemm_f64_dma_body<>(g, blockIdx.x);
}

// The same body under a second name: the Cholesky's bulk trailing updates (SYRK / GEMM on the factor's trailing columns), so
// that profilers list that population -- the one the north star states its MFMA target on -- as a kernel of its own.
__global__ __launch_bounds__(256, 2) void gemm_f64_dma_chol_update_kernel(GemmArgs g) {
  gemm_f64_dma_body<>(g, blockIdx.x);
}

template <int WGM, int WGN, int WTM, int WTN, int OCC, bool PFC = false>
__global__ __launch_bounds__(64 * WGM * WGN, OCC) void gemm_f64_kernel(GemmArgs g) {
  gemm_f64_body<WGM, WGN, WTM, WTN, PFC>(g, blockIdx.x);
}

// Batched form: blockIdx.y selects one of several INDEPENDENT products whose descriptors sit in
// device memory (each with its own schedule; surplus blocks of the shorter ones exit at once).
// One launch per level of the triangular-inverse tree instead of one per node: at N = 10k the
// per-node launches of the small levels cost 1.8 ms of a 15.6 ms gradient in launch latency alone.
template <int WGM, int WGN, int WTM, int WTN, int OCC>
__global__ __launch_bounds__(64 * WGM * WGN, OCC) void gemm_f64_batched_kernel(const GemmArgs* __restrict__ batch) {
  const GemmArgs g = batch[blockIdx.y];  // uniform address, read before any store: scalar loads
  gemm_f64_body<WGM, WGN, WTM, WTN>(g, blockIdx.x);
}

// MFMA-only microbenchmark: the GEMM's own register pattern (4 x 4 independent accumulators fed
// by 4 + 4 operand registers) with no memory traffic -- the f64 matrix rate a kernel of this shape
// can sustain, i.e. the practical ceiling the roofline is compared with.
__global__ __launch_bounds__(256, 2) void mfma_f64_peak_kernel(double* sink, int iters, double scale) {
  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
  double a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = scale * (1.0 + 1e-9 * (threadIdx.x + i));
    b[i] = scale * (1.0 - 1e-9 * (threadIdx.x + 7 * i));
  }
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  const long long t1 = __builtin_readcyclecounter();
  if (s == 12345.678) sink[0] = s;  // keep the chain alive without a store on the timed path
  if (blockIdx.x == 0 && threadIdx.x == 0) sink[1] = (double)(t1 - t0);
}

// Host side: fill g.xstart with work-balanced cuts of the row-major list of computed tiles and
// return the grid size (8 x the longest run).  *flops receives the flops the launch performs.
// A tile's work is its k range in units of KT plus a constant for prologue / epilogue; for the
// patterns the engine uses it depends on tn only, so prefix sums make this O(mt + nt).
inline int gemm_schedule(GemmArgs& g, int bm, int bn, double* flops) {
  const int KU = g.k / KT;
  auto first_of = [&](int tm) {
    if (!g.tri) return 0;
    const int f = gemm_first_tn((int64_t)tm * bm - g.tri_off - (bn - 1), bn, g.nblk_stride);
    return f > g.nt ? g.nt : f;
  };
  auto kunits = [&](int tm, int tn) {
    const int kstride = g.krow_stride > 0 ? g.krow_stride : (g.nblk_stride < 1 ? 1 : g.nblk_stride);
    const int krow = (int)gemm_noff(tn, bn, kstride) + g.krow_off;
    const int lo = std::max(0, g.klo_m * tm * bm + g.klo_n * krow) / KT;
    int hi = KU;
    if (g.khi_n && (krow + bn) / KT < hi) hi = (krow + bn) / KT;
    return hi > lo ? hi - lo : 0;
  };
  if (g.order != 0) {  // n-major list in dispatch order: no XCD runs, grid = number of tiles
    long long nact = 0, funits = 0;
    for (int tn = 0; tn < g.nt; ++tn) {
      long long cnt = g.mt;
      if (g.tri) cnt = std::min<long long>(cnt, ((long long)tn * bn + bn - 1 + g.tri_off) / bm + 1);
      nact += cnt;
      funits += cnt * kunits(0, tn);
    }
    if (flops) *flops = 2.0 * bm * bn * KT * (double)funits;
    for (int i = 0; i <= 8; ++i) g.xstart[i] = 0;
    return (int)nact;
  }
  static thread_local long long* P = nullptr;
  static thread_local long long* F = nullptr;
  static thread_local int cap = 0;
  if (cap < g.nt + 1) {
    delete[] P;
    delete[] F;
    cap = g.nt + 1;
    P = new long long[cap];
    F = new long long[cap];
  }
  const int fixed = 4;  // prologue + epilogue expressed in k-tile units
  P[0] = F[0] = 0;
  for (int tn = 0; tn < g.nt; ++tn) {
    const int w = kunits(0, tn);
    P[tn + 1] = P[tn] + w + fixed;
    F[tn + 1] = F[tn] + w;
  }
  long long total = 0, funits = 0, nact = 0;
  for (int tm = 0; tm < g.mt; ++tm) {
    const int f = first_of(tm);
    total += P[g.nt] - P[f];
    funits += F[g.nt] - F[f];
    nact += g.nt - f;
  }
  if (g.klo_m) {  // rare: recount the flops exactly
    funits = 0;
    for (int tm = 0; tm < g.mt; ++tm)
      for (int tn = first_of(tm); tn < g.nt; ++tn) funits += kunits(tm, tn);
  }
  if (flops) *flops = 2.0 * bm * bn * KT * (double)funits;
  if (g.strip > 0) {
    // work-balanced cuts of the strip-major list: walk the strips, inside the strip that holds a cut walk its
    // n-tiles (tile work depends on tn only: P)
    g.xstart[0] = 0;
    for (int i = 0; i < 8; ++i) g.xs_t0[i] = g.xs_base[i] = 0;
    long long acc = 0, ci = 0;
    int x = 1;
    for (int t0 = 0; t0 < g.mt && x < 8; t0 += g.strip) {
      const int t1 = std::min(t0 + g.strip, g.mt), sz = t1 - t0;
      long long sw = 0, sc = 0;
      for (int t = t0; t < t1; ++t) {
        const int f = first_of(t);
        sw += P[g.nt] - P[f];
        sc += g.nt - f;
      }
      if ((acc + sw) * 8 >= (long long)x * total) {  // one or more cuts fall inside this strip
        long long a2 = acc, c2 = ci;
        int have = 1;
        for (int tn = first_of(t0); tn < g.nt && x < 8; ++tn) {
          while (have < sz && first_of(t0 + have) <= tn) ++have;
          const long long w1 = P[tn + 1] - P[tn];
          for (int q = 0; q < have && x < 8; ++q) {
            if ((a2 + w1) * 8 >= (long long)x * total) {
              g.xs_t0[x] = t0;
              g.xs_base[x] = (int)ci;
              g.xstart[x++] = (int)(c2 + 1);
            }
            a2 += w1;
            ++c2;
          }
        }
      }
      acc += sw;
      ci += sc;
    }
    while (x <= 8) g.xstart[x++] = (int)nact;
    g.xstart[8] = (int)nact;
    int longest = 0;
    for (int i = 0; i < 8; ++i) {
      if (g.xstart[i + 1] < g.xstart[i]) g.xstart[i + 1] = g.xstart[i];
      longest = std::max(longest, g.xstart[i + 1] - g.xstart[i]);
    }
    return longest * 8;
  }
  g.xstart[0] = 0;
  long long acc = 0, ci = 0;
  int x = 1;
  for (int tm = 0; tm < g.mt && x < 8; ++tm) {
    const int f = first_of(tm);
    const long long roww = P[g.nt] - P[f];
    while (x < 8 && (acc + roww) * 8 >= (long long)x * total) {
      int lo = f, hi = g.nt;  // first tn with acc + (P[tn] - P[f]) >= x * total / 8
      while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if ((acc + P[mid] - P[f]) * 8 >= (long long)x * total) hi = mid; else lo = mid + 1;
      }
      g.xstart[x++] = (int)(ci + (lo - f));
    }
    acc += roww;
    ci += g.nt - f;
  }
  while (x <= 8) g.xstart[x++] = (int)nact;
  g.xstart[8] = (int)nact;
  int longest = 0;
  for (int i = 0; i < 8; ++i) {
    if (g.xstart[i + 1] < g.xstart[i]) g.xstart[i + 1] = g.xstart[i];
    const int len = g.xstart[i + 1] - g.xstart[i];
    longest = longest > len ? longest : len;
  }
  return longest * 8;
}

}  // namespace gmb
