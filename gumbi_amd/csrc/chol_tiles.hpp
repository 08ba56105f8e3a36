// chol_tiles.hpp -- the whole Cholesky factorisation of a matrix of 0.8k .. 28k rows as ONE persistent launch (and the
// triangular solve of the predict path in the same form).
//
// Why: at C2's size (N = 10k, 79 block columns) the multi-kernel schedules of engine.hip are bound by their
// latency chain (leaf -> strip solve -> rank-128 update, 79 times) and by what the stream model can overlap with it:
// 8.8 ms where the flops alone need 4.7 ms.  Here the factorisation is a list of TILE TASKS served
// by resident workgroups (512 threads, one per compute unit) from one ticket counter; dependencies are per-tile flags
// in global memory, so bulk work and the chain interleave at tile granularity without any kernel boundary.
//
// Task (I, J), I >= J, owns the 128 x 128 tile (block row I, block column J) and is LEFT-LOOKING:
//   1. T = A(I,J) - sum_{k < J} L(I,k) L(J,k)^T   -- one long-k MFMA contraction (the k loop of gemm_f64.hpp),
//      which walks the k-blocks in order and waits for the flags of L(I,k), L(J,k) only when it reaches a block
//      that is not final yet: everything that CAN be accumulated early is;
//   2. I == J:  L(J,J) = chol(T) with the leaf of potrf_leaf.hpp (T goes straight into the leaf's LDS layout),
//      sub-block inverses and log-det as in the stand-alone leaf kernel;
//      I >  J:  L(I,J) = T L(J,J)^-T with the strip solve of trsm_strip.hpp, one 16-row slab per wavefront --
//      after the flag of L(J,J), or, for the two tiles right below the diagonal, FOLLOWING the leaf column by
//      column (LeafArgs::prog);
//   3. publish the tile: its values were stored write-through (sc1), every wave drains, barrier, the flag.
// Tickets are handed out in column-major order (J outer, I = J first), which is a topological order of the
// dependency graph: a task only ever waits for tasks with smaller tickets, which are finished or held by a running
// workgroup -- the launch cannot deadlock whatever the residency or placement of its workgroups.  Every tile is
// written once by its owner and read by others only after its flag, and each contraction runs in k order in one
// accumulator, so the factor is bit-reproducible from run to run (it does differ in the last bits from the
// recursive schedule, which groups the same sums differently).
//
// The latency chain -- (J,J) -> (J+1,J) -> (J+1,J+1) -- is what bounds small matrices; its tiles hand their results on
// in pieces: the leaf publishes finished 16-column blocks (prog), the two tiles below the diagonal publish finished
// 32-column quarters (half[]), and the next column's chain tiles take their last k-block quarter by quarter.
//
// Inter-workgroup visibility follows the gfx950 recipe (per-XCD L2s are not coherent, L1 is per CU): producer =
// write-through (sc1) stores, every wave drains, barrier, relaxed agent-scope flag store; consumer = ONE wave polls
// relaxed, ONE agent-scope acquire, barrier, plain loads -- or, for the pieces above, write-through-coherent (sc1)
// loads and no fence.  Every wait is bounded: a wave that waits longer than `timeout_us` (and a million polls) raises
// the abort word, every waiter sees it and the launch drains (the host reports GMB_EHIP) -- a lost flag can cost a
// factorisation, never the GPU.
//
// Replaces the per-evaluation dpotrf of pm.gp.Marginal (gumbi/regression/pymc/GP.py:811, 845-847) for the sizes
// Gumbi users actually fit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_f64.hpp"
#include "potrf_leaf.hpp"
#include "trsm_strip.hpp"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "chol_tiles.hpp: the publication protocol (sc1 write-through stores + relaxed agent-scope flags, one acquire per consuming workgroup) is written for gfx950's cache hierarchy"
#endif

namespace gmb {

// k-tile depth of the contraction: 16 with two workgroups per compute unit (2 x 73,728 B of staging fit beside each other),
// 32 with ONE eight-wave workgroup per compute unit (147,456 B) -- half the barriers per flop, and nobody else's MFMAs to
// fill the barrier bubbles with
__host__ __device__ constexpr int ct_kt(int nw) { return nw == 8 ? 32 : 16; }
__host__ __device__ constexpr int ct_lds_doubles(int nw) {
  return LEAF_LDS_DOUBLES > 2 * ct_kt(nw) * (2 * PITCH) ? LEAF_LDS_DOUBLES : 2 * ct_kt(nw) * (2 * PITCH);
}

struct CholTilesArgs {
  double* A;        // factor buffer: lower triangle of Sigma in, L out (column-major, leading dimension ld)
  int64_t ld;
  int32_t nct, nrt; // block columns / block rows (nrt >= nct: the rows below the square ride along, e.g. the y row)
  int64_t N;        // order of the matrix (ragged last diagonal block)
  double* dinv16;   // nct x 8 x 256 sub-block inverses (out)
  double* logdet;   // += sum log L_cc
  int32_t* info;    // first non-positive pivot (global row + 1), 0 = ok
  uint32_t* flags;  // nrt * nct words, zeroed by the host before the launch; tile (I, J) is final when [I * nct + J] != 0
  uint32_t* half;   // 2 x nct words (zeroed with the flags): how many 32-column quarters (0 .. 4) of the tiles (J + 1, J) [word J] and
                    // (J + 2, J) [word nct + J] are final -- the two tiles below the diagonal publish themselves quarter by quarter
  uint32_t* prog;   // nct words (zeroed with the flags): progress of the leaf of column J (LeafArgs::prog)
  uint32_t* ctl;    // [0] ticket counter, [1] abort word (zeroed with the flags)
  int32_t ntasks;
  uint32_t timeout_us;
  unsigned long long* dbg;  // optional, 4 x ntasks: wall-clock stamps (100 MHz) taken / contraction done / solve input ready / published
  // Triangular solve V <- V L^-T with the factor in A (trsm_tiles_kernel; the predict path): V is ntm x nct tiles of 128 x 128,
  // column-major with leading dimension ldv (rows = test points); task (r, c) owns tile (r, c) of V and waits for the tiles
  // (r, k < c) of its own row only -- flags is ntm x nct then, and there are no diagonal tasks.
  double* V;
  int64_t ldv;
  int32_t ntm;
  // A launch on a PANEL of a larger matrix (engine.hip: chol_tiles_panel -- the bottom of the large matrices' recursion): A,
  // dinv16 and N are the panel's own (origin = its diagonal's first element); row_base = the global index of that element,
  // for the failure index only.
  int64_t row_base;
};

__host__ __device__ inline int64_t ct_col_start(int J, int nrt) { return (int64_t)J * nrt - (int64_t)J * (J - 1) / 2; }
__host__ __device__ inline int ct_task_count(int nct, int nrt) { return (int)ct_col_start(nct, nrt); }
// ticket -> tile, column-major over the lower block triangle (column J holds I = J .. nrt-1)
__host__ __device__ inline void ct_decode(int t, int nct, int nrt, int& I, int& J) {
  const double b = 2.0 * nrt + 1.0;
  double disc = b * b - 8.0 * (double)t;
  if (disc < 0.0) disc = 0.0;
  int j = (int)((b - sqrt(disc)) * 0.5);
  if (j < 0) j = 0;
  if (j > nct - 1) j = nct - 1;
  while (j > 0 && ct_col_start(j, nrt) > t) --j;
  while (j + 1 < nct && ct_col_start(j + 1, nrt) <= t) ++j;
  J = j;
  I = j + (t - (int)ct_col_start(j, nrt));
}

#define CT_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// a pointer every lane holds the same value of, moved to scalar registers (address arithmetic and the LDS-DMA's base stay
// on the scalar unit)
__device__ __forceinline__ const double* et_uni_ptr_c(const double* p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (const double*)(((uint64_t)hi << 32) | lo);
}

// wave-uniform: true when the wait must be abandoned (somebody raised the abort word, or this wave has waited
// longer than the time-out and raises it itself)
__device__ __forceinline__ bool ct_give_up(const CholTilesArgs& g, unsigned& spins, unsigned long long& t0) {
  if ((++spins & 63u) != 0u) return false;
  if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(g.ctl + 1, CT_RLX_AGENT)) != 0u) return true;
  const unsigned long long now = wall_clock64();
  if (t0 == 0ull) {
    t0 = now;
    return false;
  }
  // both the wall clock AND a million polls actually executed: a process that was swapped off the GPU for seconds (several
  // ranks sharing one device) comes back with the clock advanced but nothing wrong
  if (now - t0 > (unsigned long long)g.timeout_us * 100ull && spins > (1u << 20)) {
    __hip_atomic_store(g.ctl + 1, 1u, CT_RLX_AGENT);
    return true;
  }
  return false;
}

// Called by ONE wave (all 64 lanes).  Waits until k-block kb0 of block rows I and J is final, then returns the end
// of the run of final k-blocks that starts there (<= kb_end; at most 32 blocks further), after ONE agent-scope
// acquire.  -1: the launch is being abandoned.
__device__ __forceinline__ int ct_wait_rows(const CholTilesArgs& g, const int I, const int J, const int kb0, const int kb_end) {
  const int lane = threadIdx.x & 63;
  const int idx = kb0 + (lane & 31);
  const bool mine = idx < kb_end;
  const uint32_t* p = g.flags + (int64_t)(lane < 32 ? I : J) * g.nct + (mine ? idx : kb0);
  unsigned spins = 0;
  unsigned long long t0 = 0ull;
  for (;;) {
    const uint32_t v = __hip_atomic_load(p, CT_RLX_AGENT);
    const unsigned long long m = __ballot(v != 0u || !mine);
    const uint32_t both = (uint32_t)m & (uint32_t)(m >> 32);
    const int n = both == 0xffffffffu ? 32 : __builtin_ctz(~both);
    if (n > 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      const int e = kb0 + n;
      return e < kb_end ? e : kb_end;
    }
    if (ct_give_up(g, spins, t0)) return -1;
    // only the tiles next to the diagonal are on the critical chain: everybody else polls gently (hundreds of
    // workgroups wait at the start and the end of a factorisation, and their polls share the fabric with the chain)
    if (I <= J + 1) __builtin_amdgcn_s_sleep(2);
    else __builtin_amdgcn_s_sleep(40);
  }
}

// ONE wave: wait for a single flag; 0 = final (acquired), -1 = abandoned
__device__ __forceinline__ int ct_wait_one(const CholTilesArgs& g, const uint32_t* p, const bool urgent) {
  unsigned spins = 0;
  unsigned long long t0 = 0ull;
  for (;;) {
    if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(p, CT_RLX_AGENT)) != 0u) {  // every lane reads the same word
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      return 0;
    }
    if (ct_give_up(g, spins, t0)) return -1;
    if (urgent) __builtin_amdgcn_s_sleep(1);
    else __builtin_amdgcn_s_sleep(40);
  }
}

// ONE wave: wait until both words are >= need (the same word twice = one word); 0 = there (acquired unless the caller reads
// with write-through-coherent loads and asks for no fence), -1 = abandoned
__device__ __forceinline__ int ct_wait_two(const CholTilesArgs& g, const uint32_t* p, const uint32_t* q, const uint32_t need,
                                           const bool acquire) {
  unsigned spins = 0;
  unsigned long long t0 = 0ull;
  for (;;) {
    const uint32_t a = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, CT_RLX_AGENT));
    const uint32_t b = __builtin_amdgcn_readfirstlane(__hip_atomic_load(q, CT_RLX_AGENT));
    if (a >= need && b >= need) {
      if (acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      return 0;
    }
    if (ct_give_up(g, spins, t0)) return -1;
    __builtin_amdgcn_s_sleep(1);
  }
}

// T = A(I,J) - sum_{k < 128 J} L(I,k) L(J,k)^T, written over A(I,J).  The k loop of gemm_f64_body (same staging,
// same pinned instruction order) on a 128 x 128 tile, cut into segments at the k-blocks whose tiles were not final
// yet.  NW = 4: 2 x 2 waves of 4 x 4 MFMA tiles (two workgroups per compute unit); NW = 8: 4 x 2 waves of 2 x 4 MFMA
// tiles (one workgroup per compute unit).  Every thread of the workgroup calls it; false = the launch is being
// abandoned (uniform).
// The accumulators start at -A(I,J) (requested first: the latency hides behind the first wait), so the epilogue is a
// store of -acc with no read.  to_lds (diagonal tiles with a full block): the result goes straight into the leaf's
// packed LDS layout (potrf_leaf_core<.., PRE = true>) instead of global memory.
// TRSM: the "n" operand and the tile itself live in V (row tile I of the right-hand sides) instead of the factor buffer, and
// only the flags of row I matter (the factor is final).
// eight-wave tasks stage the diagonal block of their strip solve through LDS (ct_strip_solve_lds); 0 = the global-operand form (A/B)
#ifndef CT_STRIP_LDS
#define CT_STRIP_LDS 1
#endif
// the ragged last block's tasks contract their real sixteen-wide groups only (ct_ksum<.., RAG>); 0 = whole tiles (A/B)
#ifndef CT_RAGGED
#define CT_RAGGED 1
#endif

// RAG (the ragged last block): only the first nrow16 sixteen-row groups of the tile (the last block ROW of
// the bordered matrix: N + 1 - 128 I real rows) or the first ncol16 sixteen-column groups (the last block COLUMN of the solve:
// N - 128 J real columns) are contracted -- the others would multiply the padding's zeros.  Every computed element is the same
// sum in the same order as in the full form (same bits); the waves are dealt so that the live groups sit on all four SIMDs
// (a row cut keeps the waves 0 .. 3 busy: wave -> (column quarter, row half) = (wave % 4, wave / 4) instead of (wave / 2, wave % 2)).
template <int NW, bool TRSM = false, bool RAG = false>
__device__ __forceinline__ bool ct_ksum(const CholTilesArgs& g, const int I, const int J, double* __restrict__ lds,
                                        int* s_i, const bool to_lds, const int nrow16 = 8, const int ncol16 = 8) {
  constexpr int WGN = 2, WGM = NW / WGN;           // waves along n (rows) and m (columns)
  constexpr int WTM = TILE / (16 * WGM), WTN = TILE / (16 * WGN);  // MFMA tiles per wave
  constexpr int KT = ct_kt(NW);            // (shadows gmb::KT of the launch-based GEMM)
  constexpr int PA = PITCH;                // LDS row pitch (doubles)
  constexpr int LA = TILE / 2, RA = 64 * NW / LA, NA = KT / RA;  // 64 lanes per k-row, RA k-rows per pass, NA passes
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool rowcut = RAG && nrow16 < 8;
  const int wm = rowcut ? wave % WGM : wave / WGN, wn = rowcut ? wave / WGM : wave % WGN;
  // live MFMA tiles of this wave (RAG): i < ni along the columns, j < nj along the rows
  const int ni = RAG ? max(0, min(WTM, ncol16 - wm * WTM)) : WTM, nj = RAG ? max(0, min(WTN, nrow16 - wn * WTN)) : WTN;
  const int r16 = lane & 15, kq = lane >> 4;
  const int s_row = tid / LA, s_col = 2 * (tid % LA);
  const double* __restrict__ Ag = g.A + (int64_t)J * TILE + s_col;  // "m" operand: rows of block row J = columns of the tile
  const int64_t ldb = TRSM ? g.ldv : g.ld;
  const double* __restrict__ Bg = (TRSM ? g.V : g.A) + (int64_t)I * TILE + s_col;  // "n" operand: rows of block row I = rows of the tile

  // D layout of v_mfma_f64_16x16x4_f64: n = lane & 15, m = (lane >> 4) + 4 reg
  double* __restrict__ Cg = (TRSM ? g.V : g.A) + (int64_t)I * TILE + wn * (16 * WTN) + r16;
  const int64_t m0 = (int64_t)J * TILE + wm * (16 * WTM) + kq;
  d4 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < WTN; ++j) acc[i][j][r] = -Cg[(m0 + i * 16 + 4 * r) * ldb + j * 16];
  d2 ra[NA], rb[NA];

  auto gload = [&](int kt) {
#pragma unroll
    for (int p = 0; p < NA; ++p) ra[p] = *reinterpret_cast<const d2*>(Ag + ((int64_t)kt * KT + s_row + RA * p) * g.ld);
#pragma unroll
    for (int p = 0; p < NA; ++p) rb[p] = *reinterpret_cast<const d2*>(Bg + ((int64_t)kt * KT + s_row + RA * p) * ldb);
  };
  // the same through write-through-coherent (sc1) loads: data another workgroup stored sc1 moments ago, read without an
  // acquire fence (8-byte pieces: the widest agent-scope atomic load)
  auto gload_sc1 = [&](int kt) {
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      const double* pa = Ag + ((int64_t)kt * KT + s_row + RA * p) * g.ld;
      ra[p][0] = __hip_atomic_load(pa, CT_RLX_AGENT);
      ra[p][1] = __hip_atomic_load(pa + 1, CT_RLX_AGENT);
    }
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      const double* pb = Bg + ((int64_t)kt * KT + s_row + RA * p) * ldb;
      rb[p][0] = __hip_atomic_load(pb, CT_RLX_AGENT);
      rb[p][1] = __hip_atomic_load(pb + 1, CT_RLX_AGENT);
    }
  };
  auto lstore = [&](int st) {
    double* As = lds + st * (KT * 2 * PA);
    double* Bs = As + KT * PA;
#pragma unroll
    for (int p = 0; p < NA; ++p) *reinterpret_cast<d2*>(&As[(s_row + RA * p) * PA + s_col]) = ra[p];
#pragma unroll
    for (int p = 0; p < NA; ++p) *reinterpret_cast<d2*>(&Bs[(s_row + RA * p) * PA + s_col]) = rb[p];
  };
  auto compute = [&](int st) {
    const double* As = lds + st * (KT * 2 * PA);
    const double* Bs = As + KT * PA;
#pragma unroll
    for (int k4 = 0; k4 < KT; k4 += 4) {
      double a[WTM], b[WTN];
#pragma unroll
      for (int i = 0; i < WTM; ++i) a[i] = As[(k4 + kq) * PA + wm * (16 * WTM) + i * 16 + r16];
#pragma unroll
      for (int j = 0; j < WTN; ++j) b[j] = Bs[(k4 + kq) * PA + wn * (16 * WTN) + j * 16 + r16];
#pragma unroll
      for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
          if constexpr (RAG) {
            if (i < ni && j < nj) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);  // (wave-uniform)
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
          }
        }
    }
  };

  // The contraction runs in SEGMENTS of k-tiles whose operands are final.  Ordinary tiles: runs of k-blocks by the
  // tile flags of block rows I and J.  The tiles of the latency chain -- the diagonal tile and the one below it -- take
  // their LAST k-block, which comes from the tiles (I, J-1) and (J, J-1) of the column before (the chain's own output), in
  // four quarters: those tiles publish themselves 32 columns at a time while they are being solved (half[]); the
  // quarters are read with write-through-coherent loads, so their waits need no fence.
  constexpr int KPB = TILE / KT;  // k-tiles per k-block
  constexpr int KPQ = KPB / 4;    // k-tiles per quarter
  const int kt_end = J * KPB;
  const bool diag = !TRSM && I == J;
  const bool chain = !TRSM && I <= J + 1;
  int ktc = 0;
  while (ktc < kt_end) {
    const int kb = ktc / KPB;
    const bool quarter = chain && kb == J - 1;
    if (wave == 0) {
      int r;
      if (quarter) {
        // tile (J, J-1) is the first sub-diagonal tile of column J-1, tile (J+1, J-1) the second
        const uint32_t* hp = g.half + kb;
        const uint32_t* hq = diag ? hp : g.half + g.nct + kb;
        const int q = (ktc - kb * KPB) / KPQ;
        r = ct_wait_two(g, hp, hq, (uint32_t)(q + 1), false);
        if (r >= 0) r = ktc + KPQ;
      } else {
        r = ct_wait_rows(g, I, TRSM ? I : J, kb, chain ? J - 1 : J);
        if (r >= 0) r *= KPB;
      }
      s_i[1] = r;  // (every lane of wave 0 stores the same value)
    }
    __syncthreads();
    const int kt1 = __builtin_amdgcn_readfirstlane(s_i[1]);  // workgroup-uniform: keep the control flow scalar
    if (kt1 < 0) return false;
    const int kt0 = ktc;
    if (quarter) {  // KPQ k-tiles, not pipelined among themselves (one per quarter with k-tiles of 32)
      for (int kt = kt0; kt < kt1; ++kt) {
        gload_sc1(kt);
        lstore(0);
        __syncthreads();
        compute(0);
        __syncthreads();
      }
      ktc = kt1;
      continue;
    }
    gload(kt0);
    lstore(0);
    __syncthreads();
    int st = 0;
    constexpr int NMFMA = WTM * WTN * (KT / 4), NMEM = 2 * NA;
    constexpr int SLOT = NMFMA / (4 * NMEM);
    static_assert(SLOT >= 1 && 2 * SLOT * NMEM <= NMFMA, "not enough MFMAs to interleave the staging with");
    for (int kt = kt0; kt + 1 < kt1; ++kt) {
      gload(kt + 1);
      compute(st);
      lstore(st ^ 1);
      if constexpr (!RAG) {  // (the ragged form's MFMAs sit behind scalar branches: nothing to pin)
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
      }
      __syncthreads();
      st ^= 1;
    }
    compute(st);
    __syncthreads();  // the next segment's first stage store (and the next wait's s_i) must not overtake slow waves
    ktc = kt1;
  }

  // epilogue: T = -acc
  if (to_lds) {
    // (the last barrier of the loop above has retired every read of the staging buffers the packed block overlays)
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = wm * (16 * WTM) + i * 16 + kq + 4 * r;
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
          const int rr = wn * (16 * WTN) + j * 16 + r16;
          if (rr >= (c & ~15)) lds[pk(rr, c)] = rr >= c ? -acc[i][j][r] : 0.0;
        }
      }
  } else {
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double* row = Cg + (m0 + i * 16 + 4 * r) * ldb;
#pragma unroll
        for (int j = 0; j < WTN; ++j) row[j * 16] = -acc[i][j][r];
      }
  }
  return true;
}

// A task is compiled as ONE non-inlined function per kind (diagonal / off-diagonal tile).  Inlined into the persistent
// loop, the register allocator kept the loop-invariant state of every phase live everywhere (474 SGPR + 418 VGPR
// spills); one function per PHASE (contraction / leaf / strip) put the callee's register save -- ~110 scratch stores,
// 3 - 5 us -- in front of the leaf and of the strip solve, i.e. twice per column on the latency chain.  With one call
// per task the save runs when the task starts (long before its last dependency arrives) and the restore after its
// tile is published.  Pointers travel as address_space(1) / (3) parameters and are put back into the structs inside:
// through a struct they would arrive GENERIC and the bodies would fall back to flat_* accesses (which count against
// lgkmcnt and stall the LDS pipeline of the contraction).
typedef __attribute__((address_space(3))) double ct_lds_double;
typedef __attribute__((address_space(3))) int ct_lds_int;
typedef __attribute__((address_space(1))) double ct_g_double;
typedef __attribute__((address_space(1))) uint32_t ct_g_u32;
typedef __attribute__((address_space(1))) int32_t ct_g_i32;
typedef __attribute__((address_space(1))) unsigned long long ct_g_u64;

struct CtPtrs {  // the global pointers of CholTilesArgs, typed
  ct_g_double* A;
  ct_g_double* dinv16;
  ct_g_double* logdet;
  ct_g_i32* info;
  ct_g_u32* flags;
  ct_g_u32* half;
  ct_g_u32* prog;
  ct_g_u32* ctl;
  ct_g_u64* dbg;
};

__device__ __forceinline__ CholTilesArgs ct_rebuild(const CholTilesArgs& g_in, ct_g_double* A, ct_g_double* dinv16, ct_g_double* logdet,
                                                    ct_g_i32* info, ct_g_u32* flags, ct_g_u32* half, ct_g_u32* ctl, ct_g_u64* dbg,
                                                    ct_g_double* V = nullptr, ct_g_u32* prog = nullptr) {
  CholTilesArgs g = g_in;
  g.V = (double*)V;
  g.prog = (uint32_t*)prog;
  g.half = (uint32_t*)half;
  g.A = (double*)A;
  g.dinv16 = (double*)dinv16;
  g.logdet = (double*)logdet;
  g.info = (int32_t*)info;
  g.flags = (uint32_t*)flags;
  g.ctl = (uint32_t*)ctl;
  g.dbg = (unsigned long long*)dbg;
  return g;
}

// publish tile (I, J): its final values were stored write-through (sc1), so there is no release fence -- every wave
// drains its stores, barrier, then the flag (stored by all lanes of wave 0: same word, same value)
__device__ __forceinline__ void ct_publish(const CholTilesArgs& g, const int I, const int J, const int t, const int wave) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (wave == 0) {
    if (I == J + 1 || I == J + 2) __hip_atomic_store(g.half + (int64_t)(I - J - 1) * g.nct + J, 4u, CT_RLX_AGENT);  // all four quarters
    __hip_atomic_store(g.flags + (int64_t)I * g.nct + J, 1u, CT_RLX_AGENT);
    if (g.dbg) g.dbg[4 * (int64_t)t + 3] = wall_clock64();
  }
}

// Diagonal tile (J, J): contraction, leaf factorisation, publication.  false = the launch is being abandoned.
template <int NW>
__device__ __noinline__ bool ct_diag_task(const CholTilesArgs g_in, ct_g_double* A, ct_g_double* dinv16, ct_g_double* logdet, ct_g_i32* info,
                                          ct_g_u32* flags, ct_g_u32* half, ct_g_u32* prog, ct_g_u32* ctl, ct_g_u64* dbg, const int J,
                                          const int t, ct_lds_double* l3, ct_lds_int* s3) {
  const CholTilesArgs g = ct_rebuild(g_in, A, dinv16, logdet, info, flags, half, ctl, dbg, nullptr, prog);
  double* lds = (double*)l3;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nvalid = (int)(g.N - (int64_t)J * TILE < TILE ? g.N - (int64_t)J * TILE : TILE);
  const bool pre = J > 0 && nvalid == TILE;  // the contraction leaves the block in the leaf's LDS layout
  if (J > 0 && !ct_ksum<NW>(g, J, J, lds, (int*)s3, pre)) return false;
  if (g.dbg && wave == 0) g.dbg[4 * (int64_t)t + 1] = wall_clock64();
  // (not pre: this workgroup's own epilogue stores are read back by other lanes) drain, then barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (g.dbg && wave == 0) g.dbg[4 * (int64_t)t + 2] = wall_clock64();
  LeafArgs a;
  a.A = g.A + (int64_t)J * TILE * (g.ld + 1);
  a.lda = g.ld;
  a.nvalid = nvalid;
  a.dinv16 = g.dinv16 + (int64_t)J * 8 * 256;
  a.logdet = g.logdet;
  a.info = g.info;
  a.row0 = g.row_base + (int64_t)J * TILE;
  a.dbg = nullptr;
  a.prog = NW == 8 ? g.prog + J : nullptr;  // (the strip solve below it follows the leaf column by column)
  if (pre) potrf_leaf_core<NW, true, true>(a, lds);
  else potrf_leaf_core<NW, true, false>(a, lds);
  __builtin_amdgcn_s_setprio(0);
  ct_publish(g, J, J, t, wave);
  return true;
}

// The strip solve of the CHAIN's tile (J + 1, J), following the leaf of (J, J) column by column: step s needs row block s
// of L(J, J) -- block columns < s -- and the sub-block inverse s, which the leaf publishes as `prog` = s + 1
// (LeafArgs::prog) while it is still factoring the rest.  Every wave polls for itself (relaxed, agent scope) and reads its
// operands with write-through-coherent loads (sc1: they were stored sc1 and must not come from this compute unit's L1),
// so there is no fence and no workgroup barrier per step.  The operands of step s + 1 are requested before step s is
// computed, as in trsm_strip_solve_store_pf; the first 64 columns are published half way (`half_flag`), the caller
// publishes the tile.  A poll that gives up (abort word / time-out) simply stops waiting: the result is garbage, the
// caller sees the abort word.
__device__ __forceinline__ void ct_strip_solve_pipelined(const CholTilesArgs& g, const TrsmArgs& ta, const int64_t r0, strip_d4 (&X)[8],
                                                        const uint32_t* prog, uint32_t* half_flag) {
  typedef strip_d4 d4;
  const int lane = threadIdx.x & 63;
  const int r16 = lane & 15, kq = lane >> 4;
  double* Bp = ta.B + r0 + r16;
  unsigned spins = 0;
  unsigned long long t0 = 0ull;
  bool gave_up = false;
  auto wait_prog = [&](const uint32_t need) {
    while (!gave_up && __builtin_amdgcn_readfirstlane(__hip_atomic_load(prog, CT_RLX_AGENT)) < need) {
      if (ct_give_up(g, spins, t0)) gave_up = true;
      else __builtin_amdgcn_s_sleep(1);
    }
  };
  double ops[2][32];
  auto load_step = [&](const int s, double (&o)[32]) {
    const double* Lrow = ta.L + 16 * s + r16;
#pragma unroll
    for (int t = 0; t < s; ++t)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) o[4 * t + kk] = __hip_atomic_load(&Lrow[(int64_t)(16 * t + 4 * kk + kq) * ta.ldl], CT_RLX_AGENT);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) o[28 + kk] = __hip_atomic_load(&ta.dinv16[s * 256 + (4 * kk + kq) * 16 + r16], CT_RLX_AGENT);
  };
  wait_prog(1u);
  load_step(0, ops[0]);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (s < 7) {
      wait_prog((uint32_t)(s + 2));
      load_step(s + 1, ops[(s + 1) & 1]);
    }
    asm volatile("" ::: "memory");  // keep the requests above the MFMAs below
    const double(&o)[32] = ops[s & 1];
    d4 y = X[s];
#pragma unroll
    for (int t = 0; t < s; ++t)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) y = __builtin_amdgcn_mfma_f64_16x16x4f64(-o[4 * t + kk], X[t][kk], y, 0, 0, 0);
    d4 x = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) x = __builtin_amdgcn_mfma_f64_16x16x4f64(o[28 + kk], y[kk], x, 0, 0, 0);
    X[s] = x;
    if (s == 1 || s == 3 || s == 5) {  // a quarter (32 columns) is final: out it goes, and the count with it
#pragma unroll
      for (int s2 = s - 1; s2 <= s; ++s2)
#pragma unroll
        for (int q = 0; q < 4; ++q) __hip_atomic_store(&Bp[(int64_t)(16 * s2 + kq + 4 * q) * ta.ldb], X[s2][q], CT_RLX_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0) __hip_atomic_store(half_flag, (uint32_t)((s + 1) / 2), CT_RLX_AGENT);
    }
  }
#pragma unroll
  for (int s = 6; s < 8; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) __hip_atomic_store(&Bp[(int64_t)(16 * s + kq + 4 * q) * ta.ldb], X[s][q], CT_RLX_AGENT);
}

// The strip solve of an eight-wave task with its operands staged through LDS.  trsm_strip_solve_store_pf reads L(J, J) and
// the sub-block inverses straight from global memory, every wave for itself, one step ahead: eight dependent L2 round trips
// per slab, 17 - 19 us per task of which the 144 MFMAs are 3 (profiles/r05_eval_pairs_ab.txt; 2.7 % of a C2 evaluation).  Here
// the workgroup pulls the block in ONCE by LDS-DMA -- columns 0 .. 111 of the factored diagonal block as a [column][row]
// image with the contraction's pitch (the last sixteen columns hold only the diagonal sub-block, which the solve replaces by
// its inverse), the eight 16 x 16 sub-block inverses behind it: 145,408 B, inside the contraction's ring, which is dead by
// then -- one round trip, then every operand is a ds_read.  Same operands, same MFMAs in the same order as the global form:
// the same bits.  Every thread of the workgroup calls it (two barriers inside); r0 = 16 * wave.
template <bool WT>
__device__ __forceinline__ void ct_strip_solve_lds(const TrsmArgs& g, const int64_t r0, strip_d4 (&X)[8], ct_lds_double* l3) {
  typedef strip_d4 d4;
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void g_void;
  constexpr int PA = PITCH;
  constexpr int NCOL = TILE - 16;      // columns of L(J, J) the solve reads
  constexpr int DINV_AT = NCOL * PA;   // the sub-block inverses behind the image
  static_assert(DINV_AT + 8 * 256 <= ct_lds_doubles(8), "the staged diagonal block must fit the contraction's ring");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const double* Lg = et_uni_ptr_c(g.L);
  const double* Dg = et_uni_ptr_c(g.dinv16);
  const int64_t ldl = g.ldl;
#pragma unroll
  for (int q = 0; q < NCOL / 8; ++q) {
    const int j = wave + 8 * q;
    __builtin_amdgcn_global_load_lds((g_void*)(Lg + (int64_t)j * ldl + 2 * lane), (lds_void*)(l3 + j * PA), 16, 0, 0);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int c = wave + 8 * q;
    __builtin_amdgcn_global_load_lds((g_void*)(Dg + c * 128 + 2 * lane), (lds_void*)(l3 + DINV_AT + c * 128), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  double* Bp = g.B + r0 + r16;
  double ops[2][32];
  auto load_step = [&](const int s, double (&o)[32]) {
    const ct_lds_double* Lrow = l3 + 16 * s + r16;
#pragma unroll
    for (int t = 0; t < s; ++t)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) o[4 * t + kk] = Lrow[(16 * t + 4 * kk + kq) * PA];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) o[28 + kk] = l3[DINV_AT + s * 256 + (4 * kk + kq) * 16 + r16];
  };
  const int smax = (g.nvalid + 15) / 16;  // (identity-padding sub-blocks of a ragged last block: see trsm_strip_solve_store_pf)
  load_step(0, ops[0]);
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    if (s >= smax) break;
    if (s < 7) load_step(s + 1, ops[(s + 1) & 1]);
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
      if constexpr (WT) __hip_atomic_store(&Bp[(int64_t)(16 * s + kq + 4 * q) * g.ldb], X[s][q], CT_RLX_AGENT);
      else Bp[(int64_t)(16 * s + kq + 4 * q) * g.ldb] = X[s][q];
    }
  __syncthreads();  // the image is the next contraction's ring: nobody may still be reading it when that starts
}

// Off-diagonal tile (I, J): contraction, strip solve against L(J, J), publication.  The slab's own data is requested
// BEFORE the wait for the diagonal block (it does not depend on it); the solved slab is stored write-through.
template <int NW>
__device__ __noinline__ bool ct_offdiag_task(const CholTilesArgs g_in, ct_g_double* A, ct_g_double* dinv16, ct_g_double* logdet,
                                             ct_g_i32* info, ct_g_u32* flags, ct_g_u32* half, ct_g_u32* prog, ct_g_u32* ctl, ct_g_u64* dbg,
                                             const int I, const int J, const int t, ct_lds_double* l3, ct_lds_int* s3) {
  const CholTilesArgs g = ct_rebuild(g_in, A, dinv16, logdet, info, flags, half, ctl, dbg, nullptr, prog);
  int* s_i = (int*)s3;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // the last block row of the bordered matrix holds N + 1 - 128 I real rows (the y row is its last): contract those only
  const int64_t rv = g.N + 1 - (int64_t)I * TILE;
  const int nrow16 = rv >= TILE ? 8 : (int)((rv + 15) / 16);
  if (J > 0) {
    bool ok;
    if (NW == 8 && CT_RAGGED && nrow16 < 8) ok = ct_ksum<NW, false, true>(g, I, J, (double*)l3, s_i, false, nrow16, 8);
    else ok = ct_ksum<NW>(g, I, J, (double*)l3, s_i, false);
    if (!__builtin_amdgcn_readfirstlane((int)ok)) return false;
  }
  if (g.dbg && wave == 0) g.dbg[4 * (int64_t)t + 1] = wall_clock64();
  TrsmArgs ta;
  ta.B = g.A + (int64_t)I * TILE + (int64_t)J * TILE * g.ld;
  ta.ldb = g.ld;
  ta.nrows = TILE;
  ta.L = g.A + (int64_t)J * TILE * (g.ld + 1);
  ta.ldl = g.ld;
  ta.dinv16 = g.dinv16 + (int64_t)J * 8 * 256;
  ta.nvalid = (int)(g.N - (int64_t)J * TILE < TILE ? g.N - (int64_t)J * TILE : TILE);
  // own epilogue stores of this tile are read back by other lanes: drain, barrier, then load
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  strip_d4 X0[8], X1[8];
  trsm_strip_load(ta, 16 * wave, X0);
  if constexpr (NW == 4) trsm_strip_load(ta, 16 * (wave + 4), X1);
  const bool near_chain = I <= J + 2;  // the two tiles below the diagonal feed the next column's chain tiles
  if (NW == 8 && near_chain && ta.nvalid == TILE) {
    // follow the leaf of (J, J) column by column instead of waiting for its flag
    if (g.dbg && wave == 0) g.dbg[4 * (int64_t)t + 2] = wall_clock64();
    __builtin_amdgcn_s_setprio(3);
    ct_strip_solve_pipelined(g, ta, 16 * wave, X0, g.prog + J, g.half + (int64_t)(I - J - 1) * g.nct + J);
    __builtin_amdgcn_s_setprio(0);
    if (wave == 0) s_i[1] = __builtin_amdgcn_readfirstlane(__hip_atomic_load(g.ctl + 1, CT_RLX_AGENT)) != 0u ? -1 : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(s_i[1]) < 0) return false;
  } else {
    if (wave == 0) s_i[1] = ct_wait_one(g, g.flags + (int64_t)J * g.nct + J, near_chain);
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(s_i[1]) < 0) return false;
    if (g.dbg && wave == 0) g.dbg[4 * (int64_t)t + 2] = wall_clock64();
    if (near_chain) __builtin_amdgcn_s_setprio(3);
    if constexpr (NW == 8 && CT_STRIP_LDS) {
      ct_strip_solve_lds<true>(ta, 16 * wave, X0, l3);
    } else {
      trsm_strip_solve_store_pf<true>(ta, 16 * wave, X0);
      if constexpr (NW == 4) trsm_strip_solve_store_pf<true>(ta, 16 * (wave + 4), X1);
    }
  }
  __builtin_amdgcn_s_setprio(0);
  ct_publish(g, I, J, t, wave);
  return true;
}

// Tile (r, c) of V <- V L^-T: contraction over the tiles (r, k < c) of its own row, strip solve against L(c, c) (final:
// no wait), publication for the later tiles of the row.
template <int NW>
__device__ __noinline__ bool ct_trsm_task(const CholTilesArgs g_in, ct_g_double* A, ct_g_double* dinv16, ct_g_double* V, ct_g_u32* flags,
                                          ct_g_u32* ctl, ct_g_u64* dbg, const int r, const int c, const int t, ct_lds_double* l3,
                                          ct_lds_int* s3) {
  const CholTilesArgs g = ct_rebuild(g_in, A, dinv16, nullptr, nullptr, flags, nullptr, ctl, dbg, V);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // the last block column holds N - 128 c real columns (behind them identity padding, whose results nobody reads): contract those only
  const int64_t cv = g.N - (int64_t)c * TILE;
  const int ncol16 = cv >= TILE ? 8 : (int)((cv + 15) / 16);
  if (c > 0) {
    bool ok;
    if (NW == 8 && CT_RAGGED && ncol16 < 8) ok = ct_ksum<NW, true, true>(g, r, c, (double*)l3, (int*)s3, false, 8, ncol16);
    else ok = ct_ksum<NW, true>(g, r, c, (double*)l3, (int*)s3, false);
    if (!__builtin_amdgcn_readfirstlane((int)ok)) return false;
  }
  if (g.dbg && wave == 0) g.dbg[4 * (int64_t)t + 1] = wall_clock64();
  TrsmArgs ta;
  ta.B = g.V + (int64_t)r * TILE + (int64_t)c * TILE * g.ldv;
  ta.ldb = g.ldv;
  ta.nrows = TILE;
  ta.L = g.A + (int64_t)c * TILE * (g.ld + 1);
  ta.ldl = g.ld;
  ta.dinv16 = g.dinv16 + (int64_t)c * 8 * 256;
  ta.nvalid = (int)(g.N - (int64_t)c * TILE < TILE ? g.N - (int64_t)c * TILE : TILE);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own epilogue stores are read back by other lanes
  __syncthreads();
  if (g.dbg && wave == 0) g.dbg[4 * (int64_t)t + 2] = wall_clock64();
  strip_d4 X0[8], X1[8];
  trsm_strip_load(ta, 16 * wave, X0);
  if constexpr (NW == 4) trsm_strip_load(ta, 16 * (wave + 4), X1);
  if constexpr (NW == 8 && CT_STRIP_LDS) {
    ct_strip_solve_lds<true>(ta, 16 * wave, X0, l3);
  } else {
    trsm_strip_solve_store_pf<true>(ta, 16 * wave, X0);
    if constexpr (NW == 4) trsm_strip_solve_store_pf<true>(ta, 16 * (wave + 4), X1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (wave == 0) {
    __hip_atomic_store(g.flags + (int64_t)r * g.nct + c, 1u, CT_RLX_AGENT);
    if (g.dbg) g.dbg[4 * (int64_t)t + 3] = wall_clock64();
  }
  return true;
}

// V <- V L^-T as ONE persistent launch (the predict path for matrices the tile Cholesky factors): tickets in column-major
// order (c outer, r inner) -- a task waits only for earlier tiles of its own row, so the order is topological and the
// launch cannot deadlock.  Unlike the stream form (79 strip launches at N = 10k, each a kernel boundary on every row) the
// rows advance independently and the contraction runs at the tile Cholesky's rate.
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void trsm_tiles_kernel(CholTilesArgs g) {
  __shared__ __attribute__((aligned(16))) double lds[ct_lds_doubles(NW)];
  __shared__ int s_i[4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  auto draw_ticket = [&]() {  // (see chol_tiles_body: wave-uniform, every lane the same operands)
    const unsigned old = atomicAdd(g.ctl, lane == 0 ? 1u : 0u);
    s_i[0] = __builtin_amdgcn_readfirstlane((int)old);
  };
  if (wave == 0) draw_ticket();
  __syncthreads();
  for (;;) {
    const int t = __builtin_amdgcn_readfirstlane(s_i[0]);
    if (t >= g.ntasks) return;
    const int c = t / g.ntm, r = t - c * g.ntm;
    if (g.dbg && wave == 0) g.dbg[4 * (int64_t)t + 0] = wall_clock64();
    const bool ok = ct_trsm_task<NW>(g, (ct_g_double*)g.A, (ct_g_double*)g.dinv16, (ct_g_double*)g.V, (ct_g_u32*)g.flags, (ct_g_u32*)g.ctl,
                                     (ct_g_u64*)g.dbg, r, c, t, (ct_lds_double*)lds, (ct_lds_int*)s_i);
    if (!__builtin_amdgcn_readfirstlane((int)ok)) return;
    if (wave == 0) draw_ticket();
    __syncthreads();
  }
}

template <int NW>
__device__ __forceinline__ void chol_tiles_body(const CholTilesArgs& g) {
  __shared__ __attribute__((aligned(16))) double lds[ct_lds_doubles(NW)];
  __shared__ int s_i[4];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Everything one thread would do for the workgroup -- draw a ticket, publish a tile -- is done by ALL 64 lanes of
  // wave 0 with identical operands, under a wave-uniform (scalar) branch: no lane-divergent control flow anywhere near
  // a barrier.  Written as `if (tid == 0)` at the top and bottom of the loop, the compiler fused the two regions across
  // the back edge and let the other 63 lanes of wave 0 run ahead to the next barrier while lane 0 was parked: the
  // barrier completed without the new ticket and the workgroup span on the old one for ever (first hardware contact,
  // r03).
  auto draw_ticket = [&]() {  // wave 0, all lanes: ONE add of 1 (the compiler folds the lanes' adds into one), lane 0's view
    const unsigned old = atomicAdd(g.ctl, lane == 0 ? 1u : 0u);
    s_i[0] = __builtin_amdgcn_readfirstlane((int)old);
  };
  if (wave == 0) draw_ticket();
  __syncthreads();
  for (;;) {
    const int t = __builtin_amdgcn_readfirstlane(s_i[0]);  // workgroup-uniform: keep the control flow scalar
    if (t >= g.ntasks) return;
    int I, J;
    ct_decode(t, g.nct, g.nrt, I, J);
    if (g.dbg && wave == 0) g.dbg[4 * (int64_t)t + 0] = wall_clock64();
    bool ok;
    if (I == J)
      ok = ct_diag_task<NW>(g, (ct_g_double*)g.A, (ct_g_double*)g.dinv16, (ct_g_double*)g.logdet, (ct_g_i32*)g.info, (ct_g_u32*)g.flags,
                            (ct_g_u32*)g.half, (ct_g_u32*)g.prog, (ct_g_u32*)g.ctl, (ct_g_u64*)g.dbg, J, t, (ct_lds_double*)lds, (ct_lds_int*)s_i);
    else
      ok = ct_offdiag_task<NW>(g, (ct_g_double*)g.A, (ct_g_double*)g.dinv16, (ct_g_double*)g.logdet, (ct_g_i32*)g.info, (ct_g_u32*)g.flags,
                               (ct_g_u32*)g.half, (ct_g_u32*)g.prog, (ct_g_u32*)g.ctl, (ct_g_u64*)g.dbg, I, J, t, (ct_lds_double*)lds, (ct_lds_int*)s_i);
    if (!__builtin_amdgcn_readfirstlane((int)ok)) return;
    if (wave == 0) draw_ticket();  // the next one (every thread passed the barrier of the publication: s_i[0] is free)
    __syncthreads();
  }
}

// NW = 4: 256 threads, two workgroups per compute unit; NW = 8: 512 threads, one per compute unit (every phase of the
// latency chain -- leaf, strip solve, last k-block of the diagonal tile -- has the whole compute unit)
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void chol_tiles_kernel(CholTilesArgs g) {
  chol_tiles_body<NW>(g);
}

}  // namespace gmb
