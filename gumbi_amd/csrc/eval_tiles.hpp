// eval_tiles.hpp -- the matrix work of ONE MAP-objective evaluation as ONE persistent launch: the tile Cholesky of
// chol_tiles.hpp, the triangular inverse and Sigma^-1, all as 128 x 128 tile tasks served from one ticket counter.
//
// Why: after round 3 the gradient was two thirds of a C2 evaluation (11.9 of 18.0 ms at N = 10k): a tree of ~35 GEMM
// launches for L^-1 (quantisation-bound: 56 TF/s) behind a factorisation whose first and last ~11 block columns are
// bound by the latency chain with the machine mostly idle.  Here the three phases share the chip at TILE granularity:
//
//   CHOL (I, J), I >= J   L tile of chol_tiles.hpp (left-looking contraction, leaf / strip solve, publication)
//   INV  (r, c), c >= r   U = L^-T BY ROWS:  U(r,c) = (delta_rc I - sum_{k=r}^{c-1} U(r,k) L(c,k)^T) L(c,c)^-T
//                         -- the forward substitution of block row r of the identity; the nct rows are INDEPENDENT
//                         chains (unlike the Cholesky there is no serial diagonal), and U(., c) only needs block row c of
//                         L, which is final as soon as the factorisation has passed column c: the inverse runs BESIDE
//                         the factorisation and fills its chain-bound ends;
//   ZZ   (I, J), I >= J   Sigma^-1(I,J) = sum_{k >= I} U(I,k) U(J,k)^T -- no dependencies among themselves, ticketed
//                         longest contraction first after everything else;
//   FIN  (r)              alpha of block row r from the parts its INV tasks left (and v, |v|^2): the last tickets.
//
// Both operands of every contraction are k-major, as everywhere in the engine (gemm_f64.hpp): U is kept by rows
// precisely so that Sigma^-1 = U U^T contracts over the slow index.  U's strictly upper tiles live in the (free) upper
// triangle of the factor buffer, its diagonal tiles in a side buffer (`udiag`): L stays intact, predict() after a
// gradient needs no restore.  INV tasks also leave the partial products U(r,c) v_c (v = L^-1 y, row N of the factor) so
// that alpha = U v = Sigma^-1 y costs one tiny fixed-order reduction afterwards.
//
// Tickets come from a host-built task list (kind, I, J) in a topological order of the dependency graph (CHOL column c,
// then INV column c - lag, ..., ZZ last): a task only ever waits for tasks with smaller tickets, which are finished or
// held by a running workgroup -- no deadlock whatever the residency of the workgroups; every wait is bounded by the
// abort word of chol_tiles.hpp.  Every tile is written once by its owner, every contraction runs in k order in one
// accumulator: same bits run to run.
//
// Replaces, per evaluation, the reverse-mode sweep through PyTensor's Cholesky op inside pm.find_MAP
// (gumbi/regression/pymc/GP.py:811).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "chol_tiles.hpp"
#include "covariance.hpp"
#include "gradient.hpp"

namespace gmb {

constexpr uint32_t ET_CHOL = 0u, ET_INV = 1u, ET_ZZ = 2u, ET_FIN = 3u;
__host__ __device__ inline uint32_t et_pack(uint32_t kind, int I, int J) { return (kind << 30) | ((uint32_t)I << 15) | (uint32_t)J; }

struct EvalTilesArgs {
  const uint32_t* tasks;  // et_pack(kind, I, J), in ticket order
  int32_t ntasks;
  uint32_t* uflags;       // nct x nct words (zeroed before the launch): U tile (r, c) is final when [r * nct + c] != 0
  double* udiag;          // nct tiles of 128 x 128 (leading dimension 128): the diagonal tiles U(r, r)
  double* Z;              // Sigma^-1 out: lower block triangle (diagonal tiles in full), column-major
  int64_t ldz;
  double* apart;          // [(r * nct + c) * 128 + i]: (U(r,c) v_c)_i -- alpha_r = sum_c of these, added up in c order by FIN (r)
  int32_t yb;             // block row of the factor buffer that holds row N (v = L^-1 y)
  // FIN (r), r = 0 .. nct-1, the last tickets: alpha of block row r once the row's INV tasks have left their parts
  // (rowdone[r] counts them; zeroed with the flags) and -- with_v: the factorisation is part of the launch -- v = row N of
  // the factor with |v|^2 in extract_v_kernel's summation order (vpart[r] = the chunk's sum; scal as in extract_v_kernel)
  uint32_t* rowdone;
  double* alpha;
  double* v;
  double* vpart;
  double* scal;
  int32_t with_v;
};

// One block row of a k-major operand: k-block kb lives at base + kb * 128 * ld, except k-block diag_kb, which lives in a
// side tile with leading dimension 128 (U's diagonal tiles).
struct EtOperand {
  const double* base;
  int64_t ld;
  const double* diag;
  int diag_kb;
};

// Arguments of a non-inlined task function arrive in vector registers: tell the compiler they are wave-uniform, so that
// loop control and address arithmetic stay on the scalar unit (no exec-mask branches inside the k loop).
__device__ __forceinline__ int et_uni(const int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T* et_uni_ptr(T* p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (T*)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int64_t et_uni64(const int64_t x) {
  const uint64_t v = (uint64_t)x;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// ONE wave: wait until k-block kb0 of both flag rows is final; returns the end of the run of final k-blocks that starts
// there (<= kb_end, at most 32 further) after one agent-scope acquire; -1 = the launch is being abandoned.
__device__ __forceinline__ int et_wait_rows(const CholTilesArgs& g, const uint32_t* pm, const uint32_t* pn, const int kb0, const int kb_end,
                                            const bool urgent) {
  const int lane = threadIdx.x & 63;
  const int idx = kb0 + (lane & 31);
  const bool mine = idx < kb_end;
  const uint32_t* p = (lane < 32 ? pn : pm) + (mine ? idx : kb0);
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
    if (urgent) __builtin_amdgcn_s_sleep(2);
    else __builtin_amdgcn_s_sleep(40);
  }
}

// out(128 x 128) = (NEG ? - : +) sum_{kb in [kb_lo, kb_hi)} N(kb) M(kb)^T: the k loop of ct_ksum (same staging, same pinned
// instruction order), accumulators starting at zero, operands addressed per k-block (EtOperand), cut into segments at the
// k-blocks whose tiles were not final yet.  Rows of `out` follow the n operand, columns the m operand.  Every thread of the
// workgroup calls it; false = the launch is being abandoned (uniform).
template <int NW, bool NEG>
__device__ __forceinline__ bool et_ksum(const CholTilesArgs& g, const EtOperand mop, const EtOperand nop, const uint32_t* mflags,
                                        const uint32_t* nflags, const int kb_lo, const int kb_hi, double* __restrict__ out, const int64_t ldo,
                                        double* __restrict__ lds, int* s_i) {
  constexpr int WGN = 2, WGM = NW / WGN;
  constexpr int WTM = TILE / (16 * WGM), WTN = TILE / (16 * WGN);
  constexpr int KT = ct_kt(NW);
  constexpr int PA = PITCH;
  constexpr int LA = TILE / 2, RA = 64 * NW / LA, NA = KT / RA;
  constexpr int KPB = TILE / KT;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int r16 = lane & 15, kq = lane >> 4;
  const int s_row = tid / LA, s_col = 2 * (tid % LA);

  d4 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
  d2 ra[NA], rb[NA];

  auto gload = [&](int kt) {
    const int kb = kt / KPB;
    const int krow = (kt - kb * KPB) * KT + s_row;
    const bool md = kb == mop.diag_kb, nd = kb == nop.diag_kb;
    const int64_t lda = md ? (int64_t)TILE : mop.ld, ldb = nd ? (int64_t)TILE : nop.ld;
    const double* pa = (md ? mop.diag + (int64_t)krow * TILE : mop.base + ((int64_t)kt * KT + s_row) * mop.ld) + s_col;
    const double* pb = (nd ? nop.diag + (int64_t)krow * TILE : nop.base + ((int64_t)kt * KT + s_row) * nop.ld) + s_col;
#pragma unroll
    for (int p = 0; p < NA; ++p) ra[p] = *reinterpret_cast<const d2*>(pa + (int64_t)(RA * p) * lda);
#pragma unroll
    for (int p = 0; p < NA; ++p) rb[p] = *reinterpret_cast<const d2*>(pb + (int64_t)(RA * p) * ldb);
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
        for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };

  const int kt_end = kb_hi * KPB;
  int ktc = kb_lo * KPB;
  while (ktc < kt_end) {
    const int kb = ktc / KPB;
    if (wave == 0) {
      int r = et_wait_rows(g, mflags, nflags, kb, kb_hi, kb + 1 >= kb_hi);
      if (r >= 0) r *= KPB;
      s_i[1] = r;
    }
    __syncthreads();
    const int kt1 = __builtin_amdgcn_readfirstlane(s_i[1]);
    if (kt1 < 0) return false;
    const int kt0 = ktc;
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
    __syncthreads();
    ktc = kt1;
  }

  // D layout of v_mfma_f64_16x16x4_f64: n = lane & 15, m = (lane >> 4) + 4 reg
  double* __restrict__ Cg = out + wn * (16 * WTN) + r16;
  const int64_t m0 = wm * (16 * WTM) + kq;
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double* row = Cg + (m0 + i * 16 + 4 * r) * ldo;
#pragma unroll
      for (int j = 0; j < WTN; ++j) row[j * 16] = NEG ? -acc[i][j][r] : acc[i][j][r];
    }
  return true;
}

// ---- the same contraction with LDS-DMA staging (eight-wave workgroups) --------------------------------------------------
// One eight-wave workgroup per compute unit has nobody else's MFMAs to fill its barrier bubbles with, and the register-staged
// loop above stops at ~88 % of the matrix pipe: every wave meets the same barrier at the end of a k-tile, then waits for its
// first fragments.  Here the operands go global -> LDS directly (global_load_lds_dwordx4: one wave instruction per 128-double
// k-row, 1 KiB contiguous, which is exactly a row of the padded [k][row] image), into a ring of FOUR 16-deep stages:
//
//   tile kt, steps k4 = 0 .. 3 (8 MFMAs per wave each; the fragments of step s + 1 are read while step s computes, across
//   tile boundaries too):
//       step 0, step 1, step 2 (reads the last fragments of tile kt)
//       s_waitcnt lgkmcnt(0) vmcnt(8) -- this wave's reads of tile kt have returned; its four DMAs of tile kt + 1 have landed
//                                        (those of kt + 2 and kt + 3 stay in flight)
//       s_barrier                     -- ... everybody's
//       DMA of tile kt + 4 into the stage of tile kt
//       step 3 (reads the first fragments of tile kt + 1)
//
// The one barrier per tile sits INSIDE the tile's MFMA stream (the wave arrives with eight MFMAs queued and its next
// fragments in registers), three tiles of DMA are in flight across it (raw s_barrier: __syncthreads() would drain them), a
// DMA has three tile times (~5 us) to land -- 96 KB in flight per compute unit, what 16 GB/s per compute unit needs at the
// loaded HBM latency -- and no staging registers or ds_writes are left in the loop.
// RAG: only the first ncol16 sixteen-column groups of the tile are contracted (the last block column of U: N - 128 c real
// columns; the others would multiply the padding rows of L's last block row and are zeroed afterwards anyway) -- the computed
// elements are the same sums in the same order.
template <bool NEG, bool RAG = false>
__device__ __forceinline__ bool et_ksum_dma(const CholTilesArgs& g, const EtOperand mop_in, const EtOperand nop_in, const uint32_t* mflags,
                                            const uint32_t* nflags, const int kb_lo_in, const int kb_hi_in, double* __restrict__ out,
                                            const int64_t ldo, ct_lds_double* l3, int* s_i, const int k_last_in = TILE, const int ncol16 = 8) {
  constexpr int WGN = 2, WGM = 4;
  constexpr int WTM = TILE / (16 * WGM), WTN = TILE / (16 * WGN);  // 2 x 4 MFMA tiles per wave
  constexpr int KT = 16, NS = 4;
  constexpr int PA = PITCH;
  constexpr int STAGE = KT * 2 * PA;  // doubles
  constexpr int KPB = TILE / KT;
  static_assert(NS * STAGE <= ct_lds_doubles(8), "the DMA ring must fit the workgroup's LDS");
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void g_void;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int r16 = lane & 15, kq = lane >> 4;
  const EtOperand mop{et_uni_ptr(mop_in.base), et_uni64(mop_in.ld), et_uni_ptr(mop_in.diag), et_uni(mop_in.diag_kb)};
  const EtOperand nop{et_uni_ptr(nop_in.base), et_uni64(nop_in.ld), et_uni_ptr(nop_in.diag), et_uni(nop_in.diag_kb)};
  const int kb_lo = et_uni(kb_lo_in), kb_hi = et_uni(kb_hi_in);

  d4 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

  // This wave's share of a tile: k-rows wave and wave + 8 of both operands.  The DMAs walk the contraction tile by tile, so the
  // source is a RUNNING pointer per operand (scalar registers; + 16 k-rows per tile), with one switch where an operand leaves
  // its diagonal side tile -- always the first k-block of the range.  A call past the end of the current segment (`real`
  // false) re-reads the operand's first tile instead: the ring then always has exactly two tiles in flight behind the one
  // being waited for, and the loop body needs no conditionals (the garbage lands in stages nobody reads before the segment's
  // closing wait).
  struct Run {
    const double* p;      // k-row `wave` of the next tile
    int64_t ld;           // doubles between k-rows there
    int diag_left;        // tiles left in the diagonal side tile (0: not in it)
    const double* after;  // k-row `wave` of the first tile behind the diagonal k-block
  };
  auto run_init = [&](const EtOperand& op) {
    Run r;
    if (op.diag_kb == kb_lo) {
      r.p = op.diag + (int64_t)wave * TILE;
      r.ld = TILE;
      r.diag_left = KPB;
      r.after = op.base + ((int64_t)(kb_lo + 1) * TILE + wave) * op.ld;
    } else {
      r.p = op.base + ((int64_t)kb_lo * TILE + wave) * op.ld;
      r.ld = op.ld;
      r.diag_left = 0;
      r.after = nullptr;
    }
    return r;
  };
  Run ra = run_init(mop), rb = run_init(nop);
  const double* const dummy_a = mop.base + (int64_t)wave * mop.ld;
  const double* const dummy_b = nop.base + (int64_t)wave * nop.ld;
  const int lane2 = 2 * lane;
  [[maybe_unused]] int fake_ctr = 0;
  auto dma = [&](const int stage, bool real) {
#ifdef GMB_ET_FAKE_LOADS  // tuning probe: every DMA re-reads the operand's first tile (cache-resident) -- timing only, garbage out
    real = false;
#endif
    ct_lds_double* As = l3 + stage * STAGE + wave * PA;
    ct_lds_double* Bs = As + KT * PA;
#if defined(GMB_ET_FAKE_LOADS) && GMB_ET_FAKE_LOADS == 2  // ... or walks the first k-block of block row 0 round and round (L2-resident, not L1)
    static_assert(KPB == 8, "");
    const double* pa = g.A + ((int64_t)(fake_ctr & 7) * KT + wave) * mop.ld;
    const double* pb = g.A + (int64_t)TILE + ((int64_t)(fake_ctr & 7) * KT + wave) * nop.ld;
    ++fake_ctr;
#else
    const double* pa = real ? ra.p : dummy_a;
    const double* pb = real ? rb.p : dummy_b;
#endif
    const int64_t lda = real ? ra.ld : mop.ld, ldb = real ? rb.ld : nop.ld;
    __builtin_amdgcn_global_load_lds((g_void*)(pa + lane2), (lds_void*)As, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((g_void*)(pb + lane2), (lds_void*)Bs, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((g_void*)(pa + 8 * lda + lane2), (lds_void*)(As + 8 * PA), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((g_void*)(pb + 8 * ldb + lane2), (lds_void*)(Bs + 8 * PA), 16, 0, 0);
    if (real) {
      ra.p += KT * ra.ld;
      rb.p += KT * rb.ld;
      if (ra.diag_left > 0 && --ra.diag_left == 0) {
        ra.p = ra.after;
        ra.ld = mop.ld;
      }
      if (rb.diag_left > 0 && --rb.diag_left == 0) {
        rb.p = rb.after;
        rb.ld = nop.ld;
      }
    }
  };
  double fa[2][WTM], fb[2][WTN];
  auto frags = [&](const int kt, const int k4, const int buf) {
    const ct_lds_double* As = l3 + (kt & (NS - 1)) * STAGE + (4 * k4 + kq) * PA + r16;
    const ct_lds_double* Bs = As + KT * PA;
#pragma unroll
    for (int i = 0; i < WTM; ++i) fa[buf][i] = As[wm * (16 * WTM) + i * 16];
#pragma unroll
    for (int j = 0; j < WTN; ++j) fb[buf][j] = Bs[wn * (16 * WTN) + j * 16];
  };
  const int ni = RAG ? max(0, min(WTM, et_uni(ncol16) - wm * WTM)) : WTM;  // live MFMA tile columns of this wave
  auto mfmas = [&](const int buf) {
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j) {
        if constexpr (RAG) {
          if (i < ni) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[buf][i], fb[buf][j], acc[i][j], 0, 0, 0);  // (wave-uniform)
        } else {
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[buf][i], fb[buf][j], acc[i][j], 0, 0, 0);
        }
      }
  };
  // one step's issue order: the next step's three fragment reads go out behind the first MFMAs of this one
  auto pin_step = [&]() {
    if constexpr (!RAG) {  // (the ragged form's MFMAs sit behind scalar branches: nothing to pin)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
      }
      __builtin_amdgcn_sched_group_barrier(0x008, WTM * WTN - 3, 0);
    }
  };

  // (k_last < 128: only the first k_last values of the contraction's LAST k-block are non-zero -- the tiles behind them are skipped)
  const int kt_end = kb_hi * KPB - (KPB - (et_uni(k_last_in) + KT - 1) / KT);
  int ktc = kb_lo * KPB;
  while (ktc < kt_end) {
    const int kb = ktc / KPB;
    if (wave == 0) {
      int r = et_wait_rows(g, mflags, nflags, kb, kb_hi, kb + 1 >= kb_hi);
      if (r >= 0) r *= KPB;
      s_i[1] = r;
    }
    __syncthreads();  // (no DMA in flight here)
    const int kt1r = __builtin_amdgcn_readfirstlane(s_i[1]);
    const int kt1 = kt1r < kt_end ? kt1r : kt_end;
    if (kt1 < 0) return false;
    const int kt0 = ktc;
    // prologue: four tiles on their way, the first one landed
    dma(kt0 & (NS - 1), true);
    dma((kt0 + 1) & (NS - 1), kt0 + 1 < kt1);
    dma((kt0 + 2) & (NS - 1), kt0 + 2 < kt1);
    dma((kt0 + 3) & (NS - 1), kt0 + 3 < kt1);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    frags(kt0, 0, 0);
    for (int kt = kt0; kt < kt1; ++kt) {
      frags(kt, 1, 1);
      mfmas(0);
      pin_step();
      frags(kt, 2, 0);
      mfmas(1);
      pin_step();
      frags(kt, 3, 1);
      mfmas(0);
      pin_step();
      // every read of tile kt has returned (this wave); its DMAs of tile kt + 1 have landed (kt + 2, kt + 3 stay in flight)
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // ... everybody's: tile kt + 1 may be read, the stage of tile kt may be overwritten
      dma(kt & (NS - 1), kt + 4 < kt1);
      frags(kt + 1, 0, 0);  // (behind the last tile of a segment: a stage of garbage, never used)
      mfmas(1);
      pin_step();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // the next segment's (or the next task's) first writes must not overtake slow waves' reads
    ktc = kt1;
  }

  double* __restrict__ Cg = out + wn * (16 * WTN) + r16;
  const int64_t m0 = wm * (16 * WTM) + kq;
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double* row = Cg + (m0 + i * 16 + 4 * r) * ldo;
#pragma unroll
      for (int j = 0; j < WTN; ++j) row[j * 16] = NEG ? -acc[i][j][r] : acc[i][j][r];
    }
  return true;
}

// ---- two tiles per task: out0 | out1 (128 x 256) = sum_k N(k) [M0(k) | M1(k)]^T -------------------------------------------------
// The n operand (the task's own, unshared row) is streamed ONCE for two output tiles: 0.75 x the operand bytes per flop of the
// 128 x 128 form, and every wave owns 4 x 4 MFMA tiles (64 x 64) -- 8 LDS fragment reads per 16 MFMAs instead of 6 per 8.  Same
// ring discipline as gemm_f64_dma_body: stages of EIGHT k-rows, three operand images per stage ([k][144] each: M0, M1, N), a ring of
// four, one barrier per stage inside the MFMA stream, three stages of DMA in flight across it; every wave DMAs k-row `wave` of the
// three operands.  Both tiles contract over the same k range [kb_lo, kb_hi) and wait for the flags of all three rows.
template <bool NEG>
__device__ __forceinline__ bool et_ksum_dma2(const CholTilesArgs& g, const EtOperand m0_in, const EtOperand m1_in, const EtOperand nop_in,
                                             const uint32_t* m0flags, const uint32_t* m1flags, const uint32_t* nflags, const int kb_lo_in,
                                             const int kb_hi_in, double* __restrict__ out0, double* __restrict__ out1, const int64_t ldo,
                                             ct_lds_double* l3, int* s_i, const int k_last_in = TILE) {
  constexpr int WGN = 2;                 // waves along n (rows); four along m: waves 0 .. 3 -> tile 0, 4 .. 7 -> tile 1
  constexpr int WTM = 4, WTN = 4;        // MFMA tiles per wave
  constexpr int KD = 8, NS = 4;
  constexpr int PA = PITCH;
  constexpr int STAGE = KD * 3 * PA;     // doubles
  constexpr int KPB = TILE / KD;         // stages per k-block
  static_assert(NS * STAGE <= ct_lds_doubles(8), "the DMA ring must fit the workgroup's LDS");
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void g_void;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;  // wm 0 .. 3: 64 columns of the 256; wn: 64 rows of the 128
  const int r16 = lane & 15, kq = lane >> 4;
  const EtOperand op[3] = {{et_uni_ptr(m0_in.base), et_uni64(m0_in.ld), et_uni_ptr(m0_in.diag), et_uni(m0_in.diag_kb)},
                           {et_uni_ptr(m1_in.base), et_uni64(m1_in.ld), et_uni_ptr(m1_in.diag), et_uni(m1_in.diag_kb)},
                           {et_uni_ptr(nop_in.base), et_uni64(nop_in.ld), et_uni_ptr(nop_in.diag), et_uni(nop_in.diag_kb)}};
  const int kb_lo = et_uni(kb_lo_in), kb_hi = et_uni(kb_hi_in);

  d4 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};

  // Running source pointers, one per operand image: k-row `wave` of the next stage and the distance to the stage behind it.  They
  // are set at the start of every SEGMENT (a run of stages whose operands are final and which does not leave a diagonal side
  // tile), so that inside the loop a DMA is three loads and ONE uniform condition (past the end of the segment the pointers stay
  // on its last stage: the ring keeps its count of outstanding loads, the garbage lands in stages nobody reads).
  const double* rp[3];
  int64_t rstep[3];
  int issued = 0, seg_end = 0;
  auto seg_init = [&](const int kt, const int kt1) {
    const int kb = kt / KPB;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      if (kb == op[q].diag_kb) {
        rp[q] = op[q].diag + ((int64_t)(kt - kb * KPB) * KD + wave) * TILE;
        rstep[q] = (int64_t)KD * TILE;
      } else {
        rp[q] = op[q].base + ((int64_t)kt * KD + wave) * op[q].ld;
        rstep[q] = (int64_t)KD * op[q].ld;
      }
    }
    issued = kt;
    seg_end = kt1;
  };
  const int lane2 = 2 * lane;
  auto dma = [&](const int stage) {
    ct_lds_double* S = l3 + stage * STAGE + wave * PA;
    __builtin_amdgcn_global_load_lds((g_void*)(rp[0] + lane2), (lds_void*)S, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((g_void*)(rp[1] + lane2), (lds_void*)(S + KD * PA), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((g_void*)(rp[2] + lane2), (lds_void*)(S + 2 * KD * PA), 16, 0, 0);
    if (issued + 1 < seg_end) {
      rp[0] += rstep[0];
      rp[1] += rstep[1];
      rp[2] += rstep[2];
      ++issued;
    }
  };
  double fa[2][WTM], fb[2][WTN];
  const int a_off = (wm >> 1) * KD * PA + (wm & 1) * (16 * WTM);  // image M0 or M1, 64-column half
  auto frags = [&](const int kt, const int k4, const int buf) {
    const ct_lds_double* S = l3 + (kt & (NS - 1)) * STAGE + (4 * k4 + kq) * PA + r16;
#pragma unroll
    for (int i = 0; i < WTM; ++i) fa[buf][i] = S[a_off + i * 16];
#pragma unroll
    for (int j = 0; j < WTN; ++j) fb[buf][j] = S[2 * KD * PA + wn * (16 * WTN) + j * 16];
  };
  auto mfmas = [&](const int buf) {
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[buf][i], fb[buf][j], acc[i][j], 0, 0, 0);
  };
  auto pin_step = [&]() {  // the next step's fragment reads go out behind the first MFMAs of this one
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
    }
    __builtin_amdgcn_sched_group_barrier(0x008, WTM * WTN - 8, 0);
  };

  const int kt_end = kb_hi * KPB - (KPB - (et_uni(k_last_in) + KD - 1) / KD);
  int ktc = kb_lo * KPB;
  while (ktc < kt_end) {
    const int kb = ktc / KPB;
    if (wave == 0) {
      int r = et_wait_rows(g, m0flags, nflags, kb, kb_hi, kb + 1 >= kb_hi);
      if (r >= 0) {
        const int r2 = et_wait_rows(g, m1flags, nflags, kb, r, kb + 1 >= kb_hi);
        r = r2 < 0 ? r2 : (r2 < r ? r2 : r);
      }
      if (r >= 0) r *= KPB;
      s_i[1] = r;
    }
    __syncthreads();  // (no DMA in flight here)
    const int kt1r = __builtin_amdgcn_readfirstlane(s_i[1]);
    if (kt1r < 0) return false;
    int kt1 = kt1r < kt_end ? kt1r : kt_end;
    // (a k-block that sits in a diagonal side tile is a segment of its own: inside a segment the pointers step uniformly)
    if ((kb == op[0].diag_kb || kb == op[1].diag_kb || kb == op[2].diag_kb) && kt1 > (kb + 1) * KPB) kt1 = (kb + 1) * KPB;
    const int kt0 = ktc;
    seg_init(kt0, kt1);
    dma(kt0 & (NS - 1));
    dma((kt0 + 1) & (NS - 1));
    dma((kt0 + 2) & (NS - 1));
    dma((kt0 + 3) & (NS - 1));
    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    frags(kt0, 0, 0);
    for (int kt = kt0; kt < kt1; ++kt) {
      frags(kt, 1, 1);
      mfmas(0);
      pin_step();
      // this wave's reads of stage kt are back, its DMAs of stage kt + 1 have landed (kt + 2, kt + 3 stay in flight)
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      dma(kt & (NS - 1));
      frags(kt + 1, 0, 0);  // (behind the last stage of a segment: garbage, never used)
      mfmas(1);
      pin_step();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ktc = kt1;
  }

  double* __restrict__ Cg = ((wm >> 1) ? out1 : out0) + wn * (16 * WTN) + r16;
  const int64_t m0 = (wm & 1) * (16 * WTM) + kq;
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double* row = Cg + (m0 + i * 16 + 4 * r) * ldo;
#pragma unroll
      for (int j = 0; j < WTN; ++j) row[j * 16] = NEG ? -acc[i][j][r] : acc[i][j][r];
    }
  return true;
}

// the contraction of a task: LDS-DMA staging for eight-wave workgroups (ET_DMA), the register-staged loop otherwise
#ifndef ET_DMA
#define ET_DMA 1
#endif
template <int NW, bool NEG>
__device__ __forceinline__ bool et_contract(const CholTilesArgs& g, const EtOperand mop, const EtOperand nop, const uint32_t* mflags,
                                            const uint32_t* nflags, const int kb_lo, const int kb_hi, double* __restrict__ out, const int64_t ldo,
                                            ct_lds_double* l3, int* s_i, const int k_last = TILE) {
  if constexpr (NW == 8 && ET_DMA) return et_ksum_dma<NEG>(g, mop, nop, mflags, nflags, kb_lo, kb_hi, out, ldo, l3, s_i, k_last);
  else return et_ksum<NW, NEG>(g, mop, nop, mflags, nflags, kb_lo, kb_hi, out, ldo, (double*)l3, s_i);  // (the tuning build's
  // four-wave form: not cut -- the skipped tiles are zeros, the result is the same; with the cut this compiler's optimiser crashes)
}

// INV task (r, c): tile (r, c) of U = L^-T.  Contraction over the tiles (r, k) of its own row and block row c of L, strip solve
// against L(c, c), publication for the later tiles of the row and for the ZZ tasks; then the partial product with v_c.
template <int NW>
__device__ __noinline__ bool et_inv_task(const CholTilesArgs g_in, ct_g_double* A, ct_g_double* dinv16, ct_g_double* udiag, ct_g_double* apart,
                                         ct_g_u32* flags, ct_g_u32* uflags, ct_g_u32* rowdone, ct_g_u32* ctl, ct_g_u64* dbg, const int r, const int c,
                                         const int t, const int yb, ct_lds_double* l3, ct_lds_int* s3) {
  const CholTilesArgs g = ct_rebuild(g_in, A, dinv16, nullptr, nullptr, flags, nullptr, ctl, dbg);
  const uint32_t* uf = (const uint32_t*)uflags;
  double* ud = (double*)udiag;
  int* s_i = (int*)s3;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  TrsmArgs ta;
  ta.B = c == r ? ud + (int64_t)r * TILE * TILE : g.A + (int64_t)r * TILE + (int64_t)c * TILE * g.ld;
  ta.ldb = c == r ? (int64_t)TILE : g.ld;
  ta.nrows = TILE;
  ta.L = g.A + (int64_t)c * TILE * (g.ld + 1);
  ta.ldl = g.ld;
  ta.dinv16 = g.dinv16 + (int64_t)c * 8 * 256;
  ta.nvalid = (int)(g.N - (int64_t)c * TILE < TILE ? g.N - (int64_t)c * TILE : TILE);
  strip_d4 X0[8], X1[8];
  if (c > r) {
    const EtOperand mop{g.A + (int64_t)c * TILE, g.ld, nullptr, -1};                       // block row c of L
    const EtOperand nop{g.A + (int64_t)r * TILE, g.ld, ud + (int64_t)r * TILE * TILE, r};  // block row r of U
    // (the last block column: N - 128 c real columns of the tile, the rest is padding that is zeroed below)
    const int ncol16 = ta.nvalid >= TILE ? 8 : (ta.nvalid + 15) / 16;
    bool ok;
    if (NW == 8 && ET_DMA && CT_RAGGED && ncol16 < 8)
      ok = et_ksum_dma<true, true>(g, mop, nop, g.flags + (int64_t)c * g.nct, uf + (int64_t)r * g.nct, r, c, ta.B, ta.ldb, l3, s_i, TILE, ncol16);
    else
      ok = et_contract<NW, true>(g, mop, nop, g.flags + (int64_t)c * g.nct, uf + (int64_t)r * g.nct, r, c, ta.B, ta.ldb, l3, s_i);
    if (!__builtin_amdgcn_readfirstlane((int)ok)) return false;
    if (g.dbg && wave == 0) g.dbg[4 * (int64_t)t + 1] = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // own epilogue stores are read back by other lanes
    __syncthreads();
    trsm_strip_load(ta, 16 * wave, X0);
    if constexpr (NW == 4) trsm_strip_load(ta, 16 * (wave + 4), X1);
  } else {
    // block row r of the identity
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        X0[s][q] = (16 * wave + r16 == 16 * s + kq + 4 * q) ? 1.0 : 0.0;
        if constexpr (NW == 4) X1[s][q] = (16 * (wave + 4) + r16 == 16 * s + kq + 4 * q) ? 1.0 : 0.0;
      }
  }
  // L(c, c)
  if (wave == 0) s_i[1] = ct_wait_one(g, g.flags + (int64_t)c * g.nct + c, false);
  __syncthreads();
  if (__builtin_amdgcn_readfirstlane(s_i[1]) < 0) return false;
  if (g.dbg && wave == 0) g.dbg[4 * (int64_t)t + 2] = wall_clock64();
  if constexpr (NW == 8 && CT_STRIP_LDS) {
    ct_strip_solve_lds<true>(ta, 16 * wave, X0, l3);
  } else {
    trsm_strip_solve_store_pf<true>(ta, 16 * wave, X0);
    if constexpr (NW == 4) trsm_strip_solve_store_pf<true>(ta, 16 * (wave + 4), X1);
  }
  if (ta.nvalid < TILE) {
    // last block column of a ragged matrix: columns >= nvalid went through the solve as identity padding and carry the y row's
    // products -- Sigma^-1 = U U^T must not see them
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = 16 * s + kq + 4 * q;
        if (col >= ta.nvalid) {
          X0[s][q] = 0.0;
          __hip_atomic_store(&ta.B[16 * wave + r16 + (int64_t)col * ta.ldb], 0.0, CT_RLX_AGENT);
          if constexpr (NW == 4) {
            X1[s][q] = 0.0;
            __hip_atomic_store(&ta.B[16 * (wave + 4) + r16 + (int64_t)col * ta.ldb], 0.0, CT_RLX_AGENT);
          }
        }
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (wave == 0) {
    __hip_atomic_store(uflags + (int64_t)r * g.nct + c, 1u, CT_RLX_AGENT);
    if (g.dbg) g.dbg[4 * (int64_t)t + 3] = wall_clock64();
  }
  // (U(r,c) v_c)_i for the rows of this wave, in a fixed order: 32 columns per lane, then the four lane groups.  Row N of the
  // factor sits in block row yb: the LAST tile of column c in ticket order -- waited for here, behind the publication, so
  // that it never holds up the row's chain.
  if (yb != c) {
    if (wave == 0) s_i[1] = ct_wait_one(g, g.flags + (int64_t)yb * g.nct + c, false);
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(s_i[1]) < 0) return false;
  }
  {
    const double* vrow = g.A + g.N + (int64_t)c * TILE * g.ld;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = 16 * s + kq + 4 * q;
        const double vv = col < ta.nvalid ? vrow[(int64_t)col * g.ld] : 0.0;
        s0 = fma(X0[s][q], vv, s0);
        if constexpr (NW == 4) s1 = fma(X1[s][q], vv, s1);
      }
    s0 += __shfl_xor(s0, 16);
    s0 += __shfl_xor(s0, 32);
    // (write-through: FIN (r) reads them on another compute unit after the row's count)
    double* ap = (double*)apart + ((int64_t)r * g.nct + c) * TILE;
    if (kq == 0) __hip_atomic_store(&ap[16 * wave + r16], s0, CT_RLX_AGENT);
    if constexpr (NW == 4) {
      s1 += __shfl_xor(s1, 16);
      s1 += __shfl_xor(s1, 32);
      if (kq == 0) __hip_atomic_store(&ap[16 * (wave + 4) + r16], s1, CT_RLX_AGENT);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (wave == 0 && lane == 0) __hip_atomic_fetch_add((uint32_t*)rowdone + r, 1u, CT_RLX_AGENT);
  return true;
}

// FIN task (r): alpha_i = sum_{c >= r} apart[(r * nct + c) * 128 + i] in c order for the rows of block row r; with_v: v_i = row N
// of the factor and the chunk's share of |v|^2 (the LAST of these tasks adds the chunks up, see eval_finish's order in
// covariance.hpp: extract_v_kernel).
template <int NW>
__device__ __noinline__ bool et_fin_task(const CholTilesArgs g_in, ct_g_double* A, ct_g_double* apart, ct_g_double* alpha_out, ct_g_double* v_out,
                                         ct_g_double* vpart, ct_g_double* scal, ct_g_u32* rowdone, ct_g_u32* ctl, ct_g_u64* dbg, const int r,
                                         const int t, const int with_v, const int nfin, ct_lds_double* l3, ct_lds_int* s3) {
  const CholTilesArgs g = ct_rebuild(g_in, A, nullptr, nullptr, nullptr, nullptr, nullptr, ctl, dbg);
  int* s_i = (int*)s3;
  double* red = (double*)l3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t* rd = (const uint32_t*)rowdone + r;
  if (wave == 0) s_i[1] = ct_wait_two(g, rd, rd, (uint32_t)(g.nct - r), true);
  __syncthreads();
  if (__builtin_amdgcn_readfirstlane(s_i[1]) < 0) return false;
  if (g.dbg && wave == 0) g.dbg[4 * (int64_t)t + 1] = g.dbg[4 * (int64_t)t + 2] = wall_clock64();
  double x = 0.0;
  if (tid < TILE) {
    const double* ap = (const double*)apart + (int64_t)r * g.nct * TILE + tid;
    double s = 0.0;
    for (int c = r; c < g.nct; ++c) s += ap[(int64_t)c * TILE];
    const int64_t row = (int64_t)r * TILE + tid;
    if (row < g.N) {
      ((double*)alpha_out)[row] = s;
      if (with_v) {
        x = g.A[g.N + row * g.ld];
        ((double*)v_out)[row] = x;
      }
    }
  }
  if (with_v) {
    // the chunk's sum as v_chunk_sum forms it in a 128-thread workgroup: waves 0 and 1 hold the values
    double acc = x * x;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (lane == 0 && wave < 2) red[wave] = acc;
    __syncthreads();
    if (tid == 0) {
      double* vp = (double*)vpart;
      double* sc = (double*)scal;
      atomicExch((unsigned long long*)&vp[r], (unsigned long long)__double_as_longlong(red[0] + red[1]));
      __threadfence();
      if (atomicAdd((unsigned int*)&sc[40], 1u) == (unsigned)nfin - 1u) {
        __threadfence();
        double tot = 0.0;
        for (unsigned b = 0; b < (unsigned)EXTRACT_V_BLOCKS; ++b) {
          double p = 0.0;
          for (unsigned c = b; c < (unsigned)nfin; c += EXTRACT_V_BLOCKS)
            p += __longlong_as_double((long long)atomicAdd((unsigned long long*)&vp[c], 0ull));  // read at the L2
          tot += p;
        }
        sc[0] = tot;
      }
    }
  }
  if (g.dbg && wave == 0) g.dbg[4 * (int64_t)t + 3] = wall_clock64();
  return true;
}

// ZZ task (I, J), I >= J: tile (I, J) of Sigma^-1 = U U^T, contraction over k >= I.
template <int NW>
__device__ __noinline__ bool et_zz_task(const CholTilesArgs g_in, ct_g_double* A, ct_g_double* udiag, ct_g_double* Z, const int64_t ldz,
                                        ct_g_u32* uflags, ct_g_u32* ctl, ct_g_u64* dbg, const int I, const int J, const int t, ct_lds_double* l3,
                                        ct_lds_int* s3) {
  const CholTilesArgs g = ct_rebuild(g_in, A, nullptr, nullptr, nullptr, nullptr, nullptr, ctl, dbg);
  const uint32_t* uf = (const uint32_t*)uflags;
  const double* ud = (const double*)udiag;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const EtOperand mop{g.A + (int64_t)J * TILE, g.ld, ud + (int64_t)J * TILE * TILE, J};
  const EtOperand nop{g.A + (int64_t)I * TILE, g.ld, ud + (int64_t)I * TILE * TILE, I};
  double* out = (double*)Z + (int64_t)I * TILE + (int64_t)J * TILE * ldz;
  // the last block column of U holds zeros from column N on (et_inv_task): a ragged matrix's last k-block is cut there
  const int k_last = (int)(g.N - (int64_t)(g.nct - 1) * TILE);
  if (!et_contract<NW, false>(g, mop, nop, uf + (int64_t)J * g.nct, uf + (int64_t)I * g.nct, I, g.nct, out, ldz, l3, (int*)s3, k_last)) return false;
  if (g.dbg && wave == 0) {
    const unsigned long long now = wall_clock64();
    g.dbg[4 * (int64_t)t + 1] = now;
    g.dbg[4 * (int64_t)t + 2] = now;
    g.dbg[4 * (int64_t)t + 3] = now;
  }
  return true;
}

// ---- the same two-tile contraction with NO LDS and NO barrier in the loop: operands straight from L1 / L2 into MFMA fragments ------
// Both operands are k-major, so a lane's fragment values are already contiguous pairs in memory if the wave's sixteen-wide MFMA
// tiles take the EVEN and the ODD columns (rows) of a 32-wide group: lane (r16, kq) loads the two doubles 2 r16, 2 r16 + 1 of k-row
// 4 s + kq with one 16-byte load and feeds the first to tile 2a, the second to tile 2a + 1 -- four loads per sixteen MFMAs, each wave
// on its own (nothing is shared through LDS, nothing waits for the other seven waves); loads are issued PF steps ahead.
template <bool NEG>
__device__ __forceinline__ bool et_ksum_reg2(const CholTilesArgs& g, const EtOperand m0_in, const EtOperand m1_in, const EtOperand nop_in,
                                             const uint32_t* m0flags, const uint32_t* m1flags, const uint32_t* nflags, const int kb_lo_in,
                                             const int kb_hi_in, double* __restrict__ out0, double* __restrict__ out1, const int64_t ldo,
                                             int* s_i, const int k_last_in = TILE) {
  constexpr int WGN = 2;
  constexpr int PF = 4;                 // k4-steps of loads in flight per wave
  constexpr int SPB = TILE / 4;         // k4-steps per k-block
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int r16 = lane & 15, kq = lane >> 4;
  const EtOperand mop = (wm >> 1) ? EtOperand{et_uni_ptr(m1_in.base), et_uni64(m1_in.ld), et_uni_ptr(m1_in.diag), et_uni(m1_in.diag_kb)}
                                  : EtOperand{et_uni_ptr(m0_in.base), et_uni64(m0_in.ld), et_uni_ptr(m0_in.diag), et_uni(m0_in.diag_kb)};
  const EtOperand nop{et_uni_ptr(nop_in.base), et_uni64(nop_in.ld), et_uni_ptr(nop_in.diag), et_uni(nop_in.diag_kb)};
  const int d0 = et_uni(m0_in.diag_kb), d1 = et_uni(m1_in.diag_kb);
  const int kb_lo = et_uni(kb_lo_in), kb_hi = et_uni(kb_hi_in);
  const int acol = (wm & 1) * 64 + 2 * r16, brow = wn * 64 + 2 * r16;  // this lane's pair inside the first 32-wide group

  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
  d2 fa[PF][2], fb[PF][2];
  const double* pa = nullptr;
  const double* pb = nullptr;
  int64_t sa = 0, sb = 0;  // doubles between k4-steps
  int left = 0;            // steps of the segment behind the one the pointers stand at
  auto load = [&](const int slot) {
    fa[slot][0] = *reinterpret_cast<const d2*>(pa);
    fa[slot][1] = *reinterpret_cast<const d2*>(pa + 32);
    fb[slot][0] = *reinterpret_cast<const d2*>(pb);
    fb[slot][1] = *reinterpret_cast<const d2*>(pb + 32);
    // (past the end of the segment the pointers stay on its last step: loads that run ahead never leave the operands)
    const bool more = left > 0;
    pa += more ? sa : 0;
    pb += more ? sb : 0;
    left -= more ? 1 : 0;
  };
  auto mfmas = [&](const int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[slot][i >> 1][i & 1], fb[slot][j >> 1][j & 1], acc[i][j], 0, 0, 0);
  };

  const int st_end = kb_hi * SPB - (SPB - (et_uni(k_last_in) + 3) / 4);
  int stc = kb_lo * SPB;
  while (stc < st_end) {
    const int kb = stc / SPB;
    if (wave == 0) {
      int r = et_wait_rows(g, m0flags, nflags, kb, kb_hi, kb + 1 >= kb_hi);
      if (r >= 0) {
        const int r2 = et_wait_rows(g, m1flags, nflags, kb, r, kb + 1 >= kb_hi);
        r = r2 < 0 ? r2 : (r2 < r ? r2 : r);
      }
      if (r >= 0) r *= SPB;
      s_i[1] = r;
    }
    __syncthreads();
    const int st1r = __builtin_amdgcn_readfirstlane(s_i[1]);
    if (st1r < 0) return false;
    int st1 = st1r < st_end ? st1r : st_end;
    if ((kb == d0 || kb == d1 || kb == nop.diag_kb) && st1 > (kb + 1) * SPB) st1 = (kb + 1) * SPB;
    // this segment's running pointers (k-row 4 s + kq of the wave's operand images)
    if (kb == mop.diag_kb) {
      pa = mop.diag + ((int64_t)(stc - kb * SPB) * 4 + kq) * TILE + acol;
      sa = 4 * TILE;
    } else {
      pa = mop.base + ((int64_t)stc * 4 + kq) * mop.ld + acol;
      sa = 4 * mop.ld;
    }
    if (kb == nop.diag_kb) {
      pb = nop.diag + ((int64_t)(stc - kb * SPB) * 4 + kq) * TILE + brow;
      sb = 4 * TILE;
    } else {
      pb = nop.base + ((int64_t)stc * 4 + kq) * nop.ld + brow;
      sb = 4 * nop.ld;
    }
    const int nst = st1 - stc;  // (a multiple of 4 except in the cut last k-block)
    int done = 0;
    left = nst - 1;
    // prologue: PF steps on their way
#pragma unroll
    for (int q = 0; q < PF; ++q) load(q);
    for (; done + PF <= nst; done += PF) {
#pragma unroll
      for (int q = 0; q < PF; ++q) {
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // the oldest step's four loads are back (3 x 4 stay in flight)
        mfmas(q);
        load(q);
#pragma unroll
        for (int z = 0; z < 4; ++z) {
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);  // MFMA
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
    }
    for (int q = 0; done < nst; ++done, ++q) {  // ragged tail (the cut last k-block): step by step
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (q == 0) mfmas(0);
      else if (q == 1) mfmas(1);
      else if (q == 2) mfmas(2);
      else mfmas(3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stc = st1;
  }

  // D layout: n = lane & 15 -> row pair 2 r16 (+ j & 1), m = kq + 4 reg -> column 2 (kq + 4 reg) (+ i & 1) of the 32-wide groups
  double* __restrict__ Cg = ((wm >> 1) ? out1 : out0) + wn * 64 + 2 * r16;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t col = (wm & 1) * 64 + (i >> 1) * 32 + 2 * (kq + 4 * r) + (i & 1);
      double* row = Cg + col * ldo;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        d2 v{acc[i][2 * jj][r], acc[i][2 * jj + 1][r]};
        if (NEG) v = -v;
        *reinterpret_cast<d2*>(row + 32 * jj) = v;
      }
    }
  return true;
}

// ZZ pair task (I; J, J + 1), J + 1 <= I: the tiles (I, J) and (I, J + 1) of Sigma^-1 in one pass over block row I of U.
constexpr uint32_t ET_PAIR = 0x4000u;  // flag in the J field of a ZZ task word
template <int NW>
__device__ __noinline__ bool et_zz2_task(const CholTilesArgs g_in, ct_g_double* A, ct_g_double* udiag, ct_g_double* Z, const int64_t ldz,
                                         ct_g_u32* uflags, ct_g_u32* ctl, ct_g_u64* dbg, const int I, const int J, const int t, ct_lds_double* l3,
                                         ct_lds_int* s3) {
  const CholTilesArgs g = ct_rebuild(g_in, A, nullptr, nullptr, nullptr, nullptr, nullptr, ctl, dbg);
  const uint32_t* uf = (const uint32_t*)uflags;
  const double* ud = (const double*)udiag;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const EtOperand m0{g.A + (int64_t)J * TILE, g.ld, ud + (int64_t)J * TILE * TILE, J};
  const EtOperand m1{g.A + (int64_t)(J + 1) * TILE, g.ld, ud + (int64_t)(J + 1) * TILE * TILE, J + 1};
  const EtOperand nop{g.A + (int64_t)I * TILE, g.ld, ud + (int64_t)I * TILE * TILE, I};
  double* out0 = (double*)Z + (int64_t)I * TILE + (int64_t)J * TILE * ldz;
  const int k_last = (int)(g.N - (int64_t)(g.nct - 1) * TILE);
  static_assert(NW == 8 || NW == 4, "");
  if constexpr (NW != 8) {
    // no pair contraction for four-wave workgroups (the host never lists pair tasks for them): give the launch up loudly
    // rather than leave two tiles of Sigma^-1 unwritten
    if (wave == 0) __hip_atomic_store(g.ctl + 1, 1u, CT_RLX_AGENT);
    return false;
  }
  if constexpr (NW == 8) {
#ifdef ET_PAIR_REG
    if (!et_ksum_reg2<false>(g, m0, m1, nop, uf + (int64_t)J * g.nct, uf + (int64_t)(J + 1) * g.nct, uf + (int64_t)I * g.nct, I, g.nct, out0,
                             out0 + (int64_t)TILE * ldz, ldz, (int*)s3, k_last))
      return false;
#else
    if (!et_ksum_dma2<false>(g, m0, m1, nop, uf + (int64_t)J * g.nct, uf + (int64_t)(J + 1) * g.nct, uf + (int64_t)I * g.nct, I, g.nct, out0,
                             out0 + (int64_t)TILE * ldz, ldz, l3, (int*)s3, k_last))
      return false;
#endif
  }
  if (g.dbg && wave == 0) {
    const unsigned long long now = wall_clock64();
    g.dbg[4 * (int64_t)t + 1] = now;
    g.dbg[4 * (int64_t)t + 2] = now;
    g.dbg[4 * (int64_t)t + 3] = now;
  }
  return true;
}

// The persistent loop: tickets index the host-built task list.
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void eval_tiles_kernel(CholTilesArgs g, EvalTilesArgs x) {
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
    if (t >= x.ntasks) return;
    const uint32_t w = __builtin_amdgcn_readfirstlane(x.tasks[t]);
    const uint32_t kind = w >> 30;
    const int I = (int)((w >> 15) & 0x7fffu), J = (int)(w & 0x7fffu);
    if (g.dbg && wave == 0) g.dbg[4 * (int64_t)t + 0] = wall_clock64();
    bool ok;
    if (kind == ET_CHOL) {
      if (I == J)
        ok = ct_diag_task<NW>(g, (ct_g_double*)g.A, (ct_g_double*)g.dinv16, (ct_g_double*)g.logdet, (ct_g_i32*)g.info, (ct_g_u32*)g.flags,
                              (ct_g_u32*)g.half, (ct_g_u32*)g.prog, (ct_g_u32*)g.ctl, (ct_g_u64*)g.dbg, J, t, (ct_lds_double*)lds, (ct_lds_int*)s_i);
      else
        ok = ct_offdiag_task<NW>(g, (ct_g_double*)g.A, (ct_g_double*)g.dinv16, (ct_g_double*)g.logdet, (ct_g_i32*)g.info, (ct_g_u32*)g.flags,
                                 (ct_g_u32*)g.half, (ct_g_u32*)g.prog, (ct_g_u32*)g.ctl, (ct_g_u64*)g.dbg, I, J, t, (ct_lds_double*)lds,
                                 (ct_lds_int*)s_i);
    } else if (kind == ET_INV) {
      ok = et_inv_task<NW>(g, (ct_g_double*)g.A, (ct_g_double*)g.dinv16, (ct_g_double*)x.udiag, (ct_g_double*)x.apart, (ct_g_u32*)g.flags,
                           (ct_g_u32*)x.uflags, (ct_g_u32*)x.rowdone, (ct_g_u32*)g.ctl, (ct_g_u64*)g.dbg, I, J, t, x.yb, (ct_lds_double*)lds,
                           (ct_lds_int*)s_i);
    } else if (kind == ET_ZZ && (J & (int)ET_PAIR)) {
      ok = et_zz2_task<NW>(g, (ct_g_double*)g.A, (ct_g_double*)x.udiag, (ct_g_double*)x.Z, x.ldz, (ct_g_u32*)x.uflags, (ct_g_u32*)g.ctl,
                           (ct_g_u64*)g.dbg, I, J & ~(int)ET_PAIR, t, (ct_lds_double*)lds, (ct_lds_int*)s_i);
    } else if (kind == ET_ZZ) {
      ok = et_zz_task<NW>(g, (ct_g_double*)g.A, (ct_g_double*)x.udiag, (ct_g_double*)x.Z, x.ldz, (ct_g_u32*)x.uflags, (ct_g_u32*)g.ctl,
                          (ct_g_u64*)g.dbg, I, J, t, (ct_lds_double*)lds, (ct_lds_int*)s_i);
    } else {
      ok = et_fin_task<NW>(g, (ct_g_double*)g.A, (ct_g_double*)x.apart, (ct_g_double*)x.alpha, (ct_g_double*)x.v, (ct_g_double*)x.vpart,
                           (ct_g_double*)x.scal, (ct_g_u32*)x.rowdone, (ct_g_u32*)g.ctl, (ct_g_u64*)g.dbg, I, t, x.with_v, g.nct,
                           (ct_lds_double*)lds, (ct_lds_int*)s_i);
    }
    if (!__builtin_amdgcn_readfirstlane((int)ok)) return;
    __syncthreads();  // every thread is done with s_i and the staging buffers of the task
    if (wave == 0) draw_ticket();
    __syncthreads();
  }
}

// What the host wants from one evaluation, gathered by ONE small launch into pinned host memory the device writes directly
// (instead of four device-to-host copies, one of them into pageable memory): log-det and |v|^2, the failure index, the tile
// launch's abort word, and the used head of every term's gradient accumulator region.
struct EvalLandArgs {
  const double* scal;    // [0] log-det, [1] |v|^2
  const int32_t* info;
  const uint32_t* abort; // may be null
  const double* gacc;    // accumulator regions, `region` doubles apart
  int32_t nterms, region;
  int32_t used[8];       // doubles to take from the head of each region
  double* out_scal;      // host: 2 doubles
  int32_t* out_info;
  uint32_t* out_abort;
  double* out_gacc;      // host: regions at the same offsets
};
__global__ __launch_bounds__(256) void eval_land_kernel(EvalLandArgs a) {
  const int t = threadIdx.x;
  if (t < 2) a.out_scal[t] = a.scal[t];
  if (t == 2) *a.out_info = *a.info;
  if (t == 3) *a.out_abort = a.abort ? __hip_atomic_load(a.abort, CT_RLX_AGENT) : 0u;
  for (int q = 0; q < a.nterms; ++q)
    for (int i = t; i < a.used[q]; i += 256) a.out_gacc[(int64_t)q * a.region + i] = a.gacc[(int64_t)q * a.region + i];
}

// The tail of a small evaluation as ONE launch of one workgroup (gmb_evaluate's light path, one covariance term, ARD, no
// table above 8 levels): grad_sum_partials_kernel's fixed-order sums over the partial vectors (four slots at a time, one per
// 256-thread group: same order of additions, same bits), grad_diag_kernel's diagonal terms, and eval_land_kernel's copy to the
// pinned landing -- three launches of ~4 us each otherwise.
struct GradFinishArgs {
  const double* part;  // grad_sum_partials_kernel's arguments
  int32_t nparts, nslots;
  int64_t stride;
  GradRanges r;
  double* acc;
  const double* Z;     // grad_diag_kernel's (shard 0 of 1, Z not packed)
  int64_t ldz;
  const double* alpha;
  PointSet pts;
  CovParams p;
  double sigma;
  double* diag_out;
  EvalLandArgs land;
};
__global__ __launch_bounds__(1024) void grad_finish_kernel(GradFinishArgs a) {
  __shared__ double red4[4][256];
  __shared__ double red[16];
  __shared__ double tab[16][32];
  const int t = threadIdx.x, grp = t >> 8, lt = t & 255;
  for (int q0 = 0; q0 < a.nslots; q0 += 4) {
    const int q = q0 + grp;
    double s = 0.0;
    if (q < a.nslots)
      for (int b = lt; b < a.nparts; b += 256) s += a.part[(int64_t)b * a.stride + q];
    red4[grp][lt] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
      if (lt < h) red4[grp][lt] += red4[grp][lt + h];
      __syncthreads();
    }
    if (lt == 0 && q < a.nslots)
      for (int i = 0; i < a.r.n; ++i)
        if (q >= a.r.dense[i] && q < a.r.dense[i] + a.r.count[i]) {
          a.acc[a.r.dst[i] + (q - a.r.dense[i])] = red4[grp][0];
          break;
        }
    __syncthreads();
  }
  // diagonal-only terms (grad_diag_kernel)
  const int wave = t >> 6;
  for (int idx = t; idx < 16 * 32; idx += 1024) (&tab[0][0])[idx] = 0.0;
  __syncthreads();
  double gs = 0.0;
  for (int64_t i = t; i < a.pts.n; i += 1024) {
    const double al = a.alpha[i];
    const double m = 0.5 * (a.Z[i + i * a.ldz] - al * al);
    double mult = 1.0;
    if (a.p.noise_tab >= 0) {
      const int c = a.pts.cat[(int64_t)a.p.noise_tab * a.pts.npad + i];
      mult = a.p.noise_mult[c];
      atomicAdd(&tab[wave][c], m * a.sigma * a.sigma);
    }
    gs = fma(m, 2.0 * a.sigma * mult, gs);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) gs += __shfl_down(gs, off);
  if ((t & 63) == 0) red[wave] = gs;
  __syncthreads();
  if (t == 0) {
    double s = 0.0;
    for (int w = 0; w < 16; ++w) s += red[w];
    a.diag_out[0] = s;
  }
  if (a.p.noise_tab >= 0 && t < 32) {
    double s = 0.0;
    for (int w = 0; w < 16; ++w) s += tab[w][t];
    a.diag_out[1 + t] = s;
  }
  // everything this workgroup wrote to the accumulators is visible to all of it, then out to the host's landing
  __threadfence();
  __syncthreads();
  if (t < 2) a.land.out_scal[t] = a.land.scal[t];
  if (t == 2) *a.land.out_info = *a.land.info;
  if (t == 3) *a.land.out_abort = a.land.abort ? __hip_atomic_load(a.land.abort, CT_RLX_AGENT) : 0u;
  for (int i = t; i < a.land.used[0]; i += 1024)
    a.land.out_gacc[i] = __hip_atomic_load(&a.land.gacc[i], CT_RLX_AGENT);  // (past this compute unit's L1)
}

// Host side: the task list.  `with_chol`: the factorisation's tile tasks are part of the launch (column c's tasks, then the
// INV tasks of column c - lag); otherwise the factor is final and only INV / ZZ / FIN tasks are listed.  INV column c: r = 0 .. c
// (longest contraction first); ZZ: block rows I ascending (longest first), J = 0 .. I inside; FIN (r), r = 0 .. nct-1, last.
inline void et_build_tasks(int nct, int nrt, bool with_chol, int lag, std::vector<uint32_t>& out, bool zz_pairs = false) {
  out.clear();
  auto inv_col = [&](int c) {
    for (int r = 0; r <= c; ++r) out.push_back(et_pack(ET_INV, r, c));
  };
  int next_inv = 0;
  if (with_chol) {
    for (int c = 0; c < nct; ++c) {
      for (int I = c; I < nrt; ++I) out.push_back(et_pack(ET_CHOL, I, c));
      if (c - lag >= 0) {
        inv_col(c - lag);
        next_inv = c - lag + 1;
      }
    }
  }
  for (int c = next_inv; c < nct; ++c) inv_col(c);
  for (int I = 0; I < nct; ++I)
    for (int J = 0; J <= I; ++J) {
      if (zz_pairs && J + 1 <= I) {  // (I; J, J + 1): one pass over block row I for two tiles
        out.push_back(et_pack(ET_ZZ, I, J | (int)ET_PAIR));
        ++J;
      } else {
        out.push_back(et_pack(ET_ZZ, I, J));
      }
    }
  for (int r = 0; r < nct; ++r) out.push_back(et_pack(ET_FIN, r, 0));
}

}  // namespace gmb
