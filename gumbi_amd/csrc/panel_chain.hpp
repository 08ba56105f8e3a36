// panel_chain.hpp -- the latency-bound chain of one Cholesky panel as ONE cooperative kernel.
//
// Why.  A panel of 8 block columns is ~24 dependent small kernels (leaf, strip solve, in-panel
// updates).  Launched one by one beside the bulk trailing update they starve: a resident GEMM
// workgroup of the update lives ~290 us and workgroup slots come free only at those boundaries, so
// each chain kernel waits about that long.  Here the chain's workgroups
// are dispatched ONCE, while the chip is idle, one per compute unit (8 waves, ~150 KB of LDS, so a
// bulk GEMM workgroup can never share the unit and disturb the leaf's dependent chain), they
// announce their arrival through a stream-memory-op flag that releases the bulk update on the
// other stream, and they then walk a host-built op list with grid barriers in between:
//
//     LEAF   workgroup 0 factors a diagonal block            (potrf_leaf_body<8>)
//     STRIP  row slabs dealt over all wavefronts              (trsm_strip_slab)
//     GEMM   tile list dealt over all workgroups              (gemm_f64_body, 128 x 128 / 8 waves)
//
// The grid barrier is a generation counter in device memory: release fence + arrive, spin with
// s_sleep, acquire fence (agent scope, so the per-XCD L2s are written back / invalidated exactly
// as at a kernel boundary).  Every spin carries a watchdog that raises an abort flag after ~2 s,
// on which all workgroups leave: a scheduling bug cannot hang the device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_f64.hpp"
#include "potrf_leaf.hpp"
#include "trsm_strip.hpp"

namespace gmb {

enum ChainOpType : int32_t { CHAIN_LEAF = 0, CHAIN_STRIP = 1, CHAIN_GEMM = 2 };

struct ChainOp {
  int32_t type;
  int32_t nblocks;  // GEMM: virtual blocks of the tile list
  LeafArgs leaf;
  TrsmArgs trsm;
  GemmArgs gemm;
};

struct ChainSync {
  unsigned int count;      // XCD leaders arrived at the current barrier
  unsigned int gen;        // barrier generation (device-wide)
  unsigned int abort;      // watchdog fired
  unsigned int nxcd;       // XCDs that host chain workgroups (set by the first barrier of a launch)
  unsigned int xcount[8];  // per XCD: workgroups arrived
  unsigned int xgen[8];    // per XCD: generation, flipped by the XCD's leader once its L2 is coherent again
  unsigned int xwgs[8];    // per XCD: chain workgroups resident there
  unsigned int reg[8];     // registration scratch of the first barrier
  unsigned int colflag;    // chol_column_kernel: sequence number of the last finished leaf
  unsigned int pad[3];
};

__device__ __forceinline__ unsigned int chain_xcc_id() {
  unsigned int x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 7u;
}

#define CHAIN_LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define CHAIN_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define CHAIN_ADD(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// spin until *p != old; false when the watchdog (or another workgroup's) fired
__device__ __forceinline__ bool chain_spin(ChainSync* s, unsigned int* p, unsigned int old) {
  const long long t0 = wall_clock64();  // 100 MHz
  while (CHAIN_LD(p) == old) {
    __builtin_amdgcn_s_sleep(2);
    if (CHAIN_LD(&s->abort)) return false;
    if (wall_clock64() - t0 > 200000000LL) {  // 2 s
      CHAIN_ST(&s->abort, 1u);
      return false;
    }
  }
  return true;
}

// Grid barrier with memory visibility, hierarchical over XCDs.  Each XCD has its own L2 and plain
// accesses are coherent only inside it, so what a barrier has to do is: every workgroup's stores
// reach its XCD's L2 (the s_waitcnt of __syncthreads), ONE workgroup per XCD writes that L2 back
// (agent-scope release) and joins the device-wide count, and after the flip ONE workgroup per XCD
// invalidates that L2 (agent-scope acquire) before it lets its XCD's workgroups go; those only
// drop their own vector L1.  (A flat barrier -- 64 L2 write-backs and 512 L2 invalidates per
// barrier -- made the chain 3x slower and the concurrent trailing update 1.8x slower.)
__device__ __forceinline__ bool chain_barrier(ChainSync* s, unsigned int xcd) {
  __shared__ int ok;
  __syncthreads();  // workgroup-scope release: every wave's stores are acknowledged by this XCD's L2
  if (threadIdx.x == 0) {
    bool good = true;
    const unsigned int xg = CHAIN_LD(&s->xgen[xcd]);
    const unsigned int prev = CHAIN_ADD(&s->xcount[xcd], 1u);
    if (prev + 1 == CHAIN_LD(&s->xwgs[xcd])) {  // this XCD's leader for this barrier
      CHAIN_ST(&s->xcount[xcd], 0u);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // write this XCD's L2 back
      const unsigned int g = CHAIN_LD(&s->gen);
      const unsigned int arrived = CHAIN_ADD(&s->count, 1u);
      if (arrived + 1 == CHAIN_LD(&s->nxcd)) {
        CHAIN_ST(&s->count, 0u);
        CHAIN_ADD(&s->gen, 1u);
      } else {
        good = chain_spin(s, &s->gen, g);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // invalidate this XCD's L2 (and this unit's L1)
      CHAIN_ADD(&s->xgen[xcd], 1u);
    } else {
      good = chain_spin(s, &s->xgen[xcd], xg);
      asm volatile("buffer_inv sc0" ::: "memory");  // this compute unit's vector L1 only
    }
    ok = good ? 1 : 0;
  }
  __syncthreads();
  return ok != 0;
}

// First barrier of a launch: flat (every workgroup fences at agent scope) and it registers the
// workgroups per XCD so that the hierarchical barrier knows its head counts.
__device__ __forceinline__ bool chain_register(ChainSync* s, unsigned int xcd, unsigned int nwg) {
  __shared__ int ok;
  __syncthreads();
  if (threadIdx.x == 0) {
    bool good = true;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    CHAIN_ADD(&s->reg[xcd], 1u);
    const unsigned int g = CHAIN_LD(&s->gen);
    const unsigned int prev = CHAIN_ADD(&s->count, 1u);
    if (prev + 1 == nwg) {
      unsigned int n = 0;
      for (int x = 0; x < 8; ++x) {
        const unsigned int c = CHAIN_LD(&s->reg[x]);
        CHAIN_ST(&s->xwgs[x], c);
        CHAIN_ST(&s->reg[x], 0u);
        CHAIN_ST(&s->xcount[x], 0u);
        n += c > 0;
      }
      CHAIN_ST(&s->nxcd, n);
      CHAIN_ST(&s->count, 0u);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      CHAIN_ADD(&s->gen, 1u);
    } else {
      good = chain_spin(s, &s->gen, g);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    ok = good ? 1 : 0;
  }
  __syncthreads();
  return ok != 0;
}

// The three op bodies are real calls, not inlined: inlined into one function the register
// allocator spilled ~390 VGPRs into scratch inside the hot loops.
__device__ __attribute__((noinline)) void chain_op_leaf(const ChainOp* op) {
  const LeafArgs a = op->leaf;
  potrf_leaf_body<8>(a);
}
__device__ __attribute__((noinline)) void chain_op_strip(const ChainOp* op, unsigned int nwg) {
  const TrsmArgs t = op->trsm;
  const int64_t nslab = t.nrows / 16;
  const int wave = threadIdx.x >> 6;
  for (int64_t sl = (int64_t)blockIdx.x * 8 + wave; sl < nslab; sl += (int64_t)nwg * 8) trsm_strip_slab(t, sl * 16);
}
__device__ __attribute__((noinline)) void chain_op_gemm(const ChainOp* op, unsigned int nwg) {
  const GemmArgs g = op->gemm;
  const int nb = op->nblocks;
  for (int vb = blockIdx.x; vb < nb; vb += (int)nwg) gemm_f64_body<2, 4, 4, 2, false>(g, vb);
}

// grid = number of chain workgroups (<= compute units), block = 512
__global__ __launch_bounds__(512, 2) void panel_chain_kernel(const ChainOp* __restrict__ ops, int nops,
                                                             ChainSync* sync, unsigned int* arrived_flag,
                                                             unsigned int seq, long long* stamps) {
  const unsigned int nwg = gridDim.x;
  const unsigned int xcd = chain_xcc_id();
  // all workgroups resident: release the bulk update that waits on the other stream
  if (!chain_register(sync, xcd, nwg)) return;
  if (blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(arrived_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  if (stamps && blockIdx.x == 0 && threadIdx.x == 0) stamps[0] = wall_clock64();
  for (int i = 0; i < nops; ++i) {
    const ChainOp* op = ops + i;
    const int type = op->type;
    if (type == CHAIN_LEAF) {
      if (blockIdx.x == 0) chain_op_leaf(op);
    } else if (type == CHAIN_STRIP) {
      chain_op_strip(op, nwg);
    } else {
      chain_op_gemm(op, nwg);
    }
    if (stamps && blockIdx.x == 0 && threadIdx.x == 0) stamps[2 * i + 1] = wall_clock64();  // op done (this workgroup)
    if (!chain_barrier(sync, xcd)) return;
    if (stamps && blockIdx.x == 0 && threadIdx.x == 0) stamps[2 * i + 2] = wall_clock64();  // barrier passed
  }
}

// ---- one block column of the Cholesky as ONE launch ---------------------------------------------
// Workgroup 0 factors the diagonal block; the other workgroups (128 rows each, one 16-row slab per
// wavefront) load their panel rows at once, wait for a flag and then run the strip solve.  Saves
// the launch gap between leaf and strip and hides the strip's panel loads behind the leaf.
// Cache coherence: the flag is an agent-scope atomic; the leaf workgroup writes its XCD's L2 back
// before it raises it.  The waiting wavefronts need no invalidate of their own: every cache was
// invalidated when this kernel was dispatched and nothing on their side touches the diagonal block
// or its sub-block inverses before the flag (the panel rows are different cache lines).
struct ColumnArgs {
  LeafArgs leaf;
  TrsmArgs trsm;
  unsigned int* flag;
  unsigned int seq;
  unsigned int* abort_flag;
};

__global__ __launch_bounds__(512) void chol_column_kernel(ColumnArgs g) {
  if (blockIdx.x == 0) {
    potrf_leaf_body<8>(g.leaf);  // ends with a workgroup barrier: every wave's stores are acknowledged
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __hip_atomic_store(g.flag, g.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  __builtin_amdgcn_s_setprio(3);
  const int64_t r0 = (((int64_t)blockIdx.x - 1) * 8 + (threadIdx.x >> 6)) * 16;
  if (r0 >= g.trsm.nrows) return;
  strip_d4 X[8];
  trsm_strip_load(g.trsm, r0, X);
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(g.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != g.seq) {
    __builtin_amdgcn_s_sleep(24);  // ~0.7 us: ~1300 wavefronts poll one address, keep them off the fabric
    if (wall_clock64() - t0 > 200000000LL) {  // 2 s: the leaf never came -- give up instead of hanging
      __hip_atomic_store(g.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  trsm_strip_solve_store(g.trsm, r0, X);
}

}  // namespace gmb
