// dist_driver.hpp -- ONE GP over the GPUs of a node: native driver loop of the
// multi-GPU Cholesky, the row-partitioned NLML gradient and the sharded prediction.  Included at the end of
// engine.hip (same translation unit: it drives the single-GPU building blocks chol_cols / trsm_cols /
// launch_gemm / launch_cov on this rank's share).
//
// There is no reference counterpart (the reference is single-process); the work replaced is what
// `pm.find_MAP` / `pm.gp.Marginal.predict` do per evaluation (gumbi/regression/pymc/GP.py:811, 845-847).
//
// Partition.  One process per GPU.  128-row blocks of the bordered covariance matrix (row N carries y) are
// dealt round-robin: rank g of G owns block rows b = g (mod G) -- a 1-D block-cyclic ROW partition, so the
// long late rows are spread evenly and X (replicated, 8 N d bytes) lets every rank build its rows with no
// communication.  Every rank keeps a FULL-size factor buffer (80 GB at N = 1e5 of 288 GB) and ends with the
// complete factor, so prediction shards over the test points with no further exchange.
//
// Right-looking over panels [c0, c1) of `w` block columns, ONE collective type (all-gather), three per panel -- two small ones
// on the chain (SQUARE, PANEL = the head of the panel column), one large one beside it (TAIL; round 6):
//   SQUARE  the panel's diagonal square is spread over the ranks (block row i belongs to rank i mod G):
//           all-gather its block rows (w x w blocks, <= a few MB), then EVERY rank factors the square itself
//           with the single-GPU chain (redundant ~w^3 128^3 / 3 flops, but no broadcast hop, no owner
//           bottleneck, and log-det / failure index come out identical on all ranks with no reduction);
//   SOLVE   every rank solves ITS block rows below the square against it (packed, trsm_cols; no communication);
//   PANEL   all-gather of the panel column's HEAD: its first w block rows = the rows of the next panel's square, all that
//           U1, the next SQUARE and the next SOLVE need from the other ranks (main stream: on the chain, but small);
//   TAIL    all-gather of the rest of the panel column on the COMMUNICATION stream, beside the next chain; only U2 reads it.
//           PANEL + TAIL are the north star's "panel broadcast": with one sender per block row an all-gather is what drives
//           all 7 xGMI links of every GPU at once;
//   UPDATE  the trailing update is purely local: rank g updates its own block rows (strided n index of the
//           MFMA GEMM) with the panel everybody now holds.
// Look-ahead: the update by panel p is split into U1 (the columns of panel p+1, main stream) and U2 (the
// rest, bulk stream = the CU-masked stream when available, so the chain's small kernels and RCCL's
// workgroups find free compute units); SQUARE / PANEL of p+1 run on the main stream beside U2(p).
//
//   main :  SQUARE(0) SOLVE(0) PANEL(0) | U1(0) SQUARE(1) SOLVE(1) PANEL(1) | wait U2(0); U1(1) SQUARE(2) ... | ...
//   comm :                               | TAIL(0) ...........              | TAIL(1) ...........              | ...
//   bulk :                               |         wait TAIL(0); U2(0) ...........| wait TAIL(1); U2(1) ............| ...
//
// The schedule is built as a PLAN first (dist_build_plan: host-only, pure integer arithmetic, exported as
// gmb_dist_plan) and then executed; the CPU tests replay the same plan with numpy blocks over gloo.
//
// Gradient (gmb_dist_nlml): U = L^-T BY ROWS -- rank g solves V <- V L^-T for its block rows of the identity
// (no communication, ~N^3/(2G) flops with the structural zeros skipped), in column chunks: while chunk j+1
// is being solved, chunk j travels (all-gather on a second stream) straight into the upper triangle of every
// rank's factor buffer; then each rank forms ITS block rows of Sigma^-1 = U U^T (packed, in the buffer V
// occupied) and the fused trace reductions over them; the accumulators are all-gathered and summed in rank
// order, so every rank holds bit-identical (value, gradient) and an optimiser runs in lock step.
#pragma once
#include <dlfcn.h>

#include <rccl/rccl.h>  // types and prototypes only: librccl is resolved at run time with dlopen

namespace {

enum DistOp : int32_t { DIST_KBUILD = 0, DIST_SQUARE = 1, DIST_PANEL = 2, DIST_UPDATE = 3, DIST_FORK = 4, DIST_JOIN = 5, DIST_SOLVE = 6, DIST_TAIL = 7 };

// block rows of [lo, hi) owned by `rank`: first + t * G, t < count
inline void dist_owned(int rank, int G, int lo, int hi, int* first, int* count) {
  int f = lo + (((rank - lo) % G) + G) % G;
  *first = f;
  *count = f < hi ? (hi - f + G - 1) / G : 0;
}
inline int dist_max_owned(int G, int lo, int hi) { return hi > lo ? (hi - lo + G - 1) / G : 0; }

// Panel width in block columns when the caller leaves it open: a sixteenth of the matrix for up to four ranks; with more ranks a
// rank's share of the trailing update shrinks while the panel's chain (gather the square, factor it, solve, gather the panel)
// does not, so the panels get narrower -- one-rank emulation with a 300 GB/s transport model at N = 100k
// (profiles/r05_replicated_emulate.txt): G = 8: 694 / 699 / 708 / 737 / 782 ms for 16 / 24 / 32 / 48 / 64 block columns (the bulk stream
// waits 71 ms for the chain at 24, 147 ms at 48); G = 4: flat from 32 to 48; G = 2: 2508 / 2468 / 2446 ms for 32 / 48 / 64.
int dist_default_panel(int nct, int G) {
  const int base = nct / 16 + 4;
  const int w = G <= 4 ? base : G < 8 ? base * 3 / 4 : base / 2;  // (two ranks would take 64 -- 0.9 % -- at the price of wider panel buffers)
  return std::max(8, (w / 8) * 8);
}

std::vector<gmb_dist_step> dist_build_plan(int64_t N, int rank, int G, int w) {
  std::vector<gmb_dist_step> plan;
  const int nct = (int)((N + TILE - 1) / TILE), nrt = (int)((N + 1 + TILE - 1) / TILE);
  if (w <= 0) w = dist_default_panel(nct, G);
  auto push = [&](int op, int c0, int c1, int lo, int hi, int stream) {
    gmb_dist_step s{};
    s.op = op;
    s.c0 = c0;
    s.c1 = c1;
    s.lo = lo;
    s.hi = hi;
    s.stream = stream;
    dist_owned(rank, G, lo, hi, &s.first, &s.count);
    if (op == DIST_UPDATE) dist_owned(rank, G, lo, nrt, &s.first, &s.count);  // rows run to the end
    s.maxcount = dist_max_owned(G, lo, hi);
    s.elems = (op == DIST_SQUARE || op == DIST_PANEL || op == DIST_TAIL) ? (int64_t)s.maxcount * TILE * (int64_t)(c1 - c0) * TILE : 0;
    plan.push_back(s);
  };
  push(DIST_KBUILD, 0, nct, 0, nrt, 0);
  // The panel column travels in TWO pieces (round 6): PANEL = its first w block rows -- the rows of the next panel's square, all
  // that U1, the next SQUARE and the next SOLVE need from the other ranks: a small message on the main stream, i.e. on the chain
  // -- and TAIL = everything below, which only U2 reads: the large message, on the communication stream, beside the next chain.
  // (Before: one all-gather of the whole column on the chain -- 4.8 of the ~5 ms a chain spends in collectives at N = 100k, G = 8.)
  auto chain = [&](int c0, int c1) {
    push(DIST_SQUARE, c0, c1, c0, c1, 0);
    if (nrt > c1) {
      const int head_hi = std::min(c1 + w, nrt);
      push(DIST_SOLVE, c0, c1, c1, nrt, 0);
      push(DIST_PANEL, c0, c1, c1, head_hi, 0);
      if (nrt > head_hi) push(DIST_TAIL, c0, c1, head_hi, nrt, 2);
    }
  };
  chain(0, std::min(w, nct));
  for (int c0 = 0; c0 < nct; c0 += w) {
    const int c1 = std::min(c0 + w, nct), c2 = std::min(c1 + w, nct);
    if (c1 >= nct) break;
    if (c0 > 0) push(DIST_JOIN, 0, 0, 0, 0, 0);        // U2(p-1) has reached panel p+1's columns
    push(DIST_FORK, 0, 0, 0, 0, 0);                    // bulk: PANEL(p) is complete
    push(DIST_UPDATE, c0, c1, c1, c2, 0);              // U1(p)
    if (c2 < nct) push(DIST_UPDATE, c0, c1, c2, nct, 1);  // U2(p)
    chain(c1, c2);
  }
  push(DIST_JOIN, 0, 0, 0, 0, 0);
  return plan;
}

int dist_pack(gmb_engine* e, hipStream_t st, double* mat, int64_t ld, double* packed, int64_t ldp, int64_t seg_elems,
              int ncols, bool to_packed, int nseg, int first, int count, int stride, int lo, int hi, int maxcount) {
  if (ncols <= 0 || maxcount <= 0) return GMB_OK;
  PackArgs a{};
  a.mat = mat;
  a.ld = ld;
  a.packed = packed;
  a.ldp = ldp;
  a.seg_elems = seg_elems;
  a.ncols = ncols;
  a.to_packed = to_packed ? 1 : 0;
  a.nseg = nseg > 0 ? nseg : 1;
  a.by_rank = nseg > 0 ? 1 : 0;  // nseg = 0: this rank's own rows (first, count); nseg = G: the receive side
  a.first = first;
  a.count = count;
  a.stride = stride;
  a.lo = lo;
  a.hi = hi;
  if (!a.by_rank && count <= 0) return GMB_OK;
  hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)(a.by_rank ? maxcount : count), (unsigned)((ncols + 3) / 4), (unsigned)a.nseg),
                     dim3(256), 0, st, a);
  HIP_TRY(e, hipGetLastError());
  return GMB_OK;
}

// ---- communication probes --------------------------------------------------------------------------------
// A HIP event pair around every all-gather (on the stream it is ordered on: first event = the stream has
// reached the collective, second = the collective is complete, i.e. transfer + wait for the slowest peer) and
// an event on either side of every cross-stream wait of the panel loop.  Read back after the call's final
// synchronisation; gmb_timings.dist_* (include/gumbi_hip.h) says what is derived from them.
struct DistProbe {
  struct Coll {
    hipEvent_t a, b;
    double bytes;
    int group;      // factorisation: index of the panel chain the collective belongs to
    bool exposed;   // sits on the main stream of a phase with nothing beside it
  };
  struct Wait {
    hipEvent_t arrive, release;  // waiting stream reached the wait / the other stream reached the point waited for
    int group;
    int kind;  // 0: main waits for bulk (JOIN), 1: bulk waits for main (FORK), 2: main waits for the comm stream
  };
  std::vector<Coll> colls;
  std::vector<Wait> waits;
  int group = -1;
};

// one more collective in this engine's issue log (see gmb_engine::coll_hash; 52 bits, so that it travels exactly in a double)
void dist_log_collective(gmb_engine* e, int64_t count, hipStream_t st) {
  const uint64_t tag = st == e->stream ? 0u : st == e->aux[0] ? 1u : 2u;
  e->coll_count += 1;
  e->coll_hash = (e->coll_hash * 1099511628211ull + (uint64_t)count * 4u + tag + 1u) & ((1ull << 52) - 1);
}

int dist_all_gather(gmb_engine* e, const gmb_comm* comm, hipStream_t st, const double* send, double* recv, int64_t count,
                    DistProbe* probe = nullptr, bool exposed = false) {
  dist_log_collective(e, count, st);
  DistProbe::Coll c{};
  if (probe) {
    c.a = next_time_event(e);
    c.b = next_time_event(e);
    c.bytes = 8.0 * (double)count * (double)(comm->world - 1);
    c.group = probe->group;
    c.exposed = exposed;
    (void)hipEventRecord(c.a, st);
  }
  const int32_t rc = comm->all_gather(comm->ctx, send, recv, count, (void*)st);
  if (probe) {
    (void)hipEventRecord(c.b, st);
    probe->colls.push_back(c);
  }
  if (rc != 0) return fail(e, GMB_EHIP, "all-gather of %lld doubles failed on rank %d (transport status %d)",
                           (long long)count, comm->rank, rc);
  return GMB_OK;
}

// `waiter` waits for everything enqueued on `other` so far (order_after), with an event on each side
int dist_wait(gmb_engine* e, hipStream_t other, hipStream_t waiter, DistProbe* probe, int kind) {
  if (other == waiter) return GMB_OK;
  if (probe) {
    DistProbe::Wait w{};
    w.arrive = next_time_event(e);
    w.release = next_time_event(e);
    w.group = probe->group;
    w.kind = kind;
    (void)hipEventRecord(w.arrive, waiter);
    (void)hipEventRecord(w.release, other);
    probe->waits.push_back(w);
  }
  return order_after(e, other, waiter);
}

double dist_ms(hipEvent_t a, hipEvent_t b);

// Exposed collective time of a factorisation from its probes: per panel chain, min(what the bulk stream waited at the chain's
// FORK -- for the main stream and, since round 6, for the communication stream's TAIL --, what the chain's collectives took);
// chains with no update behind them (a matrix of a single panel) count in full: nothing could hide their collectives.
double dist_exposed_from_probes(const DistProbe& probe, const std::vector<double>& chain_comm, double* main_wait_ms, double* bulk_wait_ms) {
  std::vector<double> bulk_wait(chain_comm.size(), 0.0);
  std::vector<bool> seen(chain_comm.size(), false);
  for (const DistProbe::Wait& w : probe.waits) {
    const double t = dist_ms(w.arrive, w.release);
    if (w.kind == 0) {
      *main_wait_ms += t;
    } else if (w.kind == 1) {
      if (w.group < 0) *bulk_wait_ms += t;
      if (w.group >= 0) {
        bulk_wait[(size_t)w.group] = std::max(bulk_wait[(size_t)w.group], t);  // (both waits of a FORK start when the bulk stream arrives)
        seen[(size_t)w.group] = true;
      }
    }
  }
  double exposed = 0.0;
  for (size_t g = 0; g < chain_comm.size(); ++g) {
    *bulk_wait_ms += bulk_wait[g];
    exposed += seen[g] ? std::min(bulk_wait[g], chain_comm[g]) : chain_comm[g];
  }
  return exposed;
}

double dist_ms(hipEvent_t a, hipEvent_t b) {
  float t = 0.f;
  if (hipEventElapsedTime(&t, a, b) != hipSuccess) {
    (void)hipGetLastError();
    return 0.0;
  }
  return t > 0.f ? (double)t : 0.0;
}

void rccl_note_main_stream(const gmb_comm* comm, hipStream_t main_stream);  // (the library's own transport: see RcclCtx below)

int dist_check_comm(gmb_engine* e, const gmb_comm* comm) {
  if (!comm || !comm->all_gather || comm->world < 1 || comm->world > DIST_MAX_WORLD || comm->rank < 0 ||
      comm->rank >= comm->world)
    return fail(e, GMB_EINVAL, "bad communicator (1 <= world <= %d)", DIST_MAX_WORLD);
  rccl_note_main_stream(comm, e->stream);  // collectives on this stream and on any other one use different communicators
  return GMB_OK;
}

// The ranks agree on a status: every rank contributes `mine` (0 or a negative gmb_status), every rank
// receives the first non-zero one in rank order (and whose it was).  Synchronises e->stream.  A rank that
// failed locally -- an allocation, a launch, its transport -- must not simply return from a collective
// call: its peers would block in their next all-gather for ever.  gmb_dist_* therefore (i) allocate
// everything first and agree on the outcome before the first data collective, (ii) after that never leave
// the collective sequence: a failing rank keeps issuing the remaining all-gathers of the plan (on whatever its
// buffers hold) and (iii) agree on the outcome once more at the end, so that every rank returns an error when
// any rank failed.
// `payload` (optional, npayload <= DIST_MAX_PAYLOAD doubles): scalars every rank computed REDUNDANTLY and on which
// the callers' control flow depends (the factorisation's failure index, log-determinant, |v|^2 -> the NLML an
// optimiser compares between steps).  They are bit-identical by construction -- the same deterministic kernels on
// the same gathered bits -- but an optimiser running in lock step on every rank must never depend on that: all
// ranks leave with RANK 0's copy, and *repairs counts the ranks whose own copy differed (0 unless something
// -- a cosmic ray, a driver bug -- broke the redundancy; reported as gmb_timings.dist_lockstep_repairs).
int dist_agree(gmb_engine* e, const gmb_comm* comm, int mine, const char* where, double* payload = nullptr,
               int npayload = 0, int64_t* repairs = nullptr) {
  if (comm->world == 1) return mine;
  if (npayload > DIST_MAX_PAYLOAD) npayload = DIST_MAX_PAYLOAD;
  const int w = DIST_HEADER + npayload;  // words per rank: [status | collectives issued | hash of their sequence | payload]
  std::vector<double> mineb((size_t)w), all((size_t)w * comm->world);
  mineb[0] = (double)mine;
  mineb[1] = (double)e->coll_count;
  mineb[2] = (double)e->coll_hash;
  for (int i = 0; i < npayload; ++i) mineb[DIST_HEADER + i] = payload[i];
  dist_log_collective(e, w, e->stream);  // (this exchange itself: the same on every rank)
  double* dsend = e->dstat;
  double* drecv = e->dstat + (DIST_HEADER + DIST_MAX_PAYLOAD);
  hipError_t st = hipMemcpyAsync(dsend, mineb.data(), w * sizeof(double), hipMemcpyHostToDevice, e->stream);
  if (st == hipSuccess) st = hipStreamSynchronize(e->stream);
  int32_t tr = comm->all_gather(comm->ctx, dsend, drecv, w, (void*)e->stream);
  if (st == hipSuccess && tr == 0)
    st = hipMemcpyAsync(all.data(), drecv, all.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream);
  if (st == hipSuccess) st = hipStreamSynchronize(e->stream);
  if (st != hipSuccess || tr != 0) {
    (void)hipGetLastError();
    if (mine) return mine;
    return fail(e, GMB_EHIP, "%s: the ranks could not exchange their status (transport status %d, %s)", where, tr,
                hipGetErrorString(st));
  }
  for (int q = 0; q < comm->world; ++q)
    if (all[(size_t)q * w] != 0.0) {
      const int code = (int)all[(size_t)q * w];
      if (q == comm->rank) return mine;  // this rank's own message is already in e->err
      return fail(e, code, "%s: rank %d of %d failed with status %d; no rank has a usable result", where, q, comm->world,
                  code);
    }
  // every rank must have issued the same collectives in the same order on this transport (RCCL pairs them by issue order)
  for (int q = 1; q < comm->world; ++q)
    if (all[(size_t)q * w + 1] != all[1] || all[(size_t)q * w + 2] != all[2])
      return fail(e, GMB_EHIP, "%s: rank %d issued a different sequence of collectives than rank 0 (%.0f vs %.0f calls, hash %.0f vs %.0f): "
                  "the ranks' calls on this engine have diverged", where, q, all[(size_t)q * w + 1], all[1], all[(size_t)q * w + 2], all[2]);
  if (npayload > 0) {
    int64_t differ = 0;
    for (int q = 1; q < comm->world; ++q)
      if (std::memcmp(&all[(size_t)q * w + DIST_HEADER], &all[DIST_HEADER], npayload * sizeof(double)) != 0) ++differ;
    for (int i = 0; i < npayload; ++i) payload[i] = all[DIST_HEADER + i];
    if (repairs) *repairs += differ;
  }
  return GMB_OK;
}

// first error of a call that must go on issuing its collectives
struct DistDeferred {
  int rc = 0;
  std::string msg;
  void note(gmb_engine* e, int r) {
    if (r && !rc) {
      rc = r;
      msg = e->err;
    }
  }
  int give(gmb_engine* e) const {
    if (rc) e->err = msg;
    return rc;
  }
};

int dist_factorize(gmb_engine* e, const gmb_comm* comm, int panel_blocks) {
  int rc = dist_check_comm(e, comm);
  if (rc) return rc;  // nothing to talk through
  const int G = comm->world, rank = comm->rank;
  gmb_timings& tm = e->tm;
  // ---- everything that can fail locally happens before the first collective; then the ranks agree ----
  rc = require_ready(e, false);
  if (!rc && hipSetDevice(e->device) != hipSuccess) rc = fail(e, GMB_EHIP, "hipSetDevice(%d) failed", e->device);
  std::vector<gmb_dist_step> plan;
  if (!rc) rc = ensure_factor_buffer(e);
  if (!rc) {
    plan = dist_build_plan(e->N, rank, G, panel_blocks > 0 ? panel_blocks : (e->panel_auto ? 0 : e->panel_blocks));
    int64_t need = 0, need_tail = 0;
    for (const gmb_dist_step& s : plan) {
      if (s.op == DIST_TAIL) need_tail = std::max(need_tail, s.elems);
      else need = std::max(need, std::max(s.elems, s.op == DIST_SOLVE ? (int64_t)s.maxcount * TILE * (int64_t)(s.c1 - s.c0) * TILE : 0));
    }
    if (!(rc = ensure(e, &e->dsend, &e->cap_send, need))) rc = ensure(e, &e->drecv, &e->cap_recv, need * G);
    // (the TAIL travels on the communication stream while the next chain uses dsend / drecv: a staging pair of its own)
    if (!rc && need_tail > 0 && !(rc = ensure(e, &e->dsend2, &e->cap_send2, need_tail))) rc = ensure(e, &e->drecv2, &e->cap_recv2, need_tail * G);
  }
  e->coll_count = e->coll_hash = 0;  // the issue log covers this call: its closing agreement compares the ranks' sequences
  if ((rc = dist_agree(e, comm, rc, "gmb_dist_factorize (set-up)"))) return rc;
  e->factored = false;
  e->factor_kind = gmb_engine::FK_REPLICATED;
  e->factor_consumed = false;
  e->have_alpha = false;
  e->notpd = -1;
  tm.kbuild_ms = tm.chol_ms = tm.chol_gemm_ms = tm.chol_gemm_flops = 0.0;
  tm.chol_leaf_ms = tm.chol_trsm_ms = 0.0;
  tm.chol_gemm_launches = 0;
  const int nct = (int)(e->Np / TILE), nrt = (int)(e->Nr / TILE);
  hipStream_t mainS = e->stream, bulkS = e->aux[2], commS = e->aux[0];
  e->sync_next = 0;
  e->time_next = 0;
  e->cur = mainS;
  e->chol_update_kind = 0;  // chol_cols factors the panels' squares: in-panel products
  DistDeferred bad;
  DistProbe probe;
  bool tail_in_flight = false;  // a TAIL has been issued on the communication stream since the last FORK
  {
    hipError_t st = hipMemsetAsync(e->dscal, 0, 64 * sizeof(double), e->stream);
    if (st == hipSuccess) st = hipMemsetAsync(e->dinfo, 0, sizeof(int32_t), e->stream);
    if (st != hipSuccess) bad.note(e, fail(e, GMB_EHIP, "hipMemsetAsync failed: %s", hipGetErrorString(st)));
  }
  PhaseTimer tk(e);
  PhaseTimer* tc = nullptr;
  for (const gmb_dist_step& s : plan) {
    const int64_t W = (int64_t)(s.c1 - s.c0) * TILE;
    const bool has_collective = s.op == DIST_SQUARE || s.op == DIST_PANEL || s.op == DIST_TAIL;
    bool gathered = false;
    rc = GMB_OK;
    if (s.op == DIST_SQUARE) ++probe.group;
    if (s.op == DIST_TAIL) {
      // the staging pair of the TAILs is free once the previous TAIL has been unpacked; this one may leave once the main stream
      // has solved the rows (both orders are issued whatever this rank's state: the peers' streams do the same)
      bad.note(e, order_after(e, commS, mainS));
    }
    if (!bad.rc) switch (s.op) {
      case DIST_KBUILD: {  // this rank's block rows of the lower triangle, y row and padding included
        CovTileArgs a{};
        a.p = e->cp;
        a.rows = train_set(e);
        a.cols = train_set(e);
        a.out = e->dA;
        a.ldo = e->ld;
        a.ti = nrt;
        a.tj = nct;
        a.mode = COV_TRAIN;
        a.lower_only = 1;
        a.row_first = rank;
        a.row_stride = G;
        a.y = e->dy;
        if ((rc = launch_cov(e, a))) break;
        for (size_t t = 1; t < e->terms.size() && !rc; ++t) {  // additive models: accumulating passes
          if ((rc = prep_points(e, e->dX, e->N, e->D, e->Nr, e->xs, e->xl, e->cat, &e->terms[t].pa))) break;
          a.p = e->terms[t].cp;
          a.accumulate = 1;
          rc = launch_cov(e, a);
        }
        if (!rc && e->terms.size() > 1) rc = prep_points(e, e->dX, e->N, e->D, e->Nr, e->xs, e->xl, e->cat, &e->terms[0].pa);
        break;
      }
      case DIST_SQUARE: {
        double* cols = e->dA + (int64_t)s.c0 * TILE * e->ld;  // column block c0 of the factor buffer
        const int64_t ldp = (int64_t)s.maxcount * TILE;
        if ((rc = dist_pack(e, mainS, cols, e->ld, e->dsend, ldp, 0, (int)W, true, 0, s.first, s.count, G, 0, 0, s.maxcount))) break;
        gathered = true;
        if ((rc = dist_all_gather(e, comm, mainS, e->dsend, e->drecv, s.elems, &probe))) break;
        if ((rc = dist_pack(e, mainS, cols, e->ld, e->drecv, ldp, s.elems, (int)W, false, G, 0, 0, G, s.lo, s.hi, s.maxcount))) break;
        e->cur = mainS;
        rc = chol_cols(e, s.c0, s.c1, s.c1);  // the square only: rows below are PANEL's
        break;
      }
      case DIST_SOLVE: {  // my rows below the square, solved in the packed staging buffer and put back: no communication
        double* cols = e->dA + (int64_t)s.c0 * TILE * e->ld;
        const int64_t ldp = (int64_t)s.maxcount * TILE;
        if (s.count <= 0) break;
        if ((rc = dist_pack(e, mainS, cols, e->ld, e->dsend, ldp, 0, (int)W, true, 0, s.first, s.count, G, 0, 0, s.maxcount))) break;
        e->cur = mainS;
        if ((rc = trsm_cols(e, e->dsend - (int64_t)s.c0 * TILE * ldp, ldp, s.count, s.c0, s.c1, 2, 5))) break;
        rc = dist_pack(e, mainS, cols, e->ld, e->dsend, ldp, 0, (int)W, false, 0, s.first, s.count, G, 0, 0, s.maxcount);
        break;
      }
      case DIST_PANEL:   // the head of the panel column (main stream) ...
      case DIST_TAIL: {  // ... and the rest of it (communication stream, staging of its own): everybody's solved rows [lo, hi)
        double* cols = e->dA + (int64_t)s.c0 * TILE * e->ld;
        const int64_t ldp = (int64_t)s.maxcount * TILE;
        const bool tail = s.op == DIST_TAIL;
        hipStream_t st = tail ? commS : mainS;
        double* snd = tail ? e->dsend2 : e->dsend;
        double* rcv = tail ? e->drecv2 : e->drecv;
        if ((rc = dist_pack(e, mainS, cols, e->ld, snd, ldp, 0, (int)W, true, 0, s.first, s.count, G, 0, 0, s.maxcount))) break;
        if (tail && (rc = order_after(e, mainS, commS))) break;
        gathered = true;
        if ((rc = dist_all_gather(e, comm, st, snd, rcv, s.elems, &probe))) break;
        // (the rank's own rows are rewritten with the values they hold: U1 may be reading them on the main stream meanwhile)
        rc = dist_pack(e, st, cols, e->ld, rcv, ldp, s.elems, (int)W, false, G, 0, 0, G, s.lo, s.hi, s.maxcount);
        if (tail) tail_in_flight = true;
        break;
      }
      case DIST_UPDATE: {
        if (s.count <= 0) break;
        GemmArgs g{};
        g.C = e->dA + (int64_t)s.first * TILE + (int64_t)s.lo * TILE * e->ld;
        g.ldc = e->ld;
        g.A = e->dA + (int64_t)s.lo * TILE + (int64_t)s.c0 * TILE * e->ld;
        g.lda = e->ld;
        g.B = e->dA + (int64_t)s.first * TILE + (int64_t)s.c0 * TILE * e->ld;
        g.ldb = e->ld;
        g.mt = s.hi - s.lo;
        g.nt = s.count;
        g.k = (int)W;
        g.alpha = -1.0;
        g.beta = 1.0;
        g.tri = 1;
        g.tri_off = (s.first - s.lo) * TILE;
        g.nblk_stride = G;
        e->cur = s.stream ? bulkS : mainS;
        rc = launch_gemm(e, g, 7);
        e->cur = mainS;
        break;
      }
      case DIST_FORK:  // U2 reads the whole panel column: the chain's part (main stream) and the TAIL (communication stream)
        rc = dist_wait(e, mainS, bulkS, &probe, 1);
        if (!rc && tail_in_flight) rc = dist_wait(e, commS, bulkS, &probe, 1);
        tail_in_flight = false;
        break;
      case DIST_JOIN: rc = dist_wait(e, bulkS, mainS, &probe, 0); break;
    }
    bad.note(e, rc);
    // a rank in trouble keeps its place in the collective sequence: its peers are waiting in this all-gather
    if (has_collective && !gathered) {
      if (s.op == DIST_TAIL) {
        bad.note(e, order_after(e, mainS, commS));
        bad.note(e, dist_all_gather(e, comm, commS, e->dsend2, e->drecv2, s.elems));
      } else {
        bad.note(e, dist_all_gather(e, comm, mainS, e->dsend, e->drecv, s.elems));
      }
    }
    if (s.op == DIST_KBUILD) {
      tk.stop();
      tc = new PhaseTimer(e);
    }
  }
  bad.note(e, order_after(e, commS, mainS));  // the last TAIL (the y row travels in it)
  e->cur = mainS;
  // v = L^-1 y is row N of the (now complete, replicated) factor
  double hs[2] = {0.0, 0.0};
  int32_t info = 0;
  if (!bad.rc) {
    hipLaunchKernelGGL(extract_v_kernel, dim3(EXTRACT_V_BLOCKS), dim3(TILE), 0, e->stream, e->dA, e->ld, e->N, e->dv, e->dscal + 1);
    if (tc) tc->stop();
    hipError_t st = hipGetLastError();
    if (st == hipSuccess) st = hipMemcpyAsync(hs, e->dscal, 2 * sizeof(double), hipMemcpyDeviceToHost, e->stream);
    if (st == hipSuccess) st = hipMemcpyAsync(&info, e->dinfo, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream);
    if (st == hipSuccess) st = hipStreamSynchronize(e->stream);
    if (st != hipSuccess) bad.note(e, fail(e, GMB_EHIP, "distributed factorisation failed: %s", hipGetErrorString(st)));
  } else if (tc) {
    tc->stop();
  }
  double shared[3] = {(double)info, hs[0], hs[1]};  // every rank factored every square itself: identical by construction
  rc = dist_agree(e, comm, bad.give(e), "gmb_dist_factorize", shared, 3, &tm.dist_lockstep_repairs);  // synchronises the main stream
  info = (int32_t)shared[0];
  hs[0] = shared[1];
  hs[1] = shared[2];
  if (rc) {
    (void)hipStreamSynchronize(bulkS);
    e->evs.clear();  // (their events leak on this path; the engine is unusable anyway)
    delete tc;
    return rc;
  }
  tm.kbuild_ms = tk.ms();
  tm.chol_ms = tc ? tc->ms() : 0.0;
  delete tc;
  // this rank's share of the K-build bytes
  tm.kbuild_bytes = (8.0 * (double)e->N * (double)(e->N + 1) / 2.0) / G + 8.0 * (double)e->N * (double)(e->spec.n_cont + 1);
  ev_collect(e);
  if (e->profiling) {
    tm.total_kbuild_ms += tm.kbuild_ms;
    tm.total_kbuild_bytes += tm.kbuild_bytes;
    tm.total_kbuild_launches += 1;
  }
  {  // communication probes (see gmb_timings): per panel chain, what the bulk stream had to wait for
    tm.dist_world = G;
    tm.dist_chol_collectives = (int64_t)probe.colls.size();
    tm.dist_chol_comm_bytes = tm.dist_chol_comm_ms = tm.dist_chol_comm_exposed_ms = 0.0;
    tm.dist_chol_main_wait_ms = tm.dist_chol_bulk_wait_ms = 0.0;
    std::vector<double> chain_comm((size_t)probe.group + 2, 0.0);
    for (const DistProbe::Coll& c : probe.colls) {
      const double t = dist_ms(c.a, c.b);
      tm.dist_chol_comm_bytes += c.bytes;
      tm.dist_chol_comm_ms += t;
      if (c.group >= 0) chain_comm[(size_t)c.group] += t;
    }
    tm.dist_chol_comm_exposed_ms += dist_exposed_from_probes(probe, chain_comm, &tm.dist_chol_main_wait_ms, &tm.dist_chol_bulk_wait_ms);
  }
  // every rank factored every diagonal square itself: log-det and the failure index are already global and
  // identical on all ranks (same kernels on the same bits) -- no reduction
  if (info != 0) {
    e->notpd = (int64_t)info - 1;
    return fail(e, GMB_ENOTPD, "covariance matrix is not positive definite at row %lld", (long long)e->notpd);
  }
  e->logdet = hs[0];
  e->vnorm2 = hs[1];
  if (!std::isfinite(e->logdet) || !std::isfinite(e->vnorm2)) {
    e->notpd = 0;
    return fail(e, GMB_ENOTPD, "factorisation produced non-finite values");
  }
  e->factored = true;
  return GMB_OK;
}

// ---- gradient ----------------------------------------------------------------------------------------
constexpr int DIST_INV_CHUNK = 32;  // block columns of U = L^-T per pipeline stage (4096 columns)

int dist_nlml(gmb_engine* e, const gmb_comm* comm, double* nlml, double* grad) {
  int rc = dist_check_comm(e, comm);
  if (rc) return rc;
  if (!grad) {  // no collective: the value is replicated
    if ((rc = require_ready(e, true))) return rc;
    if (!nlml) return fail(e, GMB_EINVAL, "nlml output pointer is null");
    *nlml = 0.5 * (double)e->N * std::log(2.0 * M_PI) + e->logdet + 0.5 * e->vnorm2;
    return GMB_OK;
  }
  const int G = comm->world, rank = comm->rank;
  // ---- local checks and every allocation first, then the ranks agree ----
  rc = require_ready(e, true);
  if (!rc) rc = require_full_factor(e, "gmb_dist_nlml (replicated mode)");
  if (!rc && !nlml) rc = fail(e, GMB_EINVAL, "nlml output pointer is null");
  if (!rc && e->factor_consumed) rc = fail(e, GMB_EINVAL, "the factor was already consumed by a gradient call; refactorize");
  if (!rc && hipSetDevice(e->device) != hipSuccess) rc = fail(e, GMB_EHIP, "hipSetDevice(%d) failed", e->device);
  const int nt = (int)(e->Np / TILE);
  int first = 0, owned = 0;
  dist_owned(rank, G, 0, nt, &first, &owned);
  const int maxown = dist_max_owned(G, 0, nt);
  const int64_t ldv = (int64_t)maxown * TILE;
  const int cw = std::min(DIST_INV_CHUNK, std::max(nt, 1));
  const int64_t chunk_elems = (int64_t)maxown * TILE * (int64_t)cw * TILE;
  const int64_t send_need = std::max<int64_t>(chunk_elems + ldv, GACC_DOUBLES);  // [chunk | this rank's alpha rows]
  // V = this rank's block rows of U (later: of Sigma^-1), (maxown * 128) x Np column-major
  if (!rc) rc = ensure(e, &e->dW, &e->cap_W, ldv * e->Np);
  if (!rc) rc = grad_workspace(e);
  if (!rc) rc = ensure(e, &e->dsend, &e->cap_send, send_need);
  if (!rc) rc = ensure(e, &e->drecv, &e->cap_recv, send_need * G);
  e->coll_count = e->coll_hash = 0;  // the issue log covers this call: its closing agreement compares the ranks' sequences
  if ((rc = dist_agree(e, comm, rc, "gmb_dist_nlml (set-up)"))) return rc;
  *nlml = 0.5 * (double)e->N * std::log(2.0 * M_PI) + e->logdet + 0.5 * e->vnorm2;
  gmb_timings& tm = e->tm;
  tm.grad_ms = tm.grad_gemm_ms = tm.grad_gemm_flops = 0.0;
  double* V = e->dW;
  hipStream_t mainS = e->stream, commS = e->aux[0];
  PhaseTimer tg(e);
  e->sync_next = 0;
  e->time_next = 0;
  e->cur = mainS;
  e->factor_consumed = true;  // the upper triangle (and the diagonal squares) of the factor buffer become U
  DistDeferred bad;
  DistProbe probe;
  auto hip_ok = [&](hipError_t st, const char* what) {
    if (st != hipSuccess) bad.note(e, fail(e, GMB_EHIP, "%s failed: %s", what, hipGetErrorString(st)));
  };
  hip_ok(hipMemsetAsync(V, 0, (size_t)ldv * e->Np * sizeof(double), mainS), "hipMemsetAsync");
  if (!bad.rc && owned > 0) {
    hipLaunchKernelGGL(identity_rows_kernel, dim3(owned), dim3(TILE), 0, mainS, V, ldv, first, G);
    hip_ok(hipGetLastError(), "identity_rows_kernel");
  }
  bad.note(e, order_after(e, mainS, commS));
  for (int k0 = 0; k0 < nt; k0 += cw) {
    const int k1 = std::min(k0 + cw, nt);
    // rows of U with a non-zero in these columns: block rows b < k1, a prefix of the packed rows
    int f2, mine;
    dist_owned(rank, G, 0, k1, &f2, &mine);
    const int mc = dist_max_owned(G, 0, k1);
    if (!bad.rc && mine > 0) bad.note(e, trsm_cols(e, V, ldv, mine, k0, k1, 4, 6, first, G));
    bad.note(e, order_after(e, mainS, commS));  // the chunk is final
    if (!bad.rc && mine > 0 && k1 < nt) {  // right-looking: V[:, k1:] -= V[:, k0:k1] L[k1:, k0:k1]^T, structural zeros skipped
      GemmArgs g{};
      g.C = V + (int64_t)k1 * TILE * ldv;
      g.ldc = ldv;
      g.A = e->dA + (int64_t)k1 * TILE + (int64_t)k0 * TILE * e->ld;
      g.lda = e->ld;
      g.B = V + (int64_t)k0 * TILE * ldv;
      g.ldb = ldv;
      g.mt = nt - k1;
      g.nt = mine;
      g.k = (k1 - k0) * TILE;
      g.alpha = -1.0;
      g.beta = 1.0;
      g.klo_n = 1;
      g.krow_stride = G;
      g.krow_off = (first - k0) * TILE;
      bad.note(e, launch_gemm(e, g, 4));
    }
    // ship the finished chunk while the update above runs (it only READS the chunk, and the unpack writes
    // rows < k1 of the factor buffer's chunk columns, the update reads rows >= k1 of them): rows b < k1 of
    // columns [k0, k1) go into the upper triangle of every rank's factor buffer
    const int ncols = (k1 - k0) * TILE;
    const int64_t ldp = (int64_t)mc * TILE, elems = ldp * ncols;
    if (!bad.rc) bad.note(e, dist_pack(e, commS, V + (int64_t)k0 * TILE * ldv, ldv, e->dsend, ldp, 0, ncols, true, 0, 0, mine, 1, 0, 0, mc));
    bad.note(e, dist_all_gather(e, comm, commS, e->dsend, e->drecv, elems, &probe));  // issued whatever happened above
    if (!bad.rc) bad.note(e, dist_pack(e, commS, e->dA + (int64_t)k0 * TILE * e->ld, e->ld, e->drecv, ldp, elems, ncols, false, G, 0, 0, G, 0, k1, mc));
  }
  // alpha = U v: this rank's rows, then everybody's
  if (!bad.rc && owned > 0) {
    hipLaunchKernelGGL(urows_v_kernel, dim3(owned * 2), dim3(256), 0, mainS, V, ldv, e->dv, e->N, e->dsend + chunk_elems);
    hip_ok(hipGetLastError(), "urows_v_kernel");
  }
  bad.note(e, order_after(e, mainS, commS));
  {
    double* a_send = e->dsend + chunk_elems;  // behind the chunk staging area (a chunk may still be in flight)
    double* a_recv = e->drecv;
    // the comm stream is strictly ordered: by the time this all-gather runs every chunk has been unpacked
    bad.note(e, dist_all_gather(e, comm, commS, a_send, a_recv, ldv, &probe));
    if (!bad.rc) bad.note(e, dist_pack(e, commS, e->dalpha, e->Np, a_recv, ldv, ldv, 1, false, G, 0, 0, G, 0, nt, maxown));
  }
  bad.note(e, dist_wait(e, commS, mainS, &probe, 2));
  if (!bad.rc && e->Np > e->N) {
    hipLaunchKernelGGL(reset_pad_cols_kernel, dim3((unsigned)((e->Np + 255) / 256)), dim3(256), 0, mainS, e->dA, e->ld, e->N, e->Np);
    hip_ok(hipGetLastError(), "reset_pad_cols_kernel");
  }
  // this rank's block rows of Sigma^-1 = U U^T, packed, into the buffer V occupied; reductions over them
  std::vector<double> h;
  if (!bad.rc) bad.note(e, grad_sigma_inv_rows(e, rank, G, e->dW, ldv, true));
  if (!bad.rc) bad.note(e, grad_reduce(e, rank, G, e->dW, ldv, true, h));
  // accumulators of all ranks, summed in rank order on every rank (bit-identical results everywhere)
  if (!bad.rc) hip_ok(hipMemcpyAsync(e->dsend, e->dgpart, GACC_DOUBLES * sizeof(double), hipMemcpyDeviceToDevice, mainS), "hipMemcpyAsync");
  bad.note(e, dist_all_gather(e, comm, mainS, e->dsend, e->drecv, GACC_DOUBLES, &probe, true));
  std::vector<double> all((size_t)G * GACC_DOUBLES);
  if (!bad.rc) hip_ok(hipMemcpyAsync(all.data(), e->drecv, all.size() * sizeof(double), hipMemcpyDeviceToHost, mainS), "hipMemcpyAsync");
  tg.stop();
  rc = dist_agree(e, comm, bad.give(e), "gmb_dist_nlml");  // synchronises the main stream
  if (rc) {
    (void)hipStreamSynchronize(commS);
    e->evs.clear();
    return rc;
  }
  tm.grad_ms = tg.ms();
  ev_collect(e);
  {
    tm.dist_world = G;
    tm.dist_grad_collectives = (int64_t)probe.colls.size();
    tm.dist_grad_comm_bytes = tm.dist_grad_comm_ms = tm.dist_grad_comm_exposed_ms = 0.0;
    double on_comm_stream = 0.0;
    for (const DistProbe::Coll& c : probe.colls) {
      const double t = dist_ms(c.a, c.b);
      tm.dist_grad_comm_bytes += c.bytes;
      tm.dist_grad_comm_ms += t;
      if (c.exposed) tm.dist_grad_comm_exposed_ms += t;
      else on_comm_stream += t;
    }
    for (const DistProbe::Wait& w : probe.waits)
      if (w.kind == 2) tm.dist_grad_comm_exposed_ms += std::min(dist_ms(w.arrive, w.release), on_comm_stream);
  }
  e->have_alpha = true;
  h.assign(GACC_DOUBLES, 0.0);
  for (int q = 0; q < G; ++q)
    for (int i = 0; i < GACC_DOUBLES; ++i) h[i] += all[(size_t)q * GACC_DOUBLES + i];
  return grad_chain_rule(e, h, grad);
}

// ---- prediction: test points sharded over the ranks, results all-gathered -------------------------------
int dist_predict(gmb_engine* e, const gmb_comm* comm, const double* Xs, int64_t M, int64_t ldxs, int32_t with_noise,
                 double* mean, double* var, int32_t memspace) {
  int rc = dist_check_comm(e, comm);
  if (rc) return rc;
  const int G = comm->world, rank = comm->rank;
  rc = require_ready(e, true);
  if (!rc) rc = require_full_factor(e, "gmb_dist_predict (replicated mode)");
  if (!rc && (M < 0 || (M > 0 && (!Xs || !mean || !var)) || ldxs < e->D)) rc = fail(e, GMB_EINVAL, "bad Xs/M/ldxs/mean/var");
  if (!rc && hipSetDevice(e->device) != hipSuccess) rc = fail(e, GMB_EHIP, "hipSetDevice(%d) failed", e->device);
  auto bound = [&](int q) { return (int64_t)((double)M * q / G); };
  int64_t width = 0;
  for (int q = 0; q < G && M > 0; ++q) width = std::max(width, bound(q + 1) - bound(q));
  const int64_t lo = M > 0 ? bound(rank) : 0, cnt = M > 0 ? bound(rank + 1) - lo : 0;
  const int64_t stage = std::max<int64_t>(std::max<int64_t>(2 * width, width * std::max(e->D, 1)), 1);
  if (!rc) rc = ensure(e, &e->dsend, &e->cap_send, stage);
  if (!rc) rc = ensure(e, &e->drecv, &e->cap_recv, stage * G);
  e->coll_count = e->coll_hash = 0;  // the issue log covers this call: its closing agreement compares the ranks' sequences
  if ((rc = dist_agree(e, comm, rc, "gmb_dist_predict (set-up)"))) return rc;
  if (M == 0) return GMB_OK;  // the same M on every rank by contract
  DistDeferred bad;
  if (cnt > 0) {
    const double* xs_dev = Xs + lo * ldxs;
    int64_t ld_dev = ldxs;
    if (memspace != GMB_DEVICE) {  // stage this rank's slice (results stay on the device for the all-gather)
      const hipError_t st = hipMemcpy2DAsync(e->drecv, e->D * sizeof(double), Xs + lo * ldxs, ldxs * sizeof(double),
                                             e->D * sizeof(double), cnt, hipMemcpyHostToDevice, e->stream);
      if (st != hipSuccess) bad.note(e, fail(e, GMB_EHIP, "hipMemcpy2DAsync failed: %s", hipGetErrorString(st)));
      xs_dev = e->drecv;
      ld_dev = e->D;
    }
    if (!bad.rc) bad.note(e, gmb_predict(e, xs_dev, cnt, ld_dev, with_noise, e->dsend, e->dsend + width, GMB_DEVICE));
  }
  bad.note(e, dist_all_gather(e, comm, e->stream, e->dsend, e->drecv, 2 * width));
  if ((rc = dist_agree(e, comm, bad.give(e), "gmb_dist_predict"))) return rc;
  const hipMemcpyKind kind = memspace == GMB_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  for (int q = 0; q < G; ++q) {
    const int64_t lq = bound(q), cq = bound(q + 1) - lq;
    if (cq <= 0) continue;
    HIP_TRY(e, hipMemcpyAsync(mean + lq, e->drecv + (int64_t)q * 2 * width, cq * sizeof(double), kind, e->stream));
    HIP_TRY(e, hipMemcpyAsync(var + lq, e->drecv + (int64_t)q * 2 * width + width, cq * sizeof(double), kind, e->stream));
  }
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return GMB_OK;
}

// ---- RCCL transport ------------------------------------------------------------------------------------
struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommSplit) CommSplit = nullptr;  // optional (RCCL >= 2.18)
};

std::string g_rccl_error;

RcclApi* rccl_api(const char* path) {
  static std::mutex mu;
  static RcclApi api;
  std::lock_guard<std::mutex> lock(mu);
  if (api.lib) return &api;
  const char* names[] = {path && path[0] ? path : nullptr, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* lib = nullptr;
  for (const char* n : names)
    if (n && (lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!lib) {
    g_rccl_error = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "not found");
    return nullptr;
  }
  RcclApi a;
  a.lib = lib;
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(lib, "ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(lib, "ncclCommDestroy");
  a.AllGather = (decltype(a.AllGather))dlsym(lib, "ncclAllGather");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(lib, "ncclGetErrorString");
  a.CommCount = (decltype(a.CommCount))dlsym(lib, "ncclCommCount");
  a.CommSplit = (decltype(a.CommSplit))dlsym(lib, "ncclCommSplit");
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather) {
    g_rccl_error = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
    return nullptr;
  }
  api = a;
  return &api;
}

// Collectives of ONE communicator are executed in issue order whatever streams they were enqueued on (RCCL serialises a communicator's
// kernels through a stream of its own): a panel column's TAIL on the communication stream would hold up the next chain's SQUARE and
// PANEL gathers on the main stream -- the overlap the driver's plan is built for would exist in the emulation only.  The transport
// therefore keeps TWO communicators over the same ranks (the second split off the first, ncclCommSplit: no second unique id to hand
// round): the engine's main stream -- every gmb_dist_* call declares it (dist_check_comm) before its first collective -- uses the
// first, every other stream the second; before any engine has declared one (the Python side's handshake) everything uses the
// first.  Each communicator sees the same sequence on every rank (the plan's), so neither can pair the wrong buffers.  Without
// ncclCommSplit (or if it fails) there is one communicator and the collectives serialise: correct, slower.
struct RcclCtx {
  RcclApi* api;
  ncclComm_t comm;
  ncclComm_t comm2 = nullptr;   // collectives issued on any stream but the engine's main one
  void* main_stream = nullptr;
  bool have_main = false;
};

int32_t rccl_all_gather(void* ctx, const void* send, void* recv, int64_t count, void* stream) {
  RcclCtx* c = (RcclCtx*)ctx;
  ncclComm_t comm = (!c->have_main || stream == c->main_stream || !c->comm2) ? c->comm : c->comm2;
  const ncclResult_t r = c->api->AllGather(send, recv, (size_t)count, ncclFloat64, comm, (hipStream_t)stream);
  if (r != ncclSuccess) {
    g_rccl_error = std::string("ncclAllGather: ") + (c->api->GetErrorString ? c->api->GetErrorString(r) : "error");
    return (int32_t)r;
  }
  return 0;
}

void rccl_note_main_stream(const gmb_comm* comm, hipStream_t main_stream) {
  if (!comm || !comm->ctx || comm->all_gather != rccl_all_gather) return;  // (any other transport: its own business)
  RcclCtx* c = (RcclCtx*)comm->ctx;
  c->main_stream = (void*)main_stream;
  c->have_main = true;
}

}  // namespace

extern "C" {

int64_t gmb_dist_plan(int64_t N, int32_t rank, int32_t world, int32_t panel_blocks, gmb_dist_step* out, int64_t cap) {
  if (N < 1 || world < 1 || rank < 0 || rank >= world) return GMB_EINVAL;
  const std::vector<gmb_dist_step> plan = dist_build_plan(N, rank, world, panel_blocks);
  if (out)
    for (int64_t i = 0; i < (int64_t)plan.size() && i < cap; ++i) out[i] = plan[i];
  return (int64_t)plan.size();
}

// (gmb_dist_factorize / gmb_dist_nlml / gmb_dist_predict: engine.hip, behind dist_capacity.hpp -- they pick the mode)

const char* gmb_rccl_last_error(void) { return g_rccl_error.c_str(); }

int gmb_rccl_unique_id(const char* librccl_path, void* id128) {
  if (!id128) return GMB_EINVAL;
  RcclApi* api = rccl_api(librccl_path);
  if (!api) return GMB_EHIP;
  ncclUniqueId id;
  const ncclResult_t r = api->GetUniqueId(&id);
  if (r != ncclSuccess) {
    g_rccl_error = std::string("ncclGetUniqueId: ") + (api->GetErrorString ? api->GetErrorString(r) : "error");
    return GMB_EHIP;
  }
  static_assert(sizeof(ncclUniqueId) == 128, "RCCL unique id is 128 bytes");
  std::memcpy(id128, &id, sizeof id);
  return GMB_OK;
}

int gmb_rccl_comm_create(gmb_comm** out, const char* librccl_path, const void* id128, int32_t rank, int32_t world,
                         int32_t device) {
  if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return GMB_EINVAL;
  *out = nullptr;
  RcclApi* api = rccl_api(librccl_path);
  if (!api) return GMB_EHIP;
  if (hipSetDevice(device) != hipSuccess) return GMB_EHIP;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof id);
  ncclComm_t comm = nullptr;
  const ncclResult_t r = api->CommInitRank(&comm, world, id, rank);
  if (r != ncclSuccess) {
    g_rccl_error = std::string("ncclCommInitRank: ") + (api->GetErrorString ? api->GetErrorString(r) : "error");
    return GMB_EHIP;
  }
  RcclCtx* ctx = new RcclCtx();
  ctx->api = api;
  ctx->comm = comm;
  if (api->CommSplit && !getenv("GUMBI_RCCL_ONE_COMM")) {
    // (collective over the parent: every rank is here; same colour, rank order kept)
    ncclComm_t second = nullptr;
    const ncclResult_t r2 = api->CommSplit(comm, 0, rank, &second, nullptr);
    if (r2 == ncclSuccess && second) ctx->comm2 = second;
    else g_rccl_error = std::string("ncclCommSplit failed (") + (api->GetErrorString ? api->GetErrorString(r2) : "error") + "): one communicator, collectives of different streams serialise";
  }
  gmb_comm* c = new gmb_comm();
  c->rank = rank;
  c->world = world;
  c->ctx = ctx;
  c->all_gather = rccl_all_gather;
  *out = c;
  return GMB_OK;
}

int gmb_rccl_comm_split(const gmb_comm* c) {  // 1: two communicators (collectives of different streams overlap), 0: one
  if (!c || !c->ctx || c->all_gather != rccl_all_gather) return GMB_EINVAL;
  return ((const RcclCtx*)c->ctx)->comm2 ? 1 : 0;
}

int gmb_rccl_comm_ranks(const gmb_comm* c) {
  if (!c || !c->ctx || c->all_gather != rccl_all_gather) return GMB_EINVAL;
  const RcclCtx* ctx = (const RcclCtx*)c->ctx;
  int n = 0;
  if (!ctx->api->CommCount || ctx->api->CommCount(ctx->comm, &n) != ncclSuccess) return GMB_EHIP;
  return n;
}

void gmb_rccl_comm_destroy(gmb_comm* c) {
  if (!c) return;
  RcclCtx* ctx = (RcclCtx*)c->ctx;
  if (ctx) {
    if (ctx->comm2) (void)ctx->api->CommDestroy(ctx->comm2);
    if (ctx->comm) (void)ctx->api->CommDestroy(ctx->comm);
    delete ctx;
  }
  delete c;
}

}  // extern "C"
