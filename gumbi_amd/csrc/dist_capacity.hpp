// dist_capacity.hpp -- the multi-GPU driver's CAPACITY mode: ONE GP over the GPUs of a node with NO rank holding the whole
// factor (gmb_dist_set_mode(e, 1)).  Included by engine.hip behind dist_driver.hpp; same entry points (gmb_dist_factorize /
// gmb_dist_nlml / gmb_dist_predict), same partition, same transport (one collective: all-gather), same plan.
//
// Why: the replicated mode of dist_driver.hpp partitions the COMPUTE -- every rank still keeps a full Nr x Np factor buffer
// (80 GB at N = 100k whatever the number of GPUs), so the largest matrix is the one a single GPU holds (N ~ 165k).  The
// north star asks for "N beyond one GPU": here a rank's resident state is
//     dAown    its own block rows of L, packed                (owned * 128) x Np        N^2 / G      doubles
//     dPanel   TWO column panels of L with ALL their rows       2 x Nr x (w * 128)      what a panel all-gather delivers
//                                                                                        (gradient: two panels of L -- one arriving, one
//                                                                                        in use -- and one chunk of U, 32 columns each)
//     dW       its block rows of U = L^-T, overwritten chunk by chunk with its rows of Sigma^-1      (owned * 128) x Np
// N = 200k over 8 GPUs: 40 + 20 + 40 GB + staging (10 + 1.2 GB) ~ 112 GB of 288.
//
// How the kernels of the replicated layout are reused unchanged: a panel buffer is a WINDOW OF COLUMNS of a virtual full
// buffer with the same leading dimension -- element (row i, global column j) of panel [c0, c1) sits at
// P + i + (j - 128 c0) ld = (P - 128 c0 ld) + i + j ld -- so with e->dA pointed at that virtual base every routine that
// touches only the panel's columns of "the factor buffer" (chol_cols on the diagonal square, trsm_cols against it, the
// trailing updates' A and B operands, the unpack side of the all-gathers) runs as it is.  What changes is where a rank's OWN
// rows live (dAown, contiguous: the updates write C through GemmArgs::cblk_stride = 1, the covariance build through
// CovTileArgs::rows_packed) and that L has to be STREAMED again, panel by panel from its owners, whenever somebody needs all
// of it: the gradient's forward substitution of the identity (gmb_dist_nlml) and the sharded prediction (gmb_dist_predict) --
// N^2 / 2 doubles received per rank and pass, the same traffic as the factorisation's.  Sigma^-1 = U U^T is accumulated chunk
// by chunk (my rows of the chunk times everybody's rows of the chunk), so U is never held in full either.
//
// The reference is single-process (SURVEY.md section 8e defines the partition); the work replaced is what pm.find_MAP /
// Marginal.predict do per evaluation (gumbi/regression/pymc/GP.py:811, 845-847).
#pragma once

namespace {

// block columns per streamed panel of L (and chunk of U) in the gradient / prediction passes: 32 (a contraction of 4096) for
// large matrices, fewer for small ones, whose buffers of N x (chunk * 128) doubles would otherwise outweigh the N^2 / G they save
inline int cap_chunk(int nct) { return std::max(1, std::min(nct, std::min(DIST_INV_CHUNK, std::max(8, nct / 20)))); }

// packed block index of owned block row b
inline int cap_packed(int b, int rank, int G) { return (b - rank) / G; }

// RAII: e->dA points at the virtual base of a panel buffer while the kernels of the replicated layout run on it
struct CapVirtual {
  gmb_engine* e;
  double* saved;
  CapVirtual(gmb_engine* e_, double* panel, int c0) : e(e_), saved(e_->dA) { e->dA = panel - (int64_t)c0 * TILE * e->ld; }
  ~CapVirtual() { e->dA = saved; }
};

// this rank's packed rows + panel space of `panel_cols` block columns with all rows (the factorisation: two panels of the plan's
// width; the gradient: three chunk-wide buffers; the prediction: two)
int cap_ensure(gmb_engine* e, int G, int rank, int panel_cols) {
  const int nrt = (int)(e->Nr / TILE);
  int first, own;
  dist_owned(rank, G, 0, nrt, &first, &own);
  const int64_t ld_own = (int64_t)std::max(own, 1) * TILE;
  int rc;
  if ((rc = ensure(e, &e->dAown, &e->cap_Aown, ld_own * e->Np))) return rc;
  e->ld_own = ld_own;
  if ((rc = ensure(e, &e->dPanel, &e->cap_panel, e->Nr * (int64_t)panel_cols * TILE))) return rc;
  return GMB_OK;
}

// strided copy of a row range of a column-major matrix into a packed staging buffer (and back): rows x ncols doubles
int cap_copy2d(gmb_engine* e, hipStream_t st, double* dst, int64_t ldd, const double* src, int64_t lds, int64_t rows, int64_t ncols) {
  if (rows <= 0 || ncols <= 0) return GMB_OK;
  HIP_TRY(e, hipMemcpy2DAsync(dst, (size_t)ldd * sizeof(double), src, (size_t)lds * sizeof(double), (size_t)rows * sizeof(double),
                              (size_t)ncols, hipMemcpyDeviceToDevice, st));
  return GMB_OK;
}

// Everybody's block rows [lo, hi) of the block columns [c0, c1) of L, from the owners' packed rows into `panel` (all ranks
// take part; a rank in trouble passes ok = false and only issues the collective).
int cap_gather_panel(gmb_engine* e, const gmb_comm* comm, hipStream_t st, double* panel, int c0, int c1, int lo, int hi, bool ok,
                     DistProbe* probe, double* snd = nullptr, double* rcv = nullptr) {
  const int G = comm->world, rank = comm->rank;
  if (!snd) snd = e->dsend;  // (the factorisation's TAILs bring a staging pair of their own)
  if (!rcv) rcv = e->drecv;
  int first, count;
  dist_owned(rank, G, lo, hi, &first, &count);
  const int maxcount = dist_max_owned(G, lo, hi);
  const int64_t W = (int64_t)(c1 - c0) * TILE, ldp = (int64_t)maxcount * TILE, elems = ldp * W;
  int rc = GMB_OK;
  if (ok && count > 0)
    rc = cap_copy2d(e, st, snd, ldp, e->dAown + (int64_t)cap_packed(first, rank, G) * TILE + (int64_t)c0 * TILE * e->ld_own, e->ld_own,
                    (int64_t)count * TILE, W);
  const int rc2 = dist_all_gather(e, comm, st, snd, rcv, elems, probe);
  if (rc) return rc;
  if (rc2) return rc2;
  if (!ok) return GMB_OK;
  return dist_pack(e, st, panel, e->ld, rcv, ldp, elems, (int)W, false, G, 0, 0, G, lo, hi, maxcount);
}

// staging large enough for every all-gather of the capacity passes at panel width w (block columns)
int cap_staging(gmb_engine* e, int G, int w, int64_t extra_send = 0) {
  const int nrt = (int)(e->Nr / TILE);
  const int64_t need = std::max<int64_t>((int64_t)dist_max_owned(G, 0, nrt) * TILE * (int64_t)w * TILE, GACC_DOUBLES);
  int rc;
  if ((rc = ensure(e, &e->dsend, &e->cap_send, need + extra_send))) return rc;
  return ensure(e, &e->drecv, &e->cap_recv, std::max<int64_t>(need * G, (int64_t)GACC_DOUBLES * G));
}

// ---- factorisation ------------------------------------------------------------------------------------------------------
int cap_factorize(gmb_engine* e, const gmb_comm* comm, int panel_blocks) {
  int rc = dist_check_comm(e, comm);
  if (rc) return rc;
  const int G = comm->world, rank = comm->rank;
  gmb_timings& tm = e->tm;
  rc = require_ready(e, false);
  if (!rc && hipSetDevice(e->device) != hipSuccess) rc = fail(e, GMB_EHIP, "hipSetDevice(%d) failed", e->device);
  std::vector<gmb_dist_step> plan;
  int wmax = 1;
  if (!rc) {
    plan = dist_build_plan(e->N, rank, G, panel_blocks > 0 ? panel_blocks : (e->panel_auto ? 0 : e->panel_blocks));
    for (const gmb_dist_step& s : plan)
      if (s.op == DIST_SQUARE) wmax = std::max(wmax, s.c1 - s.c0);
    const int cwg = cap_chunk((int)(e->Np / TILE));  // (the gradient pass keeps THREE chunk-wide buffers: allocated here already)
    if (!(rc = cap_ensure(e, G, rank, std::max(2 * wmax, 3 * cwg)))) rc = cap_staging(e, G, std::max(wmax, cwg));
    // The TAILs travel on the communication stream while the next chain gathers its square and the head of its panel column: two
    // staging pairs.  The full-height pair (dsend / drecv: the gradient and prediction passes need it anyway) serves the TAILs,
    // a small second pair the chain's two short messages -- capacity mode is about memory.
    int64_t need_chain = 0;
    for (const gmb_dist_step& s : plan)
      if (s.op == DIST_SQUARE || s.op == DIST_PANEL) need_chain = std::max(need_chain, s.elems);
    if (!rc && need_chain > 0 && !(rc = ensure(e, &e->dsend2, &e->cap_send2, need_chain))) rc = ensure(e, &e->drecv2, &e->cap_recv2, need_chain * G);
  }
  e->coll_count = e->coll_hash = 0;
  if ((rc = dist_agree(e, comm, rc, "gmb_dist_factorize (set-up)"))) return rc;
  e->factored = false;
  e->factor_kind = gmb_engine::FK_CAPACITY;
  e->factor_consumed = false;
  e->have_alpha = false;
  e->notpd = -1;
  e->own_world = G;
  e->own_rank = rank;
  tm.kbuild_ms = tm.chol_ms = tm.chol_gemm_ms = tm.chol_gemm_flops = 0.0;
  tm.chol_leaf_ms = tm.chol_trsm_ms = 0.0;
  tm.chol_gemm_launches = 0;
  const int nct = (int)(e->Np / TILE), nrt = (int)(e->Nr / TILE);
  hipStream_t mainS = e->stream, bulkS = e->aux[2], commS = e->aux[0];
  e->sync_next = 0;
  e->time_next = 0;
  e->cur = mainS;
  e->chol_update_kind = 0;
  DistDeferred bad;
  DistProbe probe;
  bool tail_in_flight = false;  // a TAIL has been issued on the communication stream since the last FORK
  {
    hipError_t st = hipMemsetAsync(e->dscal, 0, 64 * sizeof(double), e->stream);
    if (st == hipSuccess) st = hipMemsetAsync(e->dinfo, 0, sizeof(int32_t), e->stream);
    if (st != hipSuccess) bad.note(e, fail(e, GMB_EHIP, "hipMemsetAsync failed: %s", hipGetErrorString(st)));
  }
  const int64_t panel_elems = e->Nr * (int64_t)wmax * TILE;  // (two panels of the plan's width at the head of dPanel)
  int panel_idx = -1;  // panel p lives in dPanel + (p & 1) * panel_elems: U2(p) reads it while the chain of p + 1 fills the other
  auto panel_of = [&](int c0) {  // (panels start at multiples of the plan's width: find the index by scanning the plan once)
    int p = 0;
    for (const gmb_dist_step& s : plan)
      if (s.op == DIST_SQUARE) {
        if (s.c0 == c0) return p;
        ++p;
      }
    return 0;
  };
  PhaseTimer tk(e);
  PhaseTimer* tc = nullptr;
  for (const gmb_dist_step& s : plan) {
    const int64_t W = (int64_t)(s.c1 - s.c0) * TILE;
    const bool has_collective = s.op == DIST_SQUARE || s.op == DIST_PANEL || s.op == DIST_TAIL;
    bool gathered = false;
    rc = GMB_OK;
    if (s.op == DIST_SQUARE) {
      ++probe.group;
      ++panel_idx;
    }
    // the TAILs' staging pair (dsend / drecv) is free once the previous TAIL has been unpacked; this one leaves once the main stream
    // has solved the rows (both orders are issued whatever this rank's state)
    if (s.op == DIST_TAIL) {
      bad.note(e, order_after(e, commS, mainS));
      bad.note(e, order_after(e, mainS, commS));
    }
    double* panel = e->dPanel + (int64_t)((s.op == DIST_UPDATE ? panel_of(s.c0) : std::max(panel_idx, 0)) & 1) * panel_elems;
    if (!bad.rc) switch (s.op) {
      case DIST_KBUILD: {  // this rank's block rows of the lower triangle (y row and padding included), packed
        CovTileArgs a{};
        a.p = e->cp;
        a.rows = train_set(e);
        a.cols = train_set(e);
        a.out = e->dAown;
        a.ldo = e->ld_own;
        a.ti = nrt;
        a.tj = nct;
        a.mode = COV_TRAIN;
        a.lower_only = 1;
        a.row_first = rank;
        a.row_stride = G;
        a.rows_packed = 1;
        a.y = e->dy;
        if ((rc = launch_cov(e, a))) break;
        for (size_t t = 1; t < e->terms.size() && !rc; ++t) {
          if ((rc = prep_points(e, e->dX, e->N, e->D, e->Nr, e->xs, e->xl, e->cat, &e->terms[t].pa))) break;
          a.p = e->terms[t].cp;
          a.accumulate = 1;
          rc = launch_cov(e, a);
        }
        if (!rc && e->terms.size() > 1) rc = prep_points(e, e->dX, e->N, e->D, e->Nr, e->xs, e->xl, e->cat, &e->terms[0].pa);
        break;
      }
      case DIST_SQUARE: {  // the diagonal square's block rows from their owners, then every rank factors it
        gathered = true;
        if ((rc = cap_gather_panel(e, comm, mainS, panel, s.c0, s.c1, s.lo, s.hi, true, &probe, e->dsend2, e->drecv2))) break;
        CapVirtual v(e, panel, s.c0);
        e->cur = mainS;
        rc = chol_cols(e, s.c0, s.c1, s.c1);
        // the owners' copies of the square's rows become L too (they are read again when L is streamed)
        int f, c;
        dist_owned(rank, G, s.lo, s.hi, &f, &c);
        for (int t = 0; t < c && !rc; ++t) {
          const int b = f + t * G;
          rc = cap_copy2d(e, mainS, e->dAown + (int64_t)cap_packed(b, rank, G) * TILE + (int64_t)s.c0 * TILE * e->ld_own, e->ld_own,
                          panel + (int64_t)b * TILE, e->ld, TILE, W);
        }
        break;
      }
      case DIST_SOLVE: {  // my rows below the square: solved in place in dAown, and copied into the panel buffer (U1 reads them there
                          // before the TAIL has delivered everybody's)
        if (s.count <= 0) break;
        CapVirtual v(e, panel, s.c0);
        e->cur = mainS;
        if ((rc = trsm_cols(e, e->dAown + (int64_t)cap_packed(s.first, rank, G) * TILE, e->ld_own, s.count, s.c0, s.c1, 2, 5))) break;
        rc = dist_pack(e, mainS, panel, e->ld, e->dAown + (int64_t)cap_packed(s.first, rank, G) * TILE + (int64_t)s.c0 * TILE * e->ld_own, e->ld_own, 0,
                       (int)W, false, 0, s.first, s.count, G, 0, 0, s.maxcount);
        break;
      }
      case DIST_PANEL: {  // the head of the panel column: everybody's solved rows [lo, hi) into the panel buffer (main stream)
        gathered = true;
        rc = cap_gather_panel(e, comm, mainS, panel, s.c0, s.c1, s.lo, s.hi, true, &probe, e->dsend2, e->drecv2);
        break;
      }
      case DIST_TAIL: {  // ... and the rest of it, on the communication stream through the full-height staging pair
        gathered = true;
        rc = cap_gather_panel(e, comm, commS, panel, s.c0, s.c1, s.lo, s.hi, true, &probe);
        tail_in_flight = true;
        break;
      }
      case DIST_UPDATE: {  // my rows of the trailing columns [lo, hi) -= (my rows of the panel) (the panel's rows lo .. hi)^T
        if (s.count <= 0) break;
        const double* pv = panel - (int64_t)s.c0 * TILE * e->ld;  // virtual base of panel [c0, c1)
        GemmArgs g{};
        g.C = e->dAown + (int64_t)cap_packed(s.first, rank, G) * TILE + (int64_t)s.lo * TILE * e->ld_own;
        g.ldc = e->ld_own;
        g.cblk_stride = 1;
        g.A = pv + (int64_t)s.lo * TILE + (int64_t)s.c0 * TILE * e->ld;
        g.lda = e->ld;
        g.B = pv + (int64_t)s.first * TILE + (int64_t)s.c0 * TILE * e->ld;
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
    if (has_collective && !gathered) {
      if (s.op == DIST_TAIL) bad.note(e, dist_all_gather(e, comm, commS, e->dsend, e->drecv, s.elems));
      else bad.note(e, dist_all_gather(e, comm, mainS, e->dsend2, e->drecv2, s.elems));
    }
    // v = L^-1 y is row N of the factor: its entries for this panel's columns are complete once the piece of the panel column
    // that holds the last block row (or, for the last panel, the square itself) is in the panel buffer -- copied on that
    // piece's stream (the main stream waits for the communication stream before anybody reads v)
    if (!bad.rc && (((s.op == DIST_PANEL || s.op == DIST_TAIL) && s.hi >= nrt) || (s.op == DIST_SQUARE && s.hi >= nrt))) {
      const hipError_t st = hipMemcpy2DAsync(e->dv + (int64_t)s.c0 * TILE, sizeof(double), panel + e->N, (size_t)e->ld * sizeof(double),
                                             sizeof(double), (size_t)std::min<int64_t>(W, e->N - (int64_t)s.c0 * TILE), hipMemcpyDeviceToDevice,
                                             s.op == DIST_TAIL ? commS : mainS);
      if (st != hipSuccess) bad.note(e, fail(e, GMB_EHIP, "hipMemcpy2DAsync failed: %s", hipGetErrorString(st)));
    }
    if (s.op == DIST_KBUILD) {
      tk.stop();
      tc = new PhaseTimer(e);
    }
  }
  bad.note(e, order_after(e, commS, mainS));  // the last TAIL (the y row travels in it)
  e->cur = mainS;
  double hs[2] = {0.0, 0.0};
  int32_t info = 0;
  if (!bad.rc) {
    // |v|^2 with the single engine's fixed-order reduction (row "N" of a 1-row matrix: the assembled v itself)
    hipLaunchKernelGGL(extract_v_kernel, dim3(EXTRACT_V_BLOCKS), dim3(TILE), 0, e->stream, e->dv - e->N, (int64_t)1, e->N, e->dv, e->dscal + 1);
    if (tc) tc->stop();
    hipError_t st = hipGetLastError();
    if (st == hipSuccess) st = hipMemcpyAsync(hs, e->dscal, 2 * sizeof(double), hipMemcpyDeviceToHost, e->stream);
    if (st == hipSuccess) st = hipMemcpyAsync(&info, e->dinfo, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream);
    if (st == hipSuccess) st = hipStreamSynchronize(e->stream);
    if (st != hipSuccess) bad.note(e, fail(e, GMB_EHIP, "distributed factorisation failed: %s", hipGetErrorString(st)));
  } else if (tc) {
    tc->stop();
  }
  double shared[3] = {(double)info, hs[0], hs[1]};
  rc = dist_agree(e, comm, bad.give(e), "gmb_dist_factorize", shared, 3, &tm.dist_lockstep_repairs);
  info = (int32_t)shared[0];
  hs[0] = shared[1];
  hs[1] = shared[2];
  if (rc) {
    (void)hipStreamSynchronize(bulkS);
    e->evs.clear();
    delete tc;
    return rc;
  }
  tm.kbuild_ms = tk.ms();
  tm.chol_ms = tc ? tc->ms() : 0.0;
  delete tc;
  tm.kbuild_bytes = (8.0 * (double)e->N * (double)(e->N + 1) / 2.0) / G + 8.0 * (double)e->N * (double)(e->spec.n_cont + 1);
  ev_collect(e);
  tm.dist_world = G;
  tm.dist_chol_collectives = (int64_t)probe.colls.size();
  tm.dist_chol_comm_bytes = tm.dist_chol_comm_ms = tm.dist_chol_comm_exposed_ms = 0.0;
  tm.dist_chol_main_wait_ms = tm.dist_chol_bulk_wait_ms = 0.0;
  {
    std::vector<double> chain_comm((size_t)probe.group + 2, 0.0);
    for (const DistProbe::Coll& c : probe.colls) {
      const double t = dist_ms(c.a, c.b);
      tm.dist_chol_comm_bytes += c.bytes;
      tm.dist_chol_comm_ms += t;
      if (c.group >= 0) chain_comm[(size_t)c.group] += t;
    }
    tm.dist_chol_comm_exposed_ms += dist_exposed_from_probes(probe, chain_comm, &tm.dist_chol_main_wait_ms, &tm.dist_chol_bulk_wait_ms);
  }
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

// padding columns [n, npad) of this rank's block rows of U = L^-T: identity (they picked up the y row's products)
__global__ void cap_reset_pad_cols_kernel(double* V, int64_t ldv, int64_t n, int64_t npad, int first, int stride, int owned) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // packed row
  if (r >= (int64_t)owned * TILE) return;
  const int64_t gi = ((int64_t)first + (r >> 7) * stride) * TILE + (r & 127);
  for (int64_t k = n; k < npad; ++k) V[r + k * ldv] = (gi == k) ? 1.0 : 0.0;
}

// out[r] = sum over the chunks of part[c * ld + r], in chunk order (same bits run to run)
__global__ void cap_sum_chunks_kernel(const double* part, int nchunks, int64_t ld, double* out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= ld) return;
  double s = 0.0;
  for (int c = 0; c < nchunks; ++c) s += part[(int64_t)c * ld + r];
  out[r] = s;
}

// ---- gradient -----------------------------------------------------------------------------------------------------------
int cap_nlml(gmb_engine* e, const gmb_comm* comm, double* nlml, double* grad) {
  int rc = dist_check_comm(e, comm);
  if (rc) return rc;
  if (!grad) {
    if ((rc = require_ready(e, true))) return rc;
    if (!nlml) return fail(e, GMB_EINVAL, "nlml output pointer is null");
    *nlml = 0.5 * (double)e->N * std::log(2.0 * M_PI) + e->logdet + 0.5 * e->vnorm2;
    return GMB_OK;
  }
  const int G = comm->world, rank = comm->rank;
  rc = require_ready(e, true);
  if (!rc) rc = require_capacity_factor(e, "gmb_dist_nlml");
  if (!rc && !nlml) rc = fail(e, GMB_EINVAL, "nlml output pointer is null");
  if (!rc && (e->own_world != G || e->own_rank != rank)) rc = fail(e, GMB_EINVAL, "the resident rows were factored for another rank / world size");
  if (!rc && hipSetDevice(e->device) != hipSuccess) rc = fail(e, GMB_EHIP, "hipSetDevice(%d) failed", e->device);
  const int nt = (int)(e->Np / TILE), nrt = (int)(e->Nr / TILE);
  int first = 0, owned = 0;
  dist_owned(rank, G, 0, nt, &first, &owned);
  const int maxown = dist_max_owned(G, 0, nt);
  const int64_t ldv = (int64_t)maxown * TILE;
  const int cw = cap_chunk(nt);
  if (!rc) rc = ensure(e, &e->dW, &e->cap_W, ldv * e->Np);
  if (!rc) rc = grad_workspace(e);
  if (!rc) rc = cap_ensure(e, G, rank, 3 * cw);  // two panels of L (one arriving, one in use) + the chunk of U
  const int nchunks = (nt + cw - 1) / cw;
  if (!rc) rc = cap_staging(e, G, cw, ldv * (int64_t)(nchunks + 1));  // (+ this rank's alpha rows, chunk by chunk and summed)
  e->coll_count = e->coll_hash = 0;
  if ((rc = dist_agree(e, comm, rc, "gmb_dist_nlml (set-up)"))) return rc;
  *nlml = 0.5 * (double)e->N * std::log(2.0 * M_PI) + e->logdet + 0.5 * e->vnorm2;
  gmb_timings& tm = e->tm;
  tm.grad_ms = tm.grad_gemm_ms = tm.grad_gemm_flops = 0.0;
  // V = this rank's block rows of U; Z = its block rows of Sigma^-1 IN THE SAME BUFFER: a chunk's columns of V are dead once
  // the chunk has been sent and applied to the later columns, and the chunk's contribution to Sigma^-1 only touches columns
  // left of the chunk's end -- the freshly dead ones are zeroed first, the ones before already hold partial sums
  double* V = e->dW;
  double* Z = e->dW;
  // Buffers (chunk-wide, all rows): TWO for the streamed panels of L -- panel c + 1 arrives on the communication stream while
  // chunk c is solved and applied -- and one for the chunk of U with everybody's rows.
  const int64_t pbuf = e->Nr * (int64_t)cw * TILE;
  double* Lbuf[2] = {e->dPanel, e->dPanel + pbuf};
  double* UC = e->dPanel + 2 * pbuf;  // (leading dimension Np)
  hipStream_t mainS = e->stream, commS = e->aux[0];
  PhaseTimer tg(e);
  e->sync_next = 0;
  e->time_next = 0;
  e->cur = mainS;
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
  double* a_send = e->dsend + (int64_t)dist_max_owned(G, 0, nrt) * TILE * (int64_t)cw * TILE;  // behind the chunk staging area
  // Two streams (the replicated mode's scheme, dist_nlml): EVERY collective of the loop is issued on the communication stream,
  // in one order on every rank -- L(0) | L(1), U(0) | L(2), U(1) | ... | U(last) -- with its pack / unpack kernels around it (one
  // staging pair serves them all: the stream serialises them); the main stream solves and multiplies, and waits for the other one
  // twice per chunk: for the panel of L before it starts the chunk (gathered while the chunk before was computed), and for
  // the chunk of U before it adds the chunk's share of Sigma^-1 (gathered while the chunk was applied to the later columns).
  bad.note(e, order_after(e, mainS, commS));  // (staging buffers and panels may still be read by an earlier call's kernels)
  bad.note(e, cap_gather_panel(e, comm, commS, Lbuf[0], 0, std::min(cw, nt), 0, nrt, !bad.rc, &probe));
  for (int k0 = 0, ci = 0; k0 < nt; k0 += cw, ++ci) {
    const int k1 = std::min(k0 + cw, nt);
    double* panel = Lbuf[ci & 1];
    // 1. the column panel [k0, k1) of L, rows k0 .. from their owners: on its way since the chunk before
    bad.note(e, dist_wait(e, commS, mainS, &probe, 2));
    // ... and the next one leaves now, into the buffer the chunk before has finished with (everything the main stream has been
    // given so far: its update read that buffer, its Sigma^-1 product the chunk of U that the gather after it will overwrite)
    if (k1 < nt) {
      bad.note(e, order_after(e, mainS, commS));
      bad.note(e, cap_gather_panel(e, comm, commS, Lbuf[(ci + 1) & 1], k1, std::min(k1 + cw, nt), k1, nrt, !bad.rc, &probe));
    }
    CapVirtual v(e, panel, k0);
    int f2, mine;
    dist_owned(rank, G, 0, k1, &f2, &mine);
    const int mc = dist_max_owned(G, 0, k1);
    // 2. this rank's rows of U in these columns
    if (!bad.rc && mine > 0) bad.note(e, trsm_cols(e, V, ldv, mine, k0, k1, 4, 6, first, G));
    if (!bad.rc && k1 == nt && e->Np > e->N && owned > 0) {  // (the last chunk holds the padding columns)
      hipLaunchKernelGGL(cap_reset_pad_cols_kernel, dim3((unsigned)(((int64_t)owned * TILE + 255) / 256)), dim3(256), 0, mainS, V, ldv, e->N, e->Np,
                         first, G, owned);
      hip_ok(hipGetLastError(), "cap_reset_pad_cols_kernel");
    }
    // 4. everybody's rows b < k1 of the chunk's columns of U: the chunk is final, off it goes while 3. runs
    const int ncols = (k1 - k0) * TILE;
    const int64_t ldp = (int64_t)mc * TILE, elems = ldp * ncols;
    bad.note(e, order_after(e, mainS, commS));
    if (!bad.rc) bad.note(e, dist_pack(e, commS, V + (int64_t)k0 * TILE * ldv, ldv, e->dsend, ldp, 0, ncols, true, 0, 0, mine, 1, 0, 0, mc));
    bad.note(e, dist_all_gather(e, comm, commS, e->dsend, e->drecv, elems, &probe));
    if (!bad.rc) bad.note(e, dist_pack(e, commS, UC, e->Np, e->drecv, ldp, elems, ncols, false, G, 0, 0, G, 0, k1, mc));
    // 3. the chunk's contribution to this rank's later columns of U
    if (!bad.rc && mine > 0 && k1 < nt) {
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
    // alpha = U v, this chunk's columns (the chunk's columns of V are overwritten below)
    if (!bad.rc && owned > 0) {
      const int64_t nvalid = std::min<int64_t>((int64_t)(k1 - k0) * TILE, e->N - (int64_t)k0 * TILE);
      double* part = a_send + (int64_t)(1 + k0 / cw) * ldv;
      hip_ok(hipMemsetAsync(part, 0, (size_t)ldv * sizeof(double), mainS), "hipMemsetAsync");
      if (nvalid > 0) {
        hipLaunchKernelGGL(urows_v_kernel, dim3(owned * 2), dim3(256), 0, mainS, V + (int64_t)k0 * TILE * ldv, ldv, e->dv + (int64_t)k0 * TILE, nvalid, part);
        hip_ok(hipGetLastError(), "urows_v_kernel");
      }
    }
    // 5. my rows of Sigma^-1 += (my rows of the chunk) (everybody's rows of the chunk)^T, lower triangle only; the chunk's own
    //    columns of the shared buffer held U until the pack above read them
    bad.note(e, dist_wait(e, commS, mainS, &probe, 2));
    if (!bad.rc) hip_ok(hipMemsetAsync(Z + (int64_t)k0 * TILE * ldv, 0, (size_t)ldv * ncols * sizeof(double), mainS), "hipMemsetAsync");
    if (!bad.rc && mine > 0) {
      GemmArgs g{};
      g.C = Z;
      g.ldc = ldv;
      g.cblk_stride = 1;
      g.A = UC;
      g.lda = e->Np;
      g.B = UC + (int64_t)first * TILE;
      g.ldb = e->Np;
      g.mt = k1;
      g.nt = mine;
      g.k = ncols;
      g.klo_n = 1;
      g.krow_off = (first - k0) * TILE;
      g.tri = 1;
      g.tri_off = first * TILE;
      g.nblk_stride = G;
      g.alpha = 1.0;
      g.beta = 1.0;
      bad.note(e, launch_gemm(e, g, 4));
    }
  }
  // alpha = U v: this rank's rows (the chunks' parts in chunk order), then everybody's
  if (!bad.rc) {
    hip_ok(hipMemsetAsync(a_send, 0, (size_t)ldv * sizeof(double), mainS), "hipMemsetAsync");
    if (owned > 0) {
      hipLaunchKernelGGL(cap_sum_chunks_kernel, dim3((unsigned)((ldv + 255) / 256)), dim3(256), 0, mainS, a_send + ldv, nchunks, ldv, a_send);
      hip_ok(hipGetLastError(), "cap_sum_chunks_kernel");
    }
  }
  bad.note(e, dist_all_gather(e, comm, mainS, a_send, e->drecv, ldv, &probe, true));  // (behind the last wait: the other stream is idle)
  if (!bad.rc) bad.note(e, dist_pack(e, mainS, e->dalpha, e->Np, e->drecv, ldv, ldv, 1, false, G, 0, 0, G, 0, nt, maxown));
  std::vector<double> h;
  if (!bad.rc) bad.note(e, grad_reduce(e, rank, G, Z, ldv, true, h));
  if (!bad.rc) hip_ok(hipMemcpyAsync(e->dsend, e->dgpart, GACC_DOUBLES * sizeof(double), hipMemcpyDeviceToDevice, mainS), "hipMemcpyAsync");
  bad.note(e, dist_all_gather(e, comm, mainS, e->dsend, e->drecv, GACC_DOUBLES, &probe, true));
  std::vector<double> all((size_t)G * GACC_DOUBLES);
  if (!bad.rc) hip_ok(hipMemcpyAsync(all.data(), e->drecv, all.size() * sizeof(double), hipMemcpyDeviceToHost, mainS), "hipMemcpyAsync");
  tg.stop();
  rc = dist_agree(e, comm, bad.give(e), "gmb_dist_nlml");
  if (rc) {
    (void)hipStreamSynchronize(commS);
    e->evs.clear();
    return rc;
  }
  tm.grad_ms = tg.ms();
  ev_collect(e);
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
  // what the main stream actually waited for the communication stream (never more than was spent there)
  double waited = 0.0;
  for (const DistProbe::Wait& w : probe.waits)
    if (w.kind == 2) waited += dist_ms(w.arrive, w.release);
  tm.dist_grad_comm_exposed_ms += std::min(waited, on_comm_stream);
  e->have_alpha = true;
  h.assign(GACC_DOUBLES, 0.0);
  for (int q = 0; q < G; ++q)
    for (int i = 0; i < GACC_DOUBLES; ++i) h[i] += all[(size_t)q * GACC_DOUBLES + i];
  return grad_chain_rule(e, h, grad);
}

// ---- prediction ---------------------------------------------------------------------------------------------------------
// V <- V L^-T for this rank's test points with L streamed panel by panel from its owners (every rank takes part in every
// gather, whether it has test points in this pass or not): within a panel the recursion of trsm_cols, behind it the panel's
// rows below the square applied to all later columns.
int cap_solve(gmb_engine* e, const gmb_comm* comm, double* V, int64_t ldv, int ntm, DistDeferred& bad) {
  const int nct = (int)(e->Np / TILE), nrt = (int)(e->Nr / TILE);
  const int cw = cap_chunk(nct);
  hipStream_t st = e->stream, commS = e->aux[0];
  e->cur = st;
  // panel c + 1 travels on the communication stream while panel c is solved against and applied (two buffers; the gathers of a
  // pass are issued on that one stream, in panel order, on every rank)
  const int64_t pbuf = e->Nr * (int64_t)cw * TILE;
  double* Lbuf[2] = {e->dPanel, e->dPanel + pbuf};
  bad.note(e, order_after(e, st, commS));
  bad.note(e, cap_gather_panel(e, comm, commS, Lbuf[0], 0, std::min(cw, nct), 0, nrt, !bad.rc, nullptr));
  for (int c0 = 0, ci = 0; c0 < nct; c0 += cw, ++ci) {
    const int c1 = std::min(c0 + cw, nct);
    bad.note(e, order_after(e, commS, st));  // panel [c0, c1) has arrived
    if (c1 < nct) {
      bad.note(e, order_after(e, st, commS));  // (the update before last has finished with the buffer the next panel goes into)
      bad.note(e, cap_gather_panel(e, comm, commS, Lbuf[(ci + 1) & 1], c1, std::min(c1 + cw, nct), c1, nrt, !bad.rc, nullptr));
    }
    if (bad.rc || ntm <= 0) continue;
    CapVirtual v(e, Lbuf[ci & 1], c0);
    bad.note(e, trsm_cols(e, V, ldv, ntm, c0, c1, 3, 6));
    if (!bad.rc && c1 < nct) {
      GemmArgs g{};
      g.C = V + (int64_t)c1 * TILE * ldv;
      g.ldc = ldv;
      g.A = e->dA + (int64_t)c1 * TILE + (int64_t)c0 * TILE * e->ld;
      g.lda = e->ld;
      g.B = V + (int64_t)c0 * TILE * ldv;
      g.ldb = ldv;
      g.mt = nct - c1;
      g.nt = ntm;
      g.k = (c1 - c0) * TILE;
      g.alpha = -1.0;
      g.beta = 1.0;
      bad.note(e, launch_gemm(e, g, 3));
    }
  }
  bad.note(e, order_after(e, commS, st));  // (nothing of this pass is left on the other stream when the caller goes on)
  return GMB_OK;
}

int cap_predict(gmb_engine* e, const gmb_comm* comm, const double* Xs, int64_t M, int64_t ldxs, int32_t with_noise, double* mean,
                double* var, int32_t memspace) {
  int rc = dist_check_comm(e, comm);
  if (rc) return rc;
  const int G = comm->world, rank = comm->rank;
  rc = require_ready(e, true);
  if (!rc) rc = require_capacity_factor(e, "gmb_dist_predict");
  if (!rc && (e->own_world != G || e->own_rank != rank)) rc = fail(e, GMB_EINVAL, "the resident rows were factored for another rank / world size");
  if (!rc && (M < 0 || (M > 0 && (!Xs || !mean || !var)) || ldxs < e->D)) rc = fail(e, GMB_EINVAL, "bad Xs/M/ldxs/mean/var");
  if (!rc && hipSetDevice(e->device) != hipSuccess) rc = fail(e, GMB_EHIP, "hipSetDevice(%d) failed", e->device);
  auto bound = [&](int q) { return (int64_t)((double)M * q / G); };
  int64_t width = 0;
  for (int q = 0; q < G && M > 0; ++q) width = std::max(width, bound(q + 1) - bound(q));
  const int64_t lo = M > 0 ? bound(rank) : 0, cnt = M > 0 ? bound(rank + 1) - lo : 0;
  const int cw = cap_chunk((int)(e->Np / TILE));
  // results of every pass are kept in a staging area of their own (the panel gathers of the passes reuse dsend / drecv)
  double* res = nullptr;
  int64_t cap_res = 0;
  if (!rc) rc = cap_ensure(e, G, rank, 2 * cw);  // two panels of L: one arriving, one in use
  if (!rc) rc = cap_staging(e, G, cw);
  if (!rc) rc = ensure(e, &res, &cap_res, std::max<int64_t>(2 * width * (1 + G) + width * std::max(e->D, 1), 1));
  // gmb_predict's own workspaces for the largest shard's M-tile, BEFORE the ranks agree to start: an allocation that fails
  // inside a pass would take its rank out of gmb_predict ahead of the solve hook, i.e. ahead of the pass's all-gathers
  if (!rc && width > 0) rc = predict_workspace(e, predict_tile_rows(e, width));
  e->coll_count = e->coll_hash = 0;
  rc = dist_agree(e, comm, rc, "gmb_dist_predict (set-up)");
  if (rc || M == 0) {
    release(e, res);
    return rc;
  }
  // every rank runs the same number of passes over L: the largest shard in slices of the prediction's own M-tile
  int64_t mt_max = (int64_t)(8.0 * 1024 * 1024 * 1024 / 8.0 / (double)e->Np);
  mt_max = std::max<int64_t>(TILE, std::min<int64_t>(mt_max / TILE * TILE, 32768));
  const int64_t passes = (width + mt_max - 1) / mt_max;
  double* my_mean = res;
  double* my_var = res + width;
  double* all_res = res + 2 * width;
  double* xs_stage = all_res + 2 * width * G;
  DistDeferred bad;
  for (int64_t p = 0; p < passes; ++p) {
    const int64_t m0 = p * mt_max, mc = std::max<int64_t>(0, std::min<int64_t>(mt_max, cnt - m0));
    if (mc > 0 && !bad.rc) {
      const double* xs_dev = Xs + (lo + m0) * ldxs;
      int64_t ld_dev = ldxs;
      if (memspace != GMB_DEVICE) {
        const hipError_t st = hipMemcpy2DAsync(xs_stage, e->D * sizeof(double), Xs + (lo + m0) * ldxs, ldxs * sizeof(double),
                                               e->D * sizeof(double), mc, hipMemcpyHostToDevice, e->stream);
        if (st != hipSuccess) bad.note(e, fail(e, GMB_EHIP, "hipMemcpy2DAsync failed: %s", hipGetErrorString(st)));
        xs_dev = xs_stage;
        ld_dev = e->D;
      }
      // exactly ONE visit of the hook per pass (mc <= the M-tile): a gmb_predict that leaves before it -- a failed copy or
      // launch ahead of the solve -- is followed by the pass's gathers alone, so that this rank's peers are never left waiting
      bool solved = false;
      e->solve_hook = [&](double* V, int64_t ldv, int ntm) {
        solved = true;
        return cap_solve(e, comm, V, ldv, ntm, bad);
      };
      if (!bad.rc) bad.note(e, gmb_predict(e, xs_dev, mc, ld_dev, with_noise, my_mean + m0, my_var + m0, GMB_DEVICE));
      e->solve_hook = nullptr;
      if (!solved) cap_solve(e, comm, nullptr, TILE, 0, bad);
    } else {
      cap_solve(e, comm, nullptr, TILE, 0, bad);  // nothing of mine in this pass: the gathers only
    }
  }
  bad.note(e, dist_all_gather(e, comm, e->stream, res, all_res, 2 * width));
  rc = dist_agree(e, comm, bad.give(e), "gmb_dist_predict");
  if (!rc) {
    const hipMemcpyKind kind = memspace == GMB_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    for (int q = 0; q < G && !rc; ++q) {
      const int64_t lq = bound(q), cq = bound(q + 1) - lq;
      if (cq <= 0) continue;
      if (hipMemcpyAsync(mean + lq, all_res + (int64_t)q * 2 * width, cq * sizeof(double), kind, e->stream) != hipSuccess ||
          hipMemcpyAsync(var + lq, all_res + (int64_t)q * 2 * width + width, cq * sizeof(double), kind, e->stream) != hipSuccess)
        rc = fail(e, GMB_EHIP, "copying the gathered predictions failed");
    }
    if (hipStreamSynchronize(e->stream) != hipSuccess && !rc) rc = fail(e, GMB_EHIP, "hipStreamSynchronize failed");
  }
  release(e, res);
  return rc;
}

}  // namespace
