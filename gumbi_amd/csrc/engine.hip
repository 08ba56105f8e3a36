// engine.hip -- host side of libgumbi_hip.so: the C ABI of include/gumbi_hip.h.
//
// One engine = one HIP device + one stream + the resident state of one GP:
//   raw inputs X (N x D), y; prepared coordinates (scaled / shifted / categorical, SoA);
//   the covariance / factor buffer A (column-major, (N+1 rounded up to 128) rows x (N rounded up
//   to 128) columns) whose row N carries y and, after factorisation, v = L^-1 y;
//   inv(L_kk) for every 128 x 128 diagonal block; predict workspaces.
//
// Cholesky: recursive over block columns -- chol(c0,c1) = chol(c0,mid); trailing update
// A[mid:END, mid:c1] -= L[mid:END, c0:mid] L[mid:c1, c0:mid]^T on f64 MFMA; chol(mid,c1).
// Most flops land in GEMMs with k >= N/4, so the update is MFMA- rather than HBM-bound.
// Leaves: potrf_leaf (factor + invert a diagonal block) then the panel solve as a GEMM with
// inv(L_kk).  Predict: V = K(X*,X) L^-T by the same recursion on V's columns.
//
// There is no CPU fallback anywhere in this file: without a HIP device every entry point
// returns GMB_ENODEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <functional>
#include <unordered_map>
#include <mutex>
#include <memory>
#include <vector>

#include "../../include/gumbi_hip.h"
#include "covariance.hpp"
#include "gemm_f64.hpp"
#include "gradient.hpp"
#include "potrf_leaf.hpp"
#include "trsm_strip.hpp"
#include "chol_tiles.hpp"
#include "eval_tiles.hpp"
#include "predict_form.hpp"

using namespace gmb;

namespace {

constexpr int DIST_MAX_WORLD = 64;   // ranks one GP can be spread over (status words of gmb_dist_*)
constexpr int DIST_MAX_PAYLOAD = 7;  // scalars that travel with a status word (dist_agree)
constexpr int DIST_HEADER = 3;       // words in front of them: status | collectives issued so far | hash of their sequence

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

struct EventPair {
  hipEvent_t a, b;
  int kind;  // 0 chol in-panel gemm, 7 chol bulk trailing update, 1 leaf, 2 trsm gemm, 5 / 6 strips, 3 predict gemm, 4 grad gemm
  double flops;
  int mt, nt, k, flags;
  bool masked = false;  // launched on the CU-masked bulk stream
};

}  // namespace

constexpr int GACC_REGION = 64 + MAX_TABS * GMB_MAX_LEVELS * GMB_MAX_LEVELS + 64;  // gradient accumulators of one covariance term
constexpr int GACC_DOUBLES = (1 + GMB_MAX_COREG) * GACC_REGION;  // additive models: one region per term
// the engine's scalar block, ONE allocation (one memset per evaluation): [0, 64) scalars ([0] log-det, [1] |v|^2, [2 ..] the partials
// and arrival counter of the |v|^2 sum) | [64, 72) the failure index (int32) | [72, ...) the gradient accumulators + 64 of scratch
constexpr int SCAL_INFO_AT = 64, SCAL_GACC_AT = 72, SCAL_DOUBLES = SCAL_GACC_AT + GACC_DOUBLES + 64;

struct gmb_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;

  // data
  int64_t N = 0, Np = 0, Nr = 0, ld = 0;
  int32_t D = 0;
  double* dX = nullptr;
  double* dy = nullptr;
  double* dA = nullptr;
  double* dDinv16 = nullptr; // 8 x (16 x 16) sub-block inverses per diagonal block (triangular solves)
  int64_t cap_A = 0, cap_dinv16 = 0;

  // kernel + parameters
  gmb_kernel_spec spec{};
  bool have_spec = false, have_theta = false;
  std::vector<double> theta;
  int nc_pad = 1;
  double* xs = nullptr;
  double* xl = nullptr;
  int32_t* cat = nullptr;
  int64_t cap_pts = 0;
  double* dtabs = nullptr;
  double* dnoise = nullptr;
  CovParams cp{};
  PrepArgs prep_proto{};
  // Additive models: term 0 (the global kernel) is (prep_proto, cp); term 1 + j is the kernel of
  // coregion dim j.  Covariances are built and differentiated term by term (accumulating passes).
  struct Term {
    PrepArgs pa;
    CovParams cp;
    double eta;
    int n_tab_acc;  // tables whose partials this term's gradient pass produces
  };
  std::vector<Term> terms;
  std::vector<double> htabs;  // host copies (gradient chain rule)
  std::vector<double> hnoise;

  // factorisation results
  double* dscal = nullptr;  // the scalar block (SCAL_DOUBLES): [0] logdet, [1] |v|^2, [2..] partials of that sum
  int32_t* dinfo = nullptr; // = dscal + SCAL_INFO_AT (and dgpart = dscal + SCAL_GACC_AT)
  double* dv = nullptr;
  bool factored = false;
  // which path produced the resident factorisation (meaningful while `factored`): the single-engine and the replicated
  // multi-GPU paths leave the COMPLETE factor in dA; the capacity driver leaves this rank's rows in dAown and nothing in dA
  enum { FK_NONE = 0, FK_SINGLE = 1, FK_REPLICATED = 2, FK_CAPACITY = 3 };
  int factor_kind = FK_NONE;
  bool factor_consumed = false;  // U = L^-T sits in the factor buffer's diagonal tiles (multi-GPU gradient; or a failed one)
  bool have_alpha = false;       // dalpha holds Sigma^-1 y of the current factorisation
  double* dDiagSave = nullptr;   // the factor's diagonal blocks while the gradient keeps U in their place
  int64_t cap_diag = 0;
  int64_t notpd = -1;
  double logdet = 0.0, vnorm2 = 0.0;

  // predict workspace
  int64_t Mt_cap = 0;
  double* dV = nullptr;
  // GEMM-form prediction (predict_form.hpp): behind a gradient on the tile path U = L^-T of the resident factor sits in the
  // factor buffer's upper triangle + dUdiag (`u_valid`); the first prediction transposes it into dW's lower triangle
  // (`linv_valid`; Sigma^-1 is dead by then) and every prediction until the next factorisation is one GEMM into dV2
  bool u_valid = false, linv_valid = false;
  int predict_form = -1;   // -1 = GEMM form whenever U is there; 0 = always the triangular solve; 1 = as -1 (explicit)
  double* dV2 = nullptr;   // Np x Mt: the solved cross-covariance with the TRAINING index fastest
  int64_t Mt2_cap = 0;
  double* dXs = nullptr;
  double* txs = nullptr;
  double* txl = nullptr;
  int32_t* tcat = nullptr;
  double* dkss = nullptr;
  double* dpart = nullptr;
  double* dmean = nullptr;
  double* dvar = nullptr;
  int64_t cap_part = 0;

  // gradient workspace
  double* dW = nullptr;  // Np x Np column-major: L^-1 then Sigma^-1
  int64_t cap_W = 0;
  double* dalpha = nullptr;
  int64_t cap_pts_alpha = 0;
  double* dgpart = nullptr;   // accumulators of the trace reductions (GACC_DOUBLES) + 64 doubles of scratch
  int64_t cap_gpart = 0;
  double* dgred = nullptr;    // per-workgroup partial vectors of grad_tile_kernel (two-stage, fixed-order sums)
  int64_t cap_gred = 0;
  double* dgbig = nullptr;    // per-wave copies of coregion tables with more than 8 levels
  int64_t cap_gbig = 0;

  // cached launch plan of the batched levels of the triangular inverse (static per data set)
  struct InvLevelPlan {
    int variant[2] = {0, 0};     // tile shape of the T^T and W21 batches
    int grid_x[2] = {0, 0};
    int count = 0;               // nodes on this level
    int64_t off[2] = {0, 0};     // first descriptor of each batch in dplan_gemm
    int64_t toff = 0;            // first transpose job in dplan_tr
    int tr_rows = 0, tr_cols = 0;  // largest transpose
    double flops[2] = {0.0, 0.0};
  };
  std::vector<InvLevelPlan> inv_plan;  // indexed by depth; count == 0 -> level not batched
  GemmArgs* dplan_gemm = nullptr;
  TransposeJob* dplan_tr = nullptr;
  int64_t plan_N = -1, plan_ld = -1, plan_Np = -1;
  const double* plan_W = nullptr;
  const double* plan_A = nullptr;
  bool batch_inverse = true;
  bool lpt_order = true;
  long long lpt_max_tiles = 16384;  // launches with more 128 x 128 tiles than this keep the L2-aware strip order (GMB_LPT_MAX_TILES)
  int cov_strip = 0;   // GMB_COV_STRIP: tiles per workgroup of the covariance build (0 = by size)
  int tile_strip = 8;  // GMB_TILE_STRIP: m-tiles per strip of the L2-aware tile order (0 = row-major runs)

  // timing
  bool profiling = false;
  gmb_timings tm{};
  std::vector<EventPair> evs;
  bool naive_leaf = false;
  bool gemm_dma = true;   // the 128 x 128 launches stage their operands by LDS-DMA (gemm_f64_dma_kernel); GMB_GEMM_DMA=0: the
                          // register-staged double buffer (tools/gpu_ab_gemm_dma.py)
  bool small_tiles = true;
  int gemm_variant = 0;
  long long wg_slots = 512;  // resident GEMM workgroups: two per compute unit
  bool force_variant = false;  // tuning: GMB_GEMM_VARIANT pins the tile shape of every out-of-place GEMM

  // concurrency inside one factorisation / gradient: `cur` is the stream the launch helpers use;
  // it is `stream` except inside the look-ahead Cholesky (panel chain on aux[0]) and the
  // level-parallel triangular inverse (independent merges dealt over stream + aux[0..2])
  hipStream_t cur = nullptr;
  hipStream_t aux[3] = {nullptr, nullptr, nullptr};
  // event kind of the updates chol_cols issues: 7 (bulk trailing update) for the plain recursion over the whole
  // matrix, 0 (in-panel product of the latency-bound chain) inside the look-ahead schedules' panels
  int chol_update_kind = 7;
  // -1 = by size; 0 = plain recursion; 2 = masked look-ahead; 3 = persistent tile kernel (chol_tiles.hpp)
  int chol_scheme = -1;
  // persistent tile Cholesky: control words + per-tile flags (zeroed before every launch), optional task trace
  uint32_t* dct = nullptr;
  int64_t cap_ct = 0;
  unsigned long long* dct_trace = nullptr;
  int64_t cap_ct_trace = 0;
  bool ct_trace = false;
  bool ct_used = false;      // the last factorisation ran on the tile kernel (its abort word has to be read back)
  bool ct_panels = false;    // ... as panel launches at the bottom of the recursion (chol_tiles_panel)
  bool ct_traced = false;    // ... and left task stamps in dct_trace
  bool tt_used = false;      // the current prediction ran its triangular solve on the tile kernel (same abort word)
  int ct_ntasks = 0;
  int ct_lose = 0;           // fault injection (gmb_debug_chol_lose_tickets): one shot
  bool fact_in_flight = false;  // factorize_enqueue has run, factorize_finish has not
  bool ct_injected = false;  // the last tile launch ran with injected faults (its failure is reported, not retried)
  // a factorisation that has been enqueued but not checked yet (factorize_enqueue / factorize_finish): where its scalars
  // land on the host, and the events around its two phases
  // (PINNED host memory: an asynchronous copy into pageable memory makes the host wait for the stream, and the gradient of
  // gmb_evaluate would be enqueued only after the factorisation has finished -- 80 us of idle GPU per evaluation at N = 2000)
  struct HostLanding {
    double scal[2];
    int32_t info;
    uint32_t abort;
    double gacc[GACC_DOUBLES];  // gmb_evaluate's light path: the gradient accumulators land here too (eval_land_kernel)
  };
  HostLanding* hl = nullptr;
  HostLanding* hl_dev = nullptr;  // the same memory as the device addresses it
  hipEvent_t fe[4] = {nullptr, nullptr, nullptr, nullptr};  // K-build begin / end, Cholesky begin / end
  // gmb_evaluate with a gradient on the fused launch (`light`): one memset for scalars + accumulators, no copies in between --
  // eval_land_kernel writes everything the host wants into hl at the end -- and, below 4096 rows and without gmb_set_profiling,
  // no phase events (six event records were ~30 us of a 380 us evaluation at N = 392)
  bool eval_call = false;     // set by gmb_evaluate around its factorize_enqueue
  bool light = false;         // the evaluation in flight runs that way
  bool gacc_zeroed = false;   // ... and its memset covered the accumulators (grad_reduce skips its own)
  unsigned long long* prep_zero[2] = {nullptr, nullptr};  // what the NEXT prep_points launch clears besides its own work
  int64_t prep_zero_words[2] = {0, 0};
  bool prezeroed = false;     // gmb_evaluate's prep_points launch has cleared the scalar block and the tile launch's words
  bool prep_deferred = false; // small gmb_evaluate: the points are prepared (and those words cleared) by the covariance build itself
                              // (CovTileArgs::inline_prep) -- raised by gmb_set_theta, consumed by build_sigma, flushed by gmb_evaluate
                              // if the build was never reached
  int inline_prep_max_rows = 4096;
  bool fe_recorded = false;   // fe[] belong to the factorisation in flight
  // persistent evaluation launch (eval_tiles.hpp): L^-T by rows and Sigma^-1 as tile tasks, fused with the tile Cholesky
  // when the caller asks for the gradient together with the factorisation (gmb_evaluate)
  int grad_scheme = -1;        // -1 = by size; 0 = launch tree (winv_levels); 1 = tile tasks behind the factorisation; 2 = fused
  int et_lag = -1;             // the INV tasks of column c - lag follow the CHOL tasks of column c in the task list (-1: by size)
  uint32_t* det_tasks = nullptr;  // task list on the device, and what it was built for
  int64_t cap_et_tasks = 0;
  int et_key[5] = {-1, -1, -1, -1, -1};
  int et_pairs = -1;           // Sigma^-1 tasks of two tiles (128 x 256; eval_tiles.hpp: et_zz2_task): -1 = by size, 0 = off, 1 = on
  int et_pairs_min_blocks = 72;  // (measured, profiles/r05_eval_pairs_ab.txt: +19 % at 24 block columns, +8 % at 41, 0 at 50 - 64,
                                 //  -0.7 % at 79, -1.0 ... -1.2 % at 128 - 157: a pair halves the number of tasks the tail is balanced with)
  int et_ntasks = 0;
  double* dUdiag = nullptr;    // diagonal tiles of U = L^-T
  int64_t cap_udiag = 0;
  double* dApart = nullptr;    // partial products U(r,c) v_c
  int64_t cap_apart = 0;
  bool et_fused = false;       // the factorisation in flight carries INV / ZZ tasks: Sigma^-1 (dW) and the alpha parts come with it
  int et_min_blocks = 1;       // smallest matrix (in 128-blocks) whose MAP evaluation runs as the fused launch by default
  // the bottom of the plain recursion (matrices beyond tiles_max_blocks): panels of <= panel_tiles_w block columns, with every
  // row below them, are factored by ONE launch of the tile kernel instead of leaf / strip / small-product launches
  // (chol_tiles_panel).  `panel_tiles_on` is raised by factorize_enqueue around the single-engine recursion
  // only: the multi-GPU driver's chains run beside a bulk stream whose workgroups a persistent launch would not fit beside.
  // Width by A/B on one box at N = 50k (profiles/r06_panel_tiles_ab.txt): launches only 598 ms; 4: 600, 8: 594.5, 16: 591, 32: 594 --
  // the panel launch does the in-panel work at the bulk updates' rate (73 TF/s-equivalent, chain included); what it takes away
  // is the ~500 launch boundaries of the leaves, strips and small products.
  int panel_tiles_w = 16;
  bool panel_tiles_on = false, panel_tiles_veto = false;
  int tiles_min_blocks = 6, tiles_max_blocks = 224;  // matrices (in 128-blocks) the tile kernel factors by default (measured faster from N = 768 on)
  int tiles_trsm_min_blocks = 16;                    // ... and the tile triangular solve of the predict path (measured from N = 2560 on)
  int masked_max_blocks = 128;  // GMB_MASKED_MAX_BLOCKS: largest matrix (in 128-blocks) factored with the masked bulk stream
  std::vector<hipEvent_t> sync_pool;
  size_t sync_next = 0;
  bool lookahead = true;
  bool par_inverse = true;
  bool aux_shared = false;  // aux[2] is the process-wide masked stream (not ours to destroy)
  bool aux_borrowed = false;  // aux[0..2] belong to the engine this one was created beside (gmb_create_sibling)
  int part_cus = 0;         // compute units the masked stream leaves free
  // multi-GPU driver (dist_driver.hpp): packed send / receive staging of the all-gathers
  double* dsend = nullptr;
  double* drecv = nullptr;
  int64_t cap_send = 0, cap_recv = 0;
  double* dsend2 = nullptr;  // ... and of the panel columns' TAILs, which travel on the communication stream beside the next chain
  double* drecv2 = nullptr;
  int64_t cap_send2 = 0, cap_recv2 = 0;
  int panel_blocks = 8;
  bool panel_auto = true;  // panel width grows with the matrix (GMB_PANEL_BLOCKS pins it)
  // every device buffer obtained through ensure / alloc, with its size (gmb_resident_bytes)
  std::unordered_map<void*, size_t> allocs;
  int64_t bytes_resident = 0, bytes_peak = 0;
  // Multi-GPU driver, CAPACITY mode (gmb_dist_set_mode(e, 1); dist_capacity.hpp): no rank holds the whole factor.  dAown =
  // this rank's block rows, packed ((owned blocks * 128) rows x Np columns); dPanel = one column panel of L with ALL its rows
  // (what a panel all-gather delivers), addressed through a virtual full-buffer base so that every kernel written for the
  // replicated layout works on it unchanged -- two of them (the factorisation's look-ahead; the gradient: one panel of L and
  // one column chunk of U = L^-T with all its rows).
  int dist_mode = 0;
  double* dAown = nullptr;
  int64_t cap_Aown = 0;
  int64_t ld_own = 0;
  int own_world = -1, own_rank = -1;  // what dAown holds the rows of
  double* dPanel = nullptr;
  int64_t cap_panel = 0;
  // gmb_predict's triangular solve V <- V L^-T, when somebody else has to do it (the capacity driver streams L through)
  std::function<int(double*, int64_t, int)> solve_hook;
  double* dstat = nullptr;  // status words (+ payload) the ranks exchange: this rank's, then everybody's (dist_agree)
  // Every collective this engine has issued on a transport, in host issue order: count and a running hash of (element count,
  // stream).  RCCL matches the collectives of a communicator by ISSUE ORDER on every rank -- and gmb_dist_nlml issues them on
  // two streams -- so the ranks compare these two words whenever they agree on a status: a rank that issued a different
  // sequence is reported instead of silently pairing the wrong buffers.
  uint64_t coll_count = 0, coll_hash = 0;
  std::vector<hipEvent_t> time_pool;  // timing events of the multi-GPU driver's communication probes
  size_t time_next = 0;
};

namespace {

int fail(gmb_engine* e, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) e->err = buf;
  return code;
}

#define HIP_TRY(e, call)                                                              \
  do {                                                                                \
    hipError_t _s = (call);                                                           \
    if (_s != hipSuccess)                                                             \
      return fail(e, GMB_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_s), \
                  __FILE__, __LINE__);                                                \
  } while (0)

// device allocations of an engine go through ensure / alloc / release, which keep the engine's resident byte count
// (gmb_resident_bytes: what the multi-GPU driver's capacity mode is judged by)
void release(gmb_engine* e, void* p) {
  if (!p) return;
  if (e) {
    auto it = e->allocs.find(p);
    if (it != e->allocs.end()) {
      e->bytes_resident -= (int64_t)it->second;
      e->allocs.erase(it);
    }
  }
  (void)hipFree(p);
}

template <typename T>
int ensure(gmb_engine* e, T** p, int64_t* cap, int64_t need) {
  if (*cap >= need && *p) return GMB_OK;
  release(e, (void*)*p);
  *p = nullptr;
  *cap = 0;
  if (need < 1) need = 1;
  hipError_t s = hipMalloc((void**)p, (size_t)need * sizeof(T));
  if (s != hipSuccess)
    return fail(e, GMB_ENOMEM, "hipMalloc of %lld bytes failed: %s", (long long)(need * sizeof(T)),
                hipGetErrorString(s));
  *cap = need;
  if (e) {
    e->allocs[(void*)*p] = (size_t)need * sizeof(T);
    e->bytes_resident += (int64_t)((size_t)need * sizeof(T));
    e->bytes_peak = std::max(e->bytes_peak, e->bytes_resident);
  }
  return GMB_OK;
}

template <typename T>
int alloc(gmb_engine* e, T** p, int64_t need) {
  int64_t cap = 0;
  release(e, (void*)*p);
  *p = nullptr;
  return ensure(e, p, &cap, need);
}

int pick_nc(int n) { return n <= 1 ? 1 : n <= 2 ? 2 : n <= 4 ? 4 : n <= 8 ? 8 : 16; }

// ---- timing helpers -----------------------------------------------------------------------
struct PhaseTimer {
  gmb_engine* e;
  hipEvent_t a = nullptr, b = nullptr;
  explicit PhaseTimer(gmb_engine* e_) : e(e_) {
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    (void)hipEventRecord(a, e->stream);
  }
  void stop() { (void)hipEventRecord(b, e->stream); }
  double ms() {  // caller has synchronised the stream
    float t = 0.f;
    (void)hipEventElapsedTime(&t, a, b);
    return (double)t;
  }
  ~PhaseTimer() {
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
  }
};

void ev_begin(gmb_engine* e, int kind, double flops, int mt = 0, int nt = 0, int k = 0, int flags = 0) {
  if (!e->profiling) return;
  EventPair p;
  p.mt = mt;
  p.nt = nt;
  p.k = k;
  p.flags = flags;
  (void)hipEventCreate(&p.a);
  (void)hipEventCreate(&p.b);
  p.kind = kind;
  p.flops = flops;
  p.masked = e->aux_shared && e->cur == e->aux[2];
  (void)hipEventRecord(p.a, e->cur);
  e->evs.push_back(p);
}
void ev_end(gmb_engine* e) {
  if (!e->profiling) return;
  (void)hipEventRecord(e->evs.back().b, e->cur);
}
void ev_collect(gmb_engine* e) {  // stream already synchronised
  static FILE* trace = nullptr;
  if (!trace && !e->evs.empty()) {
    const char* path = getenv("GMB_TRACE_FILE");  // per-launch log for tuning: kind mt nt k flags ms GF
    if (path && path[0]) trace = fopen(path, "a");
  }
  // Wall time during which at least one GEMM launch of this call was in flight (union of the launch
  // intervals on the device timeline): launches of the two streams of the look-ahead schedules
  // overlap, so the SUM of their durations exceeds the time they cover.
  if (!e->evs.empty()) {
    auto interval_union = [&](auto pick) {
      std::vector<std::pair<float, float>> iv;
      for (auto& p : e->evs)
        if (pick(p.kind)) {
          float ta = 0.f, tb = 0.f;
          if (hipEventElapsedTime(&ta, e->evs[0].a, p.a) == hipSuccess && hipEventElapsedTime(&tb, e->evs[0].a, p.b) == hipSuccess)
            iv.emplace_back(ta, tb);
        }
      std::sort(iv.begin(), iv.end());
      float cur_lo = 0.f, cur_hi = -1.f;
      double uni = 0.0;
      for (auto& x : iv) {
        if (cur_hi < cur_lo || x.first > cur_hi) {
          if (cur_hi >= cur_lo) uni += cur_hi - cur_lo;
          cur_lo = x.first;
          cur_hi = x.second;
        } else if (x.second > cur_hi) {
          cur_hi = x.second;
        }
      }
      if (cur_hi >= cur_lo) uni += cur_hi - cur_lo;
      return uni;
    };
    e->tm.total_gemm_wall_ms += interval_union([](int k) { return k == 0 || k == 7 || k == 10 || k == 2 || k == 3 || k == 4; });
    e->tm.total_chol_gemm_wall_ms += interval_union([](int k) { return k == 7; });
  }
  for (auto& p : e->evs) {
    float t = 0.f;
    (void)hipEventElapsedTime(&t, p.a, p.b);
    if (trace) fprintf(trace, "%d %d %d %d %d %.5f %.1f\n", p.kind, p.mt, p.nt, p.k, p.flags, t, p.flops / 1e9);
    if (p.kind == 0 || p.kind == 7 || p.kind == 8 || p.kind == 9 || p.kind == 10 || p.kind == 2 || p.kind == 3 || p.kind == 4) {
      e->tm.total_gemm_ms += t;
      e->tm.total_gemm_flops += p.flops;
      e->tm.total_gemm_launches += 1;
      if (p.masked) {
        e->tm.masked_gemm_ms += t;
        e->tm.masked_gemm_flops += p.flops;
      }
    }
    switch (p.kind) {
      case 8:  // the persistent tile kernel: contraction, leaves and strip solves of the whole factorisation in one launch
        e->tm.total_chol_tile_ms += t;
        e->tm.total_chol_tile_flops += p.flops;
        e->tm.total_chol_tile_launches += 1;
        e->tm.chol_gemm_ms += t;
        e->tm.chol_gemm_flops += p.flops;
        e->tm.chol_gemm_launches += 1;
        break;
      case 9:  // the persistent evaluation launch: (factorisation +) inverse + Sigma^-1 (flags & 1: with the factorisation)
        e->tm.total_eval_tile_ms += t;
        e->tm.total_eval_tile_flops += p.flops;
        e->tm.total_eval_tile_launches += 1;
        e->tm.grad_gemm_ms += t;
        e->tm.grad_gemm_flops += p.flops;
        break;
      case 10:  // a panel of the recursion's bottom on the tile kernel: in-panel work of the chain, like kind 0
        e->tm.chol_gemm_ms += t;
        e->tm.chol_gemm_flops += p.flops;
        e->tm.chol_gemm_launches += 1;
        e->tm.total_chol_panel_gemm_ms += t;
        e->tm.total_chol_panel_gemm_flops += p.flops;
        e->tm.total_chol_panel_tile_ms += t;
        e->tm.total_chol_panel_tile_flops += p.flops;
        e->tm.total_chol_panel_tile_launches += 1;
        break;
      case 0:
      case 7:
        e->tm.chol_gemm_ms += t;
        e->tm.chol_gemm_flops += p.flops;
        e->tm.chol_gemm_launches += 1;
        if (p.flags & (1 << 16)) {
          e->tm.total_chol_update_all_ms += t;
          e->tm.total_chol_update_all_flops += p.flops;
          e->tm.total_chol_update_all_launches += 1;
        }
        if (p.kind == 7) {
          e->tm.total_chol_gemm_ms += t;
          e->tm.total_chol_gemm_flops += p.flops;
          e->tm.total_chol_gemm_launches += 1;
        } else {
          e->tm.total_chol_panel_gemm_ms += t;
          e->tm.total_chol_panel_gemm_flops += p.flops;
        }
        break;
      case 1: e->tm.chol_leaf_ms += t; break;
      case 2:
      case 5: e->tm.chol_trsm_ms += t; break;
      case 3:
        e->tm.predict_gemm_ms += t;
        e->tm.predict_gemm_flops += p.flops;
        e->tm.predict_gemm_launches += 1;
        break;
      case 4:
        e->tm.grad_gemm_ms += t;
        e->tm.grad_gemm_flops += p.flops;
        break;
    }
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  e->evs.clear();
  if (trace) fflush(trace);
}

// ---- kernel launch helpers ----------------------------------------------------------------
// `g_in` describes the product in 128-tile units (mt, nt) and elements (k); pick the block tile so
// that the launch fills the chip, convert, schedule and launch.
long long gemm_nact128(const GemmArgs& g) {
  long long nact = 0;
  const int stride = g.nblk_stride < 1 ? 1 : g.nblk_stride;
  for (int tm = 0; tm < g.mt; ++tm) {
    int f = g.tri ? gemm_first_tn((int64_t)tm * TILE - g.tri_off - (TILE - 1), TILE, stride) : 0;
    f = f > g.nt ? g.nt : f;
    nact += g.nt - f;
  }
  return nact;
}

int launch_gemm(gmb_engine* e, const GemmArgs& g_in, int ev_kind) {
  if (g_in.mt <= 0 || g_in.nt <= 0 || g_in.k <= 0) return GMB_OK;
  GemmArgs g = g_in;
  // tiles the 128 x 128 tiling would compute
  if (g.nblk_stride < 1) g.nblk_stride = 1;
  long long nact = 0;
  for (int tm = 0; tm < g.mt; ++tm) {
    int f = g.tri ? gemm_first_tn((int64_t)tm * TILE - g.tri_off - (TILE - 1), TILE, g.nblk_stride) : 0;
    f = f > g.nt ? g.nt : f;
    nact += g.nt - f;
  }
  const bool in_place = (const double*)g.C == g.B;  // the block must own every column of its rows
  // variant: 0 = 128x128, 1 = 64x64, 2 = 128x64, 3 = 128x32 (all 4 waves, two workgroups per compute unit)
  int variant = e->gemm_variant;
  {
    // Tile shape by a small model: a launch of T tiles on S workgroup slots (two per compute
    // unit for every shape) takes max(1, T / S + 1/2) tile times -- full rounds plus an expected
    // half round of tail -- and a tile time is tile flops / intrinsic rate.  Intrinsic rates
    // measured on MI355X on large plain products (TF/s, r02 kernels with the interleaved staging):
    // 128x128 73, 64x64 63, 128x64 65, 128x32 60.  Measured launch by launch on the N = 10k fit
    // (round 1): always-128x128 61.7 ms, always-64x64 52.2 ms, best shape
    // per launch 51.0 ms.
    static const double rate[4] = {73.0, 63.0, 65.0, 60.0};
    static const int per128[4] = {1, 4, 2, 4};  // tiles per 128 x 128 block
    const double slots = (double)e->wg_slots;
    auto score = [&](int v) {  // higher is better: 1 / predicted time
      const double t = (double)(nact * per128[v]);
      const double rounds = std::max(1.0, t / slots + 0.5);
      return rate[v] * per128[v] / rounds;
    };
    if (in_place) {
      if (g.mt != 1) return fail(e, GMB_EINVAL, "internal: in-place GEMM needs m == 128");
      variant = 0;
      if (score(2) > score(variant)) variant = 2;
      if (score(3) > score(variant)) variant = 3;
    } else if (!e->force_variant) {
      variant = 0;
      if (e->small_tiles) {
        if (score(2) > score(variant)) variant = 2;
        if (score(1) > score(variant)) variant = 1;
      }
    }
  }
#ifdef GMB_TUNING
  // A/B only: variant 4 = a 256 x 128 macro-tile, eight waves (4 x 2), ONE workgroup per compute
  // unit -- the same two waves per SIMD, half the B-panel traffic per flop; m-tile counts must be even
  if (e->force_variant && e->gemm_variant == 4 && !in_place) variant = (g_in.mt % 2 == 0) ? 4 : 0;
#endif
  static const int BMs[5] = {128, 64, 128, 128, 256}, BNs[5] = {128, 64, 64, 32, 128};
  const int bm = BMs[variant], bn = BNs[variant];
  g.mt = g_in.mt * TILE / bm;
  g.nt = g_in.nt * TILE / bn;
  // triangular operand: dispatch the longest contractions first (see GemmArgs::order)
  // ... unless the launch is so large that a late long tile is a per cent of it at worst, and the L2-aware strips below -- which
  // the longest-first order gives up -- are worth more (round 6, one box, N = 50k: evaluation 1788 -> 1771 ms with the strips for
  // every product of more than lpt_max_tiles tiles; tools/gpu_grad_order_ab.py)
  g.order = 0;
  if (e->lpt_order && nact <= e->lpt_max_tiles && g.nblk_stride == 1 && g.tri_off >= 0 && !g.klo_m && (g.klo_n != 0) != (g.khi_n != 0))
    g.order = g.klo_n ? 1 : 2;
  // L2-aware rasterisation of the XCD runs for launches that are several patches large (GemmArgs::strip)
  g.strip = (g.order == 0 && e->tile_strip > 0 && g.mt >= 2 && !in_place && !g.klo_m) ? e->tile_strip : 0;
  double flops = 0.0;
  const int nblocks = gemm_schedule(g, bm, bn, &flops);
  if (nblocks <= 0) return GMB_OK;
  // Population 7 = the Cholesky's BULK trailing updates = the kind-7 launches that run the 128 x 128 kernel; they go through an
  // entry point of their own (gemm_f64_dma_chol_update_kernel, the same body), so that a rocprofv3 kernel trace lists exactly
  // the launches the live events of `roofline` time.  Kind-7 launches small enough for another tile shape count as in-panel
  // products (kind 0), like the panels' own.
  const bool bulk_update = ev_kind == 7 && variant == 0 && !in_place;
  const int was_update = ev_kind == 7 ? (1 << 16) : 0;  // issued AS a trailing update, whatever tile shape it runs (total_chol_update_all_*)
  if (ev_kind == 7 && !bulk_update) ev_kind = 0;
  ev_begin(e, ev_kind, flops, g_in.mt, g_in.nt, g.k,
           g.tri | (g.klo_n << 3) | (g.khi_n << 4) | (g.klo_m << 5) | (variant << 8) | was_update);
  const dim3 grid(nblocks);
  const bool pfc = g.beta != 0.0 && g.k <= 1024 && !in_place;
  switch (variant) {
    case 0:
      if (e->gemm_dma && bulk_update) hipLaunchKernelGGL(gemm_f64_dma_chol_update_kernel, grid, dim3(256), 0, e->cur, g);
      else if (e->gemm_dma && !in_place) hipLaunchKernelGGL(gemm_f64_dma_kernel, grid, dim3(256), 0, e->cur, g);
      else hipLaunchKernelGGL((gemm_f64_kernel<2, 2, 4, 4, 2>), grid, dim3(256), 0, e->cur, g);
      break;
#ifdef GMB_TUNING
    case 4: hipLaunchKernelGGL((gemm_f64_kernel<4, 2, 4, 4, 1>), grid, dim3(512), 0, e->cur, g); break;
#endif
    // small tiles with a short contraction and beta != 0: the C-prefetching instantiations (gemm_f64.hpp)
    case 1:
      if (pfc) hipLaunchKernelGGL((gemm_f64_kernel<2, 2, 2, 2, 2, true>), grid, dim3(256), 0, e->cur, g);
      else hipLaunchKernelGGL((gemm_f64_kernel<2, 2, 2, 2, 2>), grid, dim3(256), 0, e->cur, g);
      break;
    case 2:
      if (pfc) hipLaunchKernelGGL((gemm_f64_kernel<2, 2, 4, 2, 2, true>), grid, dim3(256), 0, e->cur, g);
      else hipLaunchKernelGGL((gemm_f64_kernel<2, 2, 4, 2, 2>), grid, dim3(256), 0, e->cur, g);
      break;
    default:
      if (pfc) hipLaunchKernelGGL((gemm_f64_kernel<2, 2, 4, 1, 2, true>), grid, dim3(256), 0, e->cur, g);
      else hipLaunchKernelGGL((gemm_f64_kernel<2, 2, 4, 1, 2>), grid, dim3(256), 0, e->cur, g);
      break;
  }
  ev_end(e);
  HIP_TRY(e, hipGetLastError());
  return GMB_OK;
}

// B <- B inv(L_kk)^T on `nrows` rows (multiple of 16): one wavefront per 16 rows
int launch_trsm_strip(gmb_engine* e, double* B, int64_t ldb, int64_t nrows, const double* L, int64_t ldl,
                      const double* dinv16, int nvalid, int ev_kind) {
  if (nrows <= 0) return GMB_OK;
  TrsmArgs t;
  t.B = B;
  t.ldb = ldb;
  t.nrows = nrows;
  t.L = L;
  t.ldl = ldl;
  t.dinv16 = dinv16;
  t.nvalid = nvalid;
  ev_begin(e, ev_kind, (double)nrows * TILE * TILE);
  hipLaunchKernelGGL(trsm_strip_kernel, dim3((unsigned)((nrows / 16 + 3) / 4)), dim3(256), 0, e->cur, t);
  ev_end(e);
  HIP_TRY(e, hipGetLastError());
  return GMB_OK;
}

int launch_leaf(gmb_engine* e, const LeafArgs& a) {
  ev_begin(e, 1, 0.0);
  if (e->naive_leaf)
    hipLaunchKernelGGL(potrf_leaf_naive_kernel, dim3(1), dim3(256), 0, e->cur, a);
  else
    hipLaunchKernelGGL(potrf_leaf_kernel, dim3(1), dim3(512), 0, e->cur, a);
  ev_end(e);
  HIP_TRY(e, hipGetLastError());
  return GMB_OK;
}

// grid = the strips of cov_tile_kernel's enumeration (strip tiles per workgroup along the column index)
long long cov_grid_blocks(int ti, int tj, int strip, int tri_grid, int row_first, int row_stride) {
  const int S = strip < 1 ? 1 : strip;
  const long long nst = (tj + S - 1) / S;
  if (row_stride > 0) return row_first < ti ? ((ti - 1 - row_first) / row_stride + 1) * nst : 0;
  if (tri_grid) return cov_tri_blocks_before(ti, tj, S);
  return (long long)ti * nst;
}

template <int KIND>
int launch_cov_nc(gmb_engine* e, const CovTileArgs& a, int nc) {
  const long long nb = cov_grid_blocks(a.ti, a.tj, a.strip, a.tri_grid, a.row_first, a.row_stride) * (a.gsplit > 1 ? a.gsplit : 1);
  if (nb <= 0) return GMB_OK;
  const dim3 grid((unsigned)nb), block(256);
  switch (nc) {
    case 1: hipLaunchKernelGGL((cov_tile_kernel<KIND, 1>), grid, block, 0, e->stream, a); break;
    case 2: hipLaunchKernelGGL((cov_tile_kernel<KIND, 2>), grid, block, 0, e->stream, a); break;
    case 4: hipLaunchKernelGGL((cov_tile_kernel<KIND, 4>), grid, block, 0, e->stream, a); break;
    case 8: hipLaunchKernelGGL((cov_tile_kernel<KIND, 8>), grid, block, 0, e->stream, a); break;
    default: hipLaunchKernelGGL((cov_tile_kernel<KIND, 16>), grid, block, 0, e->stream, a); break;
  }
  HIP_TRY(e, hipGetLastError());
  return GMB_OK;
}

int launch_cov(gmb_engine* e, const CovTileArgs& a_in) {
  if (a_in.ti <= 0 || a_in.tj <= 0) return GMB_OK;
  CovTileArgs a = a_in;
  const double tiles = (double)a.ti * a.tj * (a.lower_only ? 0.5 : 1.0) / (a.row_stride > 0 ? a.row_stride : 1);  // this launch's share
  a.stream_stores = tiles * TILE * TILE * 8.0 >= 1073741824.0 / (a.row_stride > 0 ? a.row_stride : 1);  // by the whole matrix
  // tiles per workgroup: enough strips left to fill the chip's ~800 workgroup slots several times over
  a.strip = e->cov_strip > 0 ? e->cov_strip : (tiles >= 32768.0 ? 4 : tiles >= 12288.0 ? 2 : 1);
  // small grids (up to 1024 tiles, N <= 5.7k): four workgroups per tile (N = 392: the build's one launch 29.6 -> 11.8 us, N = 2560:
  // 43 -> 30, N = 4000: 55 -> 39, N = 5200: 61 -> 50; from ~2000 tiles on the split costs more than the direct-loop tiles' tail)
  a.gsplit = (a.strip == 1 && cov_grid_blocks(a.ti, a.tj, 1, a.tri_grid, a.row_first, a.row_stride) <= 1024) ? 4 : 1;
  switch (a.p.kind) {
    case GMB_EXPQUAD: return launch_cov_nc<0>(e, a, e->nc_pad);
    case GMB_MATERN52: return launch_cov_nc<1>(e, a, e->nc_pad);
    case GMB_MATERN32: return launch_cov_nc<2>(e, a, e->nc_pad);
    case GMB_MATERN12: return launch_cov_nc<3>(e, a, e->nc_pad);
    case GMB_EXPONENTIAL: return launch_cov_nc<4>(e, a, e->nc_pad);
  }
  return fail(e, GMB_EINVAL, "unknown kernel kind %d", a.p.kind);
}

int spec_ntab(const gmb_kernel_spec& s) { return s.n_coreg + (s.out_col >= 0 ? 1 : 0); }

int validate_spec(gmb_engine* e, const gmb_kernel_spec& s, int D) {
  if (s.kind < 0 || s.kind > 4) return fail(e, GMB_EINVAL, "kernel kind %d out of range", s.kind);
  if (s.n_cont < 1 || s.n_cont > GMB_MAX_DIMS)
    return fail(e, GMB_EINVAL, "n_cont=%d must be in [1,%d]", s.n_cont, GMB_MAX_DIMS);
  if (s.n_lin < 0 || s.n_lin > GMB_MAX_LIN)
    return fail(e, GMB_EINVAL, "n_lin=%d must be in [0,%d]", s.n_lin, GMB_MAX_LIN);
  if (s.n_coreg < 0 || s.n_coreg > GMB_MAX_COREG)
    return fail(e, GMB_EINVAL, "bad n_coreg=%d", s.n_coreg);
  for (int k = 0; k < s.n_cont; ++k)
    if (s.idx_cont[k] < 0 || (D > 0 && s.idx_cont[k] >= D))
      return fail(e, GMB_EINVAL, "idx_cont[%d]=%d outside the %d columns of X", k, s.idx_cont[k], D);
  for (int k = 0; k < s.n_lin; ++k)
    if (s.idx_lin[k] < 0 || (D > 0 && s.idx_lin[k] >= D))
      return fail(e, GMB_EINVAL, "idx_lin[%d]=%d outside the %d columns of X", k, s.idx_lin[k], D);
  for (int t = 0; t < s.n_coreg; ++t) {
    if (s.coreg_col[t] < 0 || (D > 0 && s.coreg_col[t] >= D))
      return fail(e, GMB_EINVAL, "coreg_col[%d]=%d outside X", t, s.coreg_col[t]);
    if (s.coreg_levels[t] < 1 || s.coreg_levels[t] > GMB_MAX_LEVELS)
      return fail(e, GMB_EINVAL, "coreg_levels[%d]=%d must be in [1,%d]", t, s.coreg_levels[t],
                  GMB_MAX_LEVELS);
  }
  if (s.out_col >= 0) {
    if (D > 0 && s.out_col >= D) return fail(e, GMB_EINVAL, "out_col=%d outside X", s.out_col);
    if (s.n_out < 1 || s.n_out > GMB_MAX_LEVELS)
      return fail(e, GMB_EINVAL, "n_out=%d must be in [1,%d]", s.n_out, GMB_MAX_LEVELS);
  }
  if (!(s.jitter >= 0.0)) return fail(e, GMB_EINVAL, "jitter must be >= 0");
  return GMB_OK;
}

// Fill PrepArgs / CovParams from spec + theta and upload the coregion tables.
int apply_theta(gmb_engine* e) {
  const gmb_kernel_spec& s = e->spec;
  const double* th = e->theta.data();
  int k = 0;
  const int n_ls = s.ard ? s.n_cont : 1;
  PrepArgs& pa = e->prep_proto;
  pa = PrepArgs{};
  pa.nc = s.n_cont;
  pa.nc_pad = e->nc_pad;
  for (int i = 0; i < s.n_cont; ++i) {
    const double ls = th[s.ard ? i : 0];
    if (!(ls > 0.0)) return fail(e, GMB_EINVAL, "lengthscale %d = %g must be positive", i, ls);
    pa.idx_cont[i] = s.idx_cont[i];
    pa.inv_ls[i] = 1.0 / ls;
  }
  k += n_ls;
  const double eta = th[k], sigma = th[k + 1];
  k += 2;
  CovParams& cp = e->cp;
  cp = CovParams{};
  cp.kind = s.kind;
  cp.eta2 = eta * eta;
  cp.sigma2 = sigma * sigma;
  cp.jitter = s.jitter;
  cp.n_lin = s.n_lin;
  pa.n_lin = s.n_lin;
  if (s.n_lin > 0) {
    for (int i = 0; i < s.n_lin; ++i) {
      pa.idx_lin[i] = s.idx_lin[i];
      pa.c_lin[i] = th[k + i];
    }
    cp.tau = th[k + s.n_lin];
    k += s.n_lin + 1;
  }
  // coregion tables: B = W W^T + diag(kappa)
  const int ntab = spec_ntab(s);
  cp.n_tab = ntab;
  pa.n_tab = ntab;
  e->htabs.clear();
  int off = 0;
  for (int t = 0; t < ntab; ++t) {
    const bool is_out = (t == s.n_coreg);
    const int L = is_out ? s.n_out : s.coreg_levels[t];
    const double* W = th + k;
    const double* kap = th + k + 2 * L;
    cp.tab_levels[t] = L;
    cp.tab_off[t] = off;
    pa.tab_col[t] = is_out ? s.out_col : s.coreg_col[t];
    pa.tab_levels[t] = L;
    for (int a = 0; a < L; ++a)
      for (int b = 0; b < L; ++b)
        e->htabs.push_back(W[2 * a] * W[2 * b] + W[2 * a + 1] * W[2 * b + 1] + (a == b ? kap[a] : 0.0));
    off += L * L;
    k += 3 * L;
  }
  cp.noise_tab = -1;
  e->hnoise.clear();
  if (s.out_col >= 0 && s.hetero_noise) {
    const int P = s.n_out;
    const double* W = th + k;
    const double* kap = th + k + 2 * P;
    for (int a = 0; a < P; ++a) e->hnoise.push_back(W[2 * a] * W[2 * a] + W[2 * a + 1] * W[2 * a + 1] + kap[a]);
    cp.noise_tab = s.n_coreg;  // the output table's category index
    k += 3 * P;
  }
  // additive model: per coregion dim one more [ls | eta | (c, tau)] block (validated here, used below)
  const int k_add = k;
  if (s.additive) k += s.n_coreg * (n_ls + 1 + (s.n_lin > 0 ? s.n_lin + 1 : 0));
  if (k != (int)e->theta.size()) return fail(e, GMB_EINVAL, "internal: theta packing mismatch");
  if (!e->dtabs) {
    int rc = alloc(e, &e->dtabs, (int64_t)MAX_TABS * GMB_MAX_LEVELS * GMB_MAX_LEVELS);
    if (rc) return rc;
    rc = alloc(e, &e->dnoise, (int64_t)GMB_MAX_LEVELS);
    if (rc) return rc;
  }
  if (!e->htabs.empty())
    HIP_TRY(e, hipMemcpyAsync(e->dtabs, e->htabs.data(), e->htabs.size() * sizeof(double),
                              hipMemcpyHostToDevice, e->stream));
  if (!e->hnoise.empty())
    HIP_TRY(e, hipMemcpyAsync(e->dnoise, e->hnoise.data(), e->hnoise.size() * sizeof(double),
                              hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));  // host vectors may be rewritten by the caller
  cp.tabs = e->dtabs;
  cp.noise_mult = e->dnoise;
  // ---- terms ----
  e->terms.clear();
  if (!s.additive) {
    e->terms.push_back(gmb_engine::Term{pa, cp, eta, ntab});
    return GMB_OK;
  }
  const PrepArgs pa_all = pa;
  const CovParams cp_all = cp;
  auto select_tables = [&](gmb_engine::Term& t, int dim) {  // dim < 0: output table only
    int n = 0;
    auto take = [&](int src) {
      t.cp.tab_levels[n] = cp_all.tab_levels[src];
      t.cp.tab_off[n] = cp_all.tab_off[src];
      t.pa.tab_col[n] = pa_all.tab_col[src];
      t.pa.tab_levels[n] = pa_all.tab_levels[src];
      ++n;
    };
    if (dim >= 0) take(dim);
    if (s.out_col >= 0) take(s.n_coreg);
    t.cp.n_tab = t.pa.n_tab = t.n_tab_acc = n;
    t.cp.noise_tab = (dim < 0 && s.out_col >= 0 && s.hetero_noise) ? n - 1 : -1;
  };
  gmb_engine::Term t0{pa_all, cp_all, eta, 0};
  select_tables(t0, -1);
  e->terms.push_back(t0);
  int ka = k_add;
  for (int j = 0; j < s.n_coreg; ++j) {
    gmb_engine::Term t{pa_all, cp_all, 0.0, 0};
    for (int i = 0; i < s.n_cont; ++i) {
      const double ls = th[ka + (s.ard ? i : 0)];
      if (!(ls > 0.0)) return fail(e, GMB_EINVAL, "lengthscale %d of additive term %d = %g must be positive", i, j, ls);
      t.pa.inv_ls[i] = 1.0 / ls;
    }
    ka += n_ls;
    t.eta = th[ka++];
    t.cp.eta2 = t.eta * t.eta;
    if (s.n_lin > 0) {
      for (int i = 0; i < s.n_lin; ++i) t.pa.c_lin[i] = th[ka + i];
      t.cp.tau = th[ka + s.n_lin];
      ka += s.n_lin + 1;
    }
    t.cp.sigma2 = 0.0;  // the noise is written by the first pass only
    t.cp.jitter = 0.0;
    select_tables(t, j);
    e->terms.push_back(t);
  }
  pa = e->terms[0].pa;  // the engine-wide defaults are the global term
  cp = e->terms[0].cp;
  return GMB_OK;
}

PrepArgs prep_args(gmb_engine* e, const double* dXraw, int64_t n, int64_t ldx, int64_t npad, double* xs, double* xl, int32_t* cat,
                   const PrepArgs* proto) {
  PrepArgs a = proto ? *proto : e->prep_proto;
  for (int z = 0; z < 2; ++z) {  // (one shot: gmb_set_theta inside gmb_evaluate)
    a.zero[z] = e->prep_zero[z];
    a.zero_words[z] = e->prep_zero_words[z];
    e->prep_zero[z] = nullptr;
  }
  a.X = dXraw;
  a.n = n;
  a.ldx = ldx;
  a.npad = npad;
  a.xs = xs;
  a.xl = xl;
  a.cat = cat;
  return a;
}

int prep_points(gmb_engine* e, const double* dXraw, int64_t n, int64_t ldx, int64_t npad, double* xs,
                double* xl, int32_t* cat, const PrepArgs* proto = nullptr) {
  const PrepArgs a = prep_args(e, dXraw, n, ldx, npad, xs, xl, cat, proto);
  hipLaunchKernelGGL(prep_points_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, e->stream, a);
  HIP_TRY(e, hipGetLastError());
  return GMB_OK;
}

// the training points' preparation that a small gmb_evaluate left to its covariance build, if that build was never reached
int flush_deferred_prep(gmb_engine* e) {
  if (!e->prep_deferred) return GMB_OK;
  e->prep_deferred = false;
  return prep_points(e, e->dX, e->N, e->D, e->Nr, e->xs, e->xl, e->cat);
}

PointSet train_set(const gmb_engine* e) { return PointSet{e->xs, e->xl, e->cat, e->N, e->Nr}; }

// the full factor buffer (Nr x Np, leading dimension Nr), allocated on first use
int ensure_factor_buffer(gmb_engine* e) {
  if (e->dist_mode == 1 && e->dA && e->cap_A == 0) e->dA = nullptr;  // (a virtual base left behind by the capacity driver)
  return ensure(e, &e->dA, &e->cap_A, e->ld * e->Np);
}

// Covariance build: lower-triangular tiles of Sigma = K + noise + jitter, the y row, identity padding, written
// column-major into `out` (Nr x Np, leading dimension ldo) -- the factor buffer in gmb_factorize.
int build_sigma(gmb_engine* e, double* out, int64_t ldo) {
  int rc;
  CovTileArgs a{};
  a.p = e->cp;
  a.rows = train_set(e);
  a.cols = train_set(e);
  a.out = out;
  a.ldo = ldo;
  a.i0 = a.j0 = 0;
  a.ti = (int)(e->Nr / TILE);
  a.tj = (int)(e->Np / TILE);
  a.mode = COV_TRAIN;
  a.lower_only = 1;
  a.tri_grid = 1;
  a.y = e->dy;
  if (e->prep_deferred) {  // small gmb_evaluate: this launch prepares the points it stages (and clears prep_points_kernel's words)
    e->prep_deferred = false;
    a.inline_prep = 1;
    a.prep = prep_args(e, e->dX, e->N, e->D, e->Nr, e->xs, e->xl, e->cat, nullptr);
  }
  if ((rc = launch_cov(e, a))) return rc;
  // additive models: the other terms add their covariance to the real entries, one pass each
  for (size_t t = 1; t < e->terms.size(); ++t) {
    if ((rc = prep_points(e, e->dX, e->N, e->D, e->Nr, e->xs, e->xl, e->cat, &e->terms[t].pa))) return rc;
    a.p = e->terms[t].cp;
    a.accumulate = 1;
    if ((rc = launch_cov(e, a))) return rc;
  }
  if (e->terms.size() > 1 &&
      (rc = prep_points(e, e->dX, e->N, e->D, e->Nr, e->xs, e->xl, e->cat, &e->terms[0].pa)))
    return rc;
  return GMB_OK;
}

// ---- Cholesky recursion -------------------------------------------------------------------
// `rend`: block row where the panel solve / updates stop (the whole matrix, or the end of a
// diagonal square when the rows below are solved later in bulk)
int chol_leaf(gmb_engine* e, int c, int rend) {
  LeafArgs a{};
  a.A = e->dA + (int64_t)c * TILE + (int64_t)c * TILE * e->ld;
  a.lda = e->ld;
  a.nvalid = (int)std::min<int64_t>(TILE, e->N - (int64_t)c * TILE);
  a.dinv16 = e->dDinv16 + (int64_t)c * 8 * 256;
  a.logdet = e->dscal;
  a.info = e->dinfo;
  a.row0 = (int64_t)c * TILE;
  a.dbg = nullptr;
  int rc = launch_leaf(e, a);
  if (rc) return rc;
  // panel rows below the diagonal block:  P <- P inv(L_cc)^T   (in place, 16 rows per wavefront)
  return launch_trsm_strip(e, e->dA + (int64_t)(c + 1) * TILE + (int64_t)c * TILE * e->ld, e->ld,
                           (int64_t)(rend - c - 1) * TILE, a.A, e->ld, a.dinv16, a.nvalid, 5);
}

int chol_tiles_panel(gmb_engine* e, int c0, int c1, int rend);

int chol_cols(gmb_engine* e, int c0, int c1, int rend) {
  if (c1 - c0 == 1) return chol_leaf(e, c0, rend);
  if (e->panel_tiles_on && c1 - c0 <= e->panel_tiles_w) return chol_tiles_panel(e, c0, c1, rend);
  const int mid = c0 + (c1 - c0 + 1) / 2;
  int rc = chol_cols(e, c0, mid, rend);
  if (rc) return rc;
  GemmArgs g{};
  g.C = e->dA + (int64_t)mid * TILE + (int64_t)mid * TILE * e->ld;
  g.ldc = e->ld;
  g.A = e->dA + (int64_t)mid * TILE + (int64_t)c0 * TILE * e->ld;
  g.lda = e->ld;
  g.B = g.A;
  g.ldb = e->ld;
  g.mt = c1 - mid;
  g.nt = rend - mid;
  g.k = (mid - c0) * TILE;
  g.alpha = -1.0;
  g.beta = 1.0;
  g.tri = 1;
  rc = launch_gemm(e, g, e->chol_update_kind);
  if (rc) return rc;
  return chol_cols(e, mid, c1, rend);
}

// The ragged last block of the persistent launches (chol_tiles.hpp: ct_ksum<.., RAG>): the fraction of a tile's contraction that
// the last block ROW's tasks (rows: the y row included) and the last block COLUMN's solve / inverse tasks still perform
double ragged_row_frac(const gmb_engine* e) {
  const int64_t rv = e->N + 1 - (e->Nr - TILE);
  return rv >= TILE ? 1.0 : (double)((rv + 15) / 16) / 8.0;
}
double ragged_col_frac(const gmb_engine* e) {
  const int64_t cv = e->N - (e->Np - TILE);
  return cv >= TILE ? 1.0 : (double)((cv + 15) / 16) / 8.0;
}
// flops of the tile Cholesky's contractions (2 * 128^3 per k-block of a tile)
double chol_tiles_flops(const gmb_engine* e, int nct, int nrt) {
  const double fr = ragged_row_frac(e);
  double flops = 0.0;
  for (int j = 1; j < nct; ++j) {
    const double tiles = nrt - 1 > j ? (double)(nrt - j - 1) + fr : (double)(nrt - j);  // (the diagonal tile is never cut)
    flops += 2.0 * TILE * TILE * TILE * (double)j * tiles;
  }
  return flops;
}

// ---- persistent tile Cholesky (chol_tiles.hpp): the whole factorisation in one launch --------------------
int chol_tiles(gmb_engine* e) {
  const int nct = (int)(e->Np / TILE), nrt = (int)(e->Nr / TILE);
  const int ntasks = ct_task_count(nct, nrt);
  const int64_t words = 4 + (int64_t)nrt * nct + 3 * (int64_t)nct;
  int rc;
  if ((rc = ensure(e, &e->dct, &e->cap_ct, words))) return rc;
  HIP_TRY(e, hipMemsetAsync(e->dct, 0, (size_t)words * sizeof(uint32_t), e->cur));
  e->ct_injected = e->ct_lose > 0;
  if (e->ct_lose > 0) {  // fault injection: tickets 0 .. ct_lose-1 are never handed out
    const uint32_t first = (uint32_t)e->ct_lose;
    e->ct_lose = 0;
    HIP_TRY(e, hipMemcpyAsync(e->dct, &first, sizeof first, hipMemcpyHostToDevice, e->cur));
    HIP_TRY(e, hipStreamSynchronize(e->cur));  // (`first` lives on this stack frame)
  }
  CholTilesArgs a{};
  a.A = e->dA;
  a.ld = e->ld;
  a.nct = nct;
  a.nrt = nrt;
  a.N = e->N;
  a.dinv16 = e->dDinv16;
  a.logdet = e->dscal;
  a.info = e->dinfo;
  a.ctl = e->dct;
  a.flags = e->dct + 4;
  a.half = e->dct + 4 + (int64_t)nrt * nct;
  a.prog = a.half + 2 * (int64_t)nct;
  a.ntasks = ntasks;
  a.timeout_us = 4000000u;  // a wait of 4 s means a lost flag: give the factorisation up, never the GPU
  a.dbg = nullptr;
  e->ct_traced = false;
  if (e->ct_trace) {
    e->ct_traced = true;
    if ((rc = ensure(e, &e->dct_trace, &e->cap_ct_trace, 4 * (int64_t)ntasks))) return rc;
    HIP_TRY(e, hipMemsetAsync(e->dct_trace, 0, (size_t)ntasks * 4 * sizeof(unsigned long long), e->cur));
    a.dbg = e->dct_trace;
  }
  const double flops = chol_tiles_flops(e, nct, nrt);
  // eight-wave workgroups, one per compute unit: every phase of the latency chain has the whole compute unit (N = 10k:
  // 6.8 ms against 8.7 ms with four-wave workgroups, two per compute unit -- which stay available to the tuning build)
  int nw = 8;
#ifdef GMB_TUNING
  if (const char* cw = getenv("GMB_CT_WAVES")) nw = atoi(cw) == 4 ? 4 : 8;
#endif
  int grid = (int)std::min<long long>(ntasks, nw == 8 ? e->wg_slots / 2 : e->wg_slots);
#ifdef GMB_TUNING
  if (const char* cg = getenv("GMB_CT_GRID")) grid = std::max(1, std::min(grid, atoi(cg)));
#endif
  ev_begin(e, 8, flops, nct, nrt, (int)e->Np, 0);
  if (nw == 8) hipLaunchKernelGGL(chol_tiles_kernel<8>, dim3(grid), dim3(512), 0, e->cur, a);
  else hipLaunchKernelGGL(chol_tiles_kernel<4>, dim3(grid), dim3(256), 0, e->cur, a);
  ev_end(e);
  HIP_TRY(e, hipGetLastError());
  e->ct_used = true;
  e->ct_ntasks = ntasks;
  return GMB_OK;
}

// Block columns [c0, c1) of the factor with every row below them down to block row `rend`, updated by everything left of
// c0 already: ONE launch of the tile kernel on that panel (left-looking tile tasks inside it; the rows below the square ride
// along as in the whole-matrix form).  Replaces, per panel of 16 block columns at C3, 16 leaf + 16 strip + 15 product launches
// (the 64 x 64-tile products that run at 45 TF/s: profiles/r05_bench_c3_kernel_stats.csv).  The control block is the tile
// Cholesky's; its ABORT word is not cleared between the panels of one factorisation (factorize_enqueue clears it once), so a
// panel that gave up waiting makes every later one drain and the host sees it at the end.
int chol_tiles_panel(gmb_engine* e, int c0, int c1, int rend) {
  const int nct = c1 - c0, nrt = rend - c0;
  const int ntasks = ct_task_count(nct, nrt);
  const int64_t words = 4 + (int64_t)nrt * nct + 3 * (int64_t)nct;
  int rc;
  if ((rc = ensure(e, &e->dct, &e->cap_ct, std::max<int64_t>(words, 4 + (int64_t)(e->panel_tiles_w + 3) * (e->Nr / TILE))))) return rc;  // (one size for every panel of the matrix)
  HIP_TRY(e, hipMemsetAsync(e->dct, 0, sizeof(uint32_t), e->cur));                                       // the ticket counter
  HIP_TRY(e, hipMemsetAsync(e->dct + 2, 0, (size_t)(words - 2) * sizeof(uint32_t), e->cur));            // flags and chain words
  CholTilesArgs a{};
  a.A = e->dA + (int64_t)c0 * TILE * (e->ld + 1);
  a.ld = e->ld;
  a.nct = nct;
  a.nrt = nrt;
  a.N = e->N - (int64_t)c0 * TILE;
  a.row_base = (int64_t)c0 * TILE;
  a.dinv16 = e->dDinv16 + (int64_t)c0 * 8 * 256;
  a.logdet = e->dscal;
  a.info = e->dinfo;
  a.ctl = e->dct;
  a.flags = e->dct + 4;
  a.half = e->dct + 4 + (int64_t)nrt * nct;
  a.prog = a.half + 2 * (int64_t)nct;
  a.ntasks = ntasks;
  a.timeout_us = 4000000u;
  a.dbg = nullptr;
  // flops of the panel's contractions (the last block row of the bordered matrix contracts its real rows only)
  const double fr = rend == (int)(e->Nr / TILE) ? ragged_row_frac(e) : 1.0;
  double flops = 0.0;
  for (int j = 1; j < nct; ++j) flops += 2.0 * TILE * TILE * TILE * (double)j * (nrt - 1 > j ? (double)(nrt - j - 1) + fr : (double)(nrt - j));
  const int grid = (int)std::min<long long>(ntasks, e->wg_slots / 2);
  ev_begin(e, 10, flops, nct, nrt, nct * TILE, 0);
  hipLaunchKernelGGL(chol_tiles_kernel<8>, dim3(grid), dim3(512), 0, e->cur, a);
  ev_end(e);
  HIP_TRY(e, hipGetLastError());
  e->ct_used = true;  // (the abort word is read back with the factorisation's scalars)
  e->ct_panels = true;
  return GMB_OK;
}

// V <- V L^-T (V: ntm x Np/128 tiles, leading dimension ldv) as one persistent launch (chol_tiles.hpp: trsm_tiles_kernel)
int trsm_tiles(gmb_engine* e, double* V, int64_t ldv, int ntm, int ev_kind) {
  const int nct = (int)(e->Np / TILE);
  const long long ntasks = (long long)ntm * nct;
  if (ntasks <= 0) return GMB_OK;
  const int64_t words = 4 + (int64_t)ntm * nct;
  int rc;
  if ((rc = ensure(e, &e->dct, &e->cap_ct, words))) return rc;
  HIP_TRY(e, hipMemsetAsync(e->dct, 0, (size_t)words * sizeof(uint32_t), e->cur));
  CholTilesArgs a{};
  a.A = e->dA;
  a.ld = e->ld;
  a.nct = nct;
  a.nrt = nct;
  a.N = e->N;
  a.dinv16 = e->dDinv16;
  a.ctl = e->dct;
  a.flags = e->dct + 4;
  a.ntasks = (int)ntasks;
  a.timeout_us = 4000000u;
  a.V = V;
  a.ldv = ldv;
  a.ntm = ntm;
  double flops = 0.0;
  for (int c = 0; c < nct; ++c)
    flops += (double)ntm * (2.0 * TILE * TILE * TILE * c * (c == nct - 1 ? ragged_col_frac(e) : 1.0) + (double)TILE * TILE * TILE);
  const int grid = (int)std::min<long long>(ntasks, e->wg_slots / 2);
  ev_begin(e, ev_kind, flops, nct, ntm, (int)e->Np, 0);
  hipLaunchKernelGGL(trsm_tiles_kernel<8>, dim3(grid), dim3(512), 0, e->cur, a);
  ev_end(e);
  HIP_TRY(e, hipGetLastError());
  e->tt_used = true;
  return GMB_OK;
}

// One persistent launch for L^-T (rows), Sigma^-1 = U U^T into dW and the alpha parts -- with the factorisation's own tile
// tasks in the same launch (with_chol) or behind a factorisation that is already final (eval_tiles.hpp).
int grad_workspace(gmb_engine* e);
// 32-bit words of the evaluation launch's control block:
// [4 control | nrt x nct tile flags | 3 nct chain words | nct x nct U flags | nct counts of finished INV tasks per block row]
int64_t eval_tiles_words(int nct, int nrt) { return 4 + (int64_t)nrt * nct + 3 * (int64_t)nct + (int64_t)nct * nct + nct; }

// gmb_evaluate with a gradient runs as the fused launch (the test factorize_enqueue applies)
bool eval_will_fuse(const gmb_engine* e) {
  const int nblocks = (int)(e->Np / TILE);
  return !e->naive_leaf && (e->chol_scheme == 3 || e->chol_scheme < 0) && e->grad_scheme != 0 && e->grad_scheme != 1 &&
         nblocks <= e->tiles_max_blocks && (e->grad_scheme == 2 ? nblocks >= 1 : nblocks >= e->et_min_blocks);
}

int eval_tiles(gmb_engine* e, bool with_chol) {
  const int nct = (int)(e->Np / TILE), nrt = (int)(e->Nr / TILE);
  int rc;
  // The ticket counter runs ahead of the factorisation's latency chain by about as many tasks as there are workgroups: an
  // inverse task drawn before its diagonal block L(c, c) exists would sit on a compute unit waiting for it (lag 1 at
  // N = 10k: 75 us median per task, 18.3 ms per evaluation; a quarter of the block columns behind: none, 16.3 ms;
  // tools/gpu_eval_lag.py)
  const int lag = e->et_lag >= 0 ? e->et_lag : std::max(2, std::min(24, nct / 4));
  int nw = 8;
#ifdef GMB_TUNING
  if (const char* cw = getenv("GMB_ET_WAVES")) nw = atoi(cw) == 4 ? 4 : 8;
#endif
  // (a traced launch keeps the list its stamps are decoded with; the pair task exists for eight-wave workgroups only -- ADVICE r05)
  const int pairs = (nw == 8 && !e->ct_trace && (e->et_pairs > 0 || (e->et_pairs < 0 && nct >= e->et_pairs_min_blocks))) ? 1 : 0;
  if (e->et_key[0] != nct || e->et_key[1] != nrt || e->et_key[2] != (int)with_chol || e->et_key[3] != lag || e->et_key[4] != pairs) {
    std::vector<uint32_t> list;
    et_build_tasks(nct, nrt, with_chol, lag, list, pairs != 0);
    if ((rc = ensure(e, &e->det_tasks, &e->cap_et_tasks, (int64_t)list.size()))) return rc;
    HIP_TRY(e, hipMemcpy(e->det_tasks, list.data(), list.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    e->et_key[0] = nct;
    e->et_key[1] = nrt;
    e->et_key[2] = (int)with_chol;
    e->et_key[3] = lag;
    e->et_key[4] = pairs;
    e->et_ntasks = (int)list.size();
  }
  const int ntasks = e->et_ntasks;
  const int64_t chol_words = (int64_t)nrt * nct + 3 * (int64_t)nct;
  const int64_t words = eval_tiles_words(nct, nrt);
  if ((rc = ensure(e, &e->dct, &e->cap_ct, words + 2))) return rc;
  if ((rc = ensure(e, &e->dW, &e->cap_W, e->Np * e->Np))) return rc;
  e->u_valid = e->linv_valid = false;  // (U is rewritten, Sigma^-1 takes dW: set again once the launch is known to have succeeded)
  if ((rc = ensure(e, &e->dUdiag, &e->cap_udiag, (int64_t)nct * TILE * TILE))) return rc;
  if ((rc = ensure(e, &e->dApart, &e->cap_apart, (int64_t)nct * nct * TILE + nct))) return rc;  // (+ the partials of |v|^2)
  if ((rc = grad_workspace(e))) return rc;
  if (!(with_chol && e->prezeroed)) HIP_TRY(e, hipMemsetAsync(e->dct, 0, (size_t)words * sizeof(uint32_t), e->cur));
  // a factor that is final already: every tile flag reads "final" (any non-zero word)
  if (!with_chol) HIP_TRY(e, hipMemsetAsync(e->dct + 4, 1, (size_t)nrt * nct * sizeof(uint32_t), e->cur));
  e->ct_injected = with_chol && e->ct_lose > 0;
  if (with_chol && e->ct_lose > 0) {  // fault injection: tickets 0 .. ct_lose-1 are never handed out
    const uint32_t first = (uint32_t)e->ct_lose;
    e->ct_lose = 0;
    HIP_TRY(e, hipMemcpyAsync(e->dct, &first, sizeof first, hipMemcpyHostToDevice, e->cur));
    HIP_TRY(e, hipStreamSynchronize(e->cur));
  }
  CholTilesArgs a{};
  a.A = e->dA;
  a.ld = e->ld;
  a.nct = nct;
  a.nrt = nrt;
  a.N = e->N;
  a.dinv16 = e->dDinv16;
  a.logdet = e->dscal;
  a.info = e->dinfo;
  a.ctl = e->dct;
  a.flags = e->dct + 4;
  a.half = e->dct + 4 + (int64_t)nrt * nct;
  a.prog = a.half + 2 * (int64_t)nct;
  a.ntasks = ntasks;
  a.timeout_us = 4000000u;
  a.dbg = nullptr;
  e->ct_traced = false;
  if (e->ct_trace) {
    e->ct_traced = true;
    if ((rc = ensure(e, &e->dct_trace, &e->cap_ct_trace, 4 * (int64_t)ntasks))) return rc;
    HIP_TRY(e, hipMemsetAsync(e->dct_trace, 0, (size_t)ntasks * 4 * sizeof(unsigned long long), e->cur));
    a.dbg = e->dct_trace;
  }
  EvalTilesArgs x{};
  x.tasks = e->det_tasks;
  x.ntasks = ntasks;
  x.uflags = e->dct + 4 + chol_words;
  x.udiag = e->dUdiag;
  x.Z = e->dW;
  x.ldz = e->Np;
  x.apart = e->dApart;
  x.yb = (int)(e->N / TILE);
  x.rowdone = x.uflags + (int64_t)nct * nct;
  x.alpha = e->dalpha;
  x.v = e->dv;
  x.vpart = e->dApart + (int64_t)nct * nct * TILE;
  x.scal = e->dscal + 1;
  x.with_v = with_chol ? 1 : 0;  // v = L^-1 y with |v|^2, which gmb_factorize takes with extract_v_kernel
  double flops = with_chol ? chol_tiles_flops(e, nct, nrt) : 0.0;
  for (int c = 0; c < nct; ++c)                                                                                       // INV column c
    flops += 2.0 * TILE * TILE * TILE * 0.5 * (double)c * (double)(c + 1) * (c == nct - 1 ? ragged_col_frac(e) : 1.0);
  for (int I = 0; I < nct; ++I) flops += 2.0 * TILE * TILE * TILE * (double)(nct - I) * (double)(I + 1);             // ZZ block row I
  const int grid = (int)std::min<long long>(ntasks, nw == 8 ? e->wg_slots / 2 : e->wg_slots);
  ev_begin(e, 9, flops, nct, nrt, (int)e->Np, with_chol ? 1 : 0);
  if (nw == 8) hipLaunchKernelGGL(eval_tiles_kernel<8>, dim3(grid), dim3(512), 0, e->cur, a, x);
#ifdef GMB_TUNING
  else hipLaunchKernelGGL(eval_tiles_kernel<4>, dim3(grid), dim3(256), 0, e->cur, a, x);
#endif
  ev_end(e);
  HIP_TRY(e, hipGetLastError());
  e->ct_used = true;
  e->ct_ntasks = ntasks;
  return GMB_OK;
}

// ---- cross-stream ordering helpers -----------------------------------------------------------
hipEvent_t next_sync_event(gmb_engine* e) {
  if (e->sync_next == e->sync_pool.size()) {
    hipEvent_t ev = nullptr;
    (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    e->sync_pool.push_back(ev);
  }
  return e->sync_pool[e->sync_next++];
}
hipEvent_t next_time_event(gmb_engine* e) {  // timing-enabled, reused from call to call
  if (e->time_next == e->time_pool.size()) {
    hipEvent_t ev = nullptr;
    (void)hipEventCreate(&ev);
    e->time_pool.push_back(ev);
  }
  return e->time_pool[e->time_next++];
}
// everything enqueued on `from` so far happens before anything enqueued on `to` from now on
int order_after(gmb_engine* e, hipStream_t from, hipStream_t to) {
  if (from == to) return GMB_OK;
  hipEvent_t ev = next_sync_event(e);
  HIP_TRY(e, hipEventRecord(ev, from));
  HIP_TRY(e, hipStreamWaitEvent(to, ev, 0));
  return GMB_OK;
}

int trsm_cols(gmb_engine* e, double* V, int64_t ldz, int ntm, int c0, int c1, int kind_gemm, int kind_strip,
              int lfirst = -1, int lstride = 0);

// Look-ahead Cholesky for SMALL matrices (<= masked_max_blocks block columns, N <= 16k): right-looking over
// panels; the latency-bound chain of panel p+1 (leaves, strips, in-panel products) runs on the main stream
// beside U2(p), the bulk of the trailing update, which runs on the process-wide CU-masked stream (it leaves 4
// compute units of every XCD to the chain).  Larger matrices use the plain recursion (chol_cols over the whole
// matrix, one stream): there the chain is a few per cent of the work, and bulk updates that share the chip with
// chain kernels lose more than the overlap wins -- r02: N = 50k 627 ms with look-ahead (bulk updates at 63
// TF/s) vs 607 ms plain (70.8 TF/s); N = 100k 5.02 vs 4.75 s; N = 24k / 32k 1 % apart; N = 16k equal.
int chol_lookahead_masked(gmb_engine* e) {
  const int nct = (int)(e->Np / TILE), nrt = (int)(e->Nr / TILE);
  const int w = e->panel_blocks;
  hipStream_t mainS = e->stream, bulkS = e->aux[2];
  e->sync_next = 0;
  int rc;
  e->chol_update_kind = 0;  // chol_cols below factors panels: its products belong to the chain
  e->cur = mainS;
  if ((rc = chol_cols(e, 0, std::min(w, nct), nrt))) return rc;
  for (int c0 = 0; c0 < nct; c0 += w) {
    const int c1 = std::min(c0 + w, nct);
    const int n0 = c1, n1 = std::min(c1 + w, nct);
    if (n0 >= nct) break;
    auto update = [&](int col_lo, int col_hi) {
      GemmArgs g{};
      g.C = e->dA + (int64_t)col_lo * TILE + (int64_t)col_lo * TILE * e->ld;
      g.ldc = e->ld;
      g.A = e->dA + (int64_t)col_lo * TILE + (int64_t)c0 * TILE * e->ld;
      g.lda = e->ld;
      g.B = g.A;
      g.ldb = e->ld;
      g.mt = col_hi - col_lo;
      g.nt = nrt - col_lo;
      g.k = (c1 - c0) * TILE;
      g.alpha = -1.0;
      g.beta = 1.0;
      g.tri = 1;
      return launch_gemm(e, g, 7);
    };
    e->cur = mainS;
    if ((rc = order_after(e, bulkS, mainS))) return rc;   // U2(p-1) reached these columns
    if ((rc = update(n0, n1))) return rc;                 // U1
    if ((rc = order_after(e, mainS, bulkS))) return rc;
    e->cur = bulkS;
    rc = (n1 < nct) ? update(n1, nct) : 0;          // U2, on the compute units its mask allows
    e->cur = mainS;
    if (rc) return rc;
    if ((rc = chol_cols(e, n0, n1, nrt))) return rc;      // panel p+1, beside U2
  }
  e->cur = mainS;
  return order_after(e, bulkS, mainS);
}

// ---- predict recursion: V <- W L^-T over column blocks [c0, c1) ------------------------------
// lfirst / lstride (optional): row tile t of V stands for block row lfirst + t*lstride of a matrix
// whose rows are zero left of their own diagonal block (rows of the identity being solved into
// rows of L^-T): the update products then skip the structurally zero part of their contraction.
int trsm_cols(gmb_engine* e, double* V, int64_t ldz, int ntm, int c0, int c1, int kind_gemm, int kind_strip,
              int lfirst, int lstride) {
  if (c1 - c0 == 1)
    return launch_trsm_strip(e, V + (int64_t)c0 * TILE * ldz, ldz, (int64_t)ntm * TILE,
                             e->dA + (int64_t)c0 * TILE * (e->ld + 1), e->ld, e->dDinv16 + (int64_t)c0 * 8 * 256,
                             (int)std::min<int64_t>(TILE, e->N - (int64_t)c0 * TILE), kind_strip);
  const int mid = c0 + (c1 - c0 + 1) / 2;
  int rc = trsm_cols(e, V, ldz, ntm, c0, mid, kind_gemm, kind_strip, lfirst, lstride);
  if (rc) return rc;
  GemmArgs g{};
  g.C = V + (int64_t)mid * TILE * ldz;
  g.ldc = ldz;
  g.A = e->dA + (int64_t)mid * TILE + (int64_t)c0 * TILE * e->ld;
  g.lda = e->ld;
  g.B = V + (int64_t)c0 * TILE * ldz;
  g.ldb = ldz;
  g.mt = c1 - mid;
  g.nt = ntm;
  g.k = (mid - c0) * TILE;
  g.alpha = -1.0;
  g.beta = 1.0;
  if (lstride > 0) {
    g.klo_n = 1;
    g.krow_stride = lstride;
    g.krow_off = (lfirst - c0) * TILE;
  }
  rc = launch_gemm(e, g, kind_gemm);
  if (rc) return rc;
  return trsm_cols(e, V, ldz, ntm, mid, c1, kind_gemm, kind_strip, lfirst, lstride);
}

// ---- NLML gradient ---------------------------------------------------------------------------
// Triangular inverse by recursive block inversion, kept in BOTH orientations so that every
// product is a k-major ("NT") MFMA GEMM with fully coalesced operand loads:
//   W = L^-1   (lower) in e->dW,
//   U = L^-T   (upper) in the upper triangle + diagonal tiles of the factor buffer e->dA
//              (the diagonal blocks of L are no longer needed once inv(L_kk) exists; the
//              off-diagonal blocks of L stay intact in the lower triangle).
//   [[A,0],[B,C]]^-1 = [[A^-1,0],[-C^-1 B A^-1, C^-1]]:
//     T^T = U_A B^T (scratch, mirrored region of dW),  W21 = -W_C T,  U12 = W21^T (transpose).
int launch_transpose(gmb_engine* e, const double* src, int64_t lds_, double* dst, int64_t ldd, int rows,
                     int cols) {
  hipLaunchKernelGGL(transpose_kernel, dim3((rows + 31) / 32, (cols + 31) / 32), dim3(256), 0, e->cur, src,
                     lds_, dst, ldd, rows, cols);
  HIP_TRY(e, hipGetLastError());
  return GMB_OK;
}

int winv_node(gmb_engine* e, int c0, int c1, bool recurse);

int winv_cols(gmb_engine* e, int c0, int c1) { return winv_node(e, c0, c1, true); }

// The two halves of a node are independent (both read only L and the inv(L_kk) blocks), so the
// whole recursion tree can be executed level by level, bottom-up, with the nodes of a level dealt
// round-robin over four streams: the many small GEMMs at the bottom of the tree (a few tiles
// each) run side by side instead of one after the other.
struct InvNode {
  int c0, c1;
};
void collect_inv_nodes(int c0, int c1, int depth, std::vector<std::vector<InvNode>>& levels) {
  if ((int)levels.size() <= depth) levels.resize(depth + 1);
  levels[depth].push_back(InvNode{c0, c1});
  if (c1 - c0 > 1) {
    const int mid = c0 + (c1 - c0 + 1) / 2;
    collect_inv_nodes(c0, mid, depth + 1, levels);
    collect_inv_nodes(mid, c1, depth + 1, levels);
  }
}
int build_inv_plan(gmb_engine* e, const std::vector<std::vector<InvNode>>& levels);
int winv_level_batched(gmb_engine* e, const gmb_engine::InvLevelPlan& lp);

int winv_levels(gmb_engine* e, int nt) {
  std::vector<std::vector<InvNode>> levels;
  collect_inv_nodes(0, nt, 0, levels);
  hipStream_t streams[4] = {e->stream, e->aux[0], e->aux[1], e->aux_shared ? e->aux[0] : e->aux[2]};
  // Small matrices (<= 64 block columns): only the two top levels are not batched, i.e. at most two nodes could run
  // side by side -- less than the event records / stream waits of the fork and the joins cost on the host, which is
  // what such a gradient is bound by (timeline at N = 2048: 70 us of idle GPU before the first batched level).  One
  // stream, no events.
  if (nt <= 64) streams[1] = streams[2] = streams[3] = e->stream;
  int rc;
  for (int a = 1; a < 4; ++a)
    if ((rc = order_after(e, streams[0], streams[a]))) return rc;
  const bool batched = e->batch_inverse && build_inv_plan(e, levels) == GMB_OK;
  bool aux_behind = false;  // the main stream has run batched levels the auxiliary streams have not waited for
  for (int depth = (int)levels.size() - 1; depth >= 0; --depth) {
    if (batched && e->inv_plan[depth].count > 0) {
      // consecutive batched levels live on the main stream alone: no event traffic between them (an
      // event record + three stream waits per level showed up as 20-25 us bubbles in the timeline)
      e->cur = e->stream;  // every stream was joined into the main one after the previous level
      if ((rc = winv_level_batched(e, e->inv_plan[depth]))) return rc;
      aux_behind = true;
      continue;
    }
    if (aux_behind) {
      for (int a = 1; a < 4; ++a)
        if ((rc = order_after(e, streams[0], streams[a]))) return rc;
      aux_behind = false;
    }
    int idx = 0;
    for (const InvNode& nd : levels[depth]) {
      e->cur = streams[levels[depth].size() > 1 ? (idx++ & 3) : 0];
      if ((rc = winv_node(e, nd.c0, nd.c1, false))) {
        e->cur = e->stream;
        return rc;
      }
    }
    e->cur = e->stream;
    for (int a = 1; a < 4; ++a)
      if ((rc = order_after(e, streams[a], streams[0]))) return rc;
    if (depth > 0)
      for (int a = 1; a < 4; ++a)
        if ((rc = order_after(e, streams[0], streams[a]))) return rc;
  }
  return GMB_OK;
}

// The two products and the transpose of one merge node [c0, c1) of the inverse tree.
void winv_node_products(gmb_engine* e, int c0, int c1, GemmArgs& gt, GemmArgs& gw, TransposeJob& tr) {
  const int64_t ldw = e->Np, lda = e->ld;
  double* W = e->dW;
  double* A = e->dA;
  const int mid = c0 + (c1 - c0 + 1) / 2;
  const int n1 = mid - c0, n2 = c1 - mid;
  // T^T[j][r] = sum_{k>=j} U_A[j][k] * B[r][k]   -> scratch at W[c0.., mid..]
  gt = GemmArgs{};
  gt.C = W + (int64_t)c0 * TILE + (int64_t)mid * TILE * ldw;
  gt.ldc = ldw;
  gt.A = A + (int64_t)mid * TILE + (int64_t)c0 * TILE * lda;  // B = L21
  gt.lda = lda;
  gt.B = A + (int64_t)c0 * TILE + (int64_t)c0 * TILE * lda;   // U_A (upper triangle of the factor buffer)
  gt.ldb = lda;
  gt.mt = n2;
  gt.nt = n1;
  gt.k = n1 * TILE;
  gt.klo_n = 1;
  gt.alpha = 1.0;
  gt.beta = 0.0;
  gt.nblk_stride = 1;
  // W21[r][j] = -sum_{s<=r} W_C[r][s] * T^T[j][s]
  gw = GemmArgs{};
  gw.C = W + (int64_t)mid * TILE + (int64_t)c0 * TILE * ldw;
  gw.ldc = ldw;
  gw.A = W + (int64_t)c0 * TILE + (int64_t)mid * TILE * ldw;   // T^T
  gw.lda = ldw;
  gw.B = W + (int64_t)mid * TILE + (int64_t)mid * TILE * ldw;  // W_C
  gw.ldb = ldw;
  gw.mt = n1;
  gw.nt = n2;
  gw.k = n2 * TILE;
  gw.khi_n = 1;
  gw.alpha = -1.0;
  gw.beta = 0.0;
  gw.nblk_stride = 1;
  // U12 = W21^T into the upper triangle of the factor buffer
  tr.src = W + (int64_t)mid * TILE + (int64_t)c0 * TILE * ldw;
  tr.dst = A + (int64_t)c0 * TILE + (int64_t)mid * TILE * lda;
  tr.rows = n2 * TILE;
  tr.cols = n1 * TILE;
}

int winv_node(gmb_engine* e, int c0, int c1, bool recurse) {
  if (c1 - c0 == 1) return GMB_OK;  // leaf_invert_kernel wrote both diagonal blocks already
  const int mid = c0 + (c1 - c0 + 1) / 2;
  int rc;
  if (recurse) {
    if ((rc = winv_node(e, c0, mid, true))) return rc;
    if ((rc = winv_node(e, mid, c1, true))) return rc;
  }
  GemmArgs gt, gw;
  TransposeJob tr;
  winv_node_products(e, c0, c1, gt, gw, tr);
  if ((rc = launch_gemm(e, gt, 4))) return rc;
  if ((rc = launch_gemm(e, gw, 4))) return rc;
  return launch_transpose(e, tr.src, e->Np, tr.dst, e->ld, tr.rows, tr.cols);
}

// ---- batched levels ---------------------------------------------------------------------------
// Levels of the tree with at least MIN_BATCH_NODES merge nodes run as three launches (all T^T
// products, all W21 products, all transposes) whose descriptors are built once per data set.
constexpr int MIN_BATCH_NODES = 4;

int build_inv_plan(gmb_engine* e, const std::vector<std::vector<InvNode>>& levels) {
  if (e->plan_N == e->N && e->plan_ld == e->ld && e->plan_Np == e->Np && e->plan_W == e->dW && e->plan_A == e->dA)
    return GMB_OK;
  std::vector<GemmArgs> hg;
  std::vector<TransposeJob> ht;
  e->inv_plan.assign(levels.size(), gmb_engine::InvLevelPlan{});
  static const int BMs[3] = {128, 64, 128}, BNs[3] = {128, 64, 64};
  static const double rate[3] = {73.0, 63.0, 65.0};
  static const int per128[3] = {1, 4, 2};
  for (size_t depth = 0; depth < levels.size(); ++depth) {
    std::vector<InvNode> nodes;
    for (const InvNode& nd : levels[depth])
      if (nd.c1 - nd.c0 > 1) nodes.push_back(nd);
    if ((int)nodes.size() < MIN_BATCH_NODES) continue;
    gmb_engine::InvLevelPlan& lp = e->inv_plan[depth];
    lp.count = (int)nodes.size();
    std::vector<GemmArgs> gs[2];
    lp.toff = (int64_t)ht.size();
    for (const InvNode& nd : nodes) {
      GemmArgs gt, gw;
      TransposeJob tr;
      winv_node_products(e, nd.c0, nd.c1, gt, gw, tr);
      gs[0].push_back(gt);
      gs[1].push_back(gw);
      ht.push_back(tr);
      lp.tr_rows = std::max(lp.tr_rows, tr.rows);
      lp.tr_cols = std::max(lp.tr_cols, tr.cols);
    }
    for (int b = 0; b < 2; ++b) {
      long long nact = 0;
      for (const GemmArgs& g : gs[b]) nact += gemm_nact128(g);
      int variant = 0;
      double best = 0.0;
      for (int v = 0; v < 3; ++v) {  // the launch_gemm model on the level's total tile count
        const double t = (double)(nact * per128[v]);
        const double sc = rate[v] * per128[v] / std::max(1.0, t / (double)e->wg_slots + 0.5);
        if (sc > best) {
          best = sc;
          variant = v;
        }
      }
      lp.variant[b] = variant;
      lp.off[b] = (int64_t)hg.size();
      for (GemmArgs g : gs[b]) {
        g.mt = g.mt * TILE / BMs[variant];
        g.nt = g.nt * TILE / BNs[variant];
        g.order = e->lpt_order ? (b == 0 ? 1 : 2) : 0;
        double fl = 0.0;
        const int nb = gemm_schedule(g, BMs[variant], BNs[variant], &fl);
            lp.grid_x[b] = std::max(lp.grid_x[b], nb);
        lp.flops[b] += fl;
        hg.push_back(g);
      }
    }
  }
  if (e->dplan_gemm) (void)hipFree(e->dplan_gemm);
  if (e->dplan_tr) (void)hipFree(e->dplan_tr);
  e->dplan_gemm = nullptr;
  e->dplan_tr = nullptr;
  e->plan_N = -1;
  if (!hg.empty()) {
    HIP_TRY(e, hipMalloc((void**)&e->dplan_gemm, hg.size() * sizeof(GemmArgs)));
    HIP_TRY(e, hipMalloc((void**)&e->dplan_tr, ht.size() * sizeof(TransposeJob)));
    HIP_TRY(e, hipMemcpy(e->dplan_gemm, hg.data(), hg.size() * sizeof(GemmArgs), hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->dplan_tr, ht.data(), ht.size() * sizeof(TransposeJob), hipMemcpyHostToDevice));
  }
  e->plan_N = e->N;
  e->plan_ld = e->ld;
  e->plan_Np = e->Np;
  e->plan_W = e->dW;
  e->plan_A = e->dA;
  return GMB_OK;
}

int winv_level_batched(gmb_engine* e, const gmb_engine::InvLevelPlan& lp) {
  for (int b = 0; b < 2; ++b) {
    if (lp.grid_x[b] <= 0) continue;
    const GemmArgs* batch = e->dplan_gemm + lp.off[b];
    const dim3 grid(lp.grid_x[b], lp.count);
    ev_begin(e, 4, lp.flops[b], lp.count, 0, 0, (lp.variant[b] << 8) | 64);
    switch (lp.variant[b]) {
      case 0: hipLaunchKernelGGL((gemm_f64_batched_kernel<2, 2, 4, 4, 2>), grid, dim3(256), 0, e->cur, batch); break;
      case 1: hipLaunchKernelGGL((gemm_f64_batched_kernel<2, 2, 2, 2, 2>), grid, dim3(256), 0, e->cur, batch); break;
      default: hipLaunchKernelGGL((gemm_f64_batched_kernel<2, 2, 4, 2, 2>), grid, dim3(256), 0, e->cur, batch); break;
    }
    ev_end(e);
    HIP_TRY(e, hipGetLastError());
  }
  hipLaunchKernelGGL(transpose_batched_kernel, dim3((lp.tr_rows + 31) / 32, (lp.tr_cols + 31) / 32, lp.count),
                     dim3(256), 0, e->cur, e->dplan_tr + lp.toff, e->Np, e->ld);
  HIP_TRY(e, hipGetLastError());
  return GMB_OK;
}

template <int KIND>
int launch_grad_nc(gmb_engine* e, const GradArgs& a_in, int nblocks) {
  const dim3 grid(nblocks), block(256);
  const size_t lds = grad_lds_bytes(e->nc_pad, a_in.p.n_lin, a_in.p.n_tab);
  GradArgs a = a_in;
  a.part_block0 = 0;
  const int gs = a.gsplit > 1 ? a.gsplit : 1;
  const dim3 grid0((a.split ? a.general_tiles : nblocks) * gs);  // (split: one tile of the general-tile list per gs workgroups)
  switch (e->nc_pad) {
    case 1: hipLaunchKernelGGL((grad_tile_kernel<KIND, 1>), grid0, block, lds, e->stream, a); break;
    case 2: hipLaunchKernelGGL((grad_tile_kernel<KIND, 2>), grid0, block, lds, e->stream, a); break;
    case 4: hipLaunchKernelGGL((grad_tile_kernel<KIND, 4>), grid0, block, lds, e->stream, a); break;
    case 8: hipLaunchKernelGGL((grad_tile_kernel<KIND, 8>), grid0, block, lds, e->stream, a); break;
    default: hipLaunchKernelGGL((grad_tile_kernel<KIND, 16>), grid0, block, lds, e->stream, a); break;
  }
  if constexpr (KIND <= 2) {
    if (a.split) {  // the interior tiles on the matrix pipe: runs of the full enumeration, the partial vectors behind the first set
      a.part_block0 = a.general_tiles * gs;
      switch (e->nc_pad) {
        case 1: hipLaunchKernelGGL((grad_interior_kernel<KIND, 1>), grid, block, 0, e->stream, a); break;
        case 2: hipLaunchKernelGGL((grad_interior_kernel<KIND, 2>), grid, block, 0, e->stream, a); break;
        case 4: hipLaunchKernelGGL((grad_interior_kernel<KIND, 4>), grid, block, 0, e->stream, a); break;
        case 8: hipLaunchKernelGGL((grad_interior_kernel<KIND, 8>), grid, block, 0, e->stream, a); break;
        default: hipLaunchKernelGGL((grad_interior_kernel<KIND, 16>), grid, block, 0, e->stream, a); break;
      }
    }
  }
  HIP_TRY(e, hipGetLastError());
  return GMB_OK;
}

// ---- gradient building blocks ---------------------------------------------------------------------
// (a) grad_sigma_inv_rows: block rows shard, shard + nshards, ... of Sigma^-1 = U U^T (lower triangle) from
//     U = L^-T in the upper triangle of the factor buffer, into Z -- the full Np x Np matrix (packed = false:
//     rows at their global position, ldz = Np) or ONLY the owned rows, packed (multi-GPU: ldz = owned * 128);
// (b) grad_reduce: the fused trace reductions over those rows, one pass per covariance term (additive models
//     have several); term t accumulates into region t of dgpart: [ls.. | eta | tau | c.. ] at 0, its tables
//     from 64.  `h` receives the raw accumulators: plain sums over tiles, so shards of several GPUs add up.
int grad_workspace(gmb_engine* e) {
  int rc;
  if (e->cap_pts_alpha < e->Np) {
    if ((rc = alloc(e, &e->dalpha, e->Np))) return rc;
    e->cap_pts_alpha = e->Np;
  }
  return GMB_OK;  // (the accumulators live in the engine's scalar block)
}

int grad_sigma_inv_rows(gmb_engine* e, int shard, int nshards, double* Z, int64_t ldz, bool packed) {
  const int nt = (int)(e->Np / TILE);
  const int owned = shard < nt ? (nt - shard + nshards - 1) / nshards : 0;
  GemmArgs g{};
  g.C = packed ? Z : Z + (int64_t)shard * TILE;
  g.ldc = ldz;
  g.A = e->dA;
  g.lda = e->ld;
  g.B = e->dA + (int64_t)shard * TILE;
  g.ldb = e->ld;
  g.mt = nt;
  g.nt = owned;
  g.k = (int)e->Np;
  g.klo_n = 1;
  g.krow_off = shard * TILE;
  g.tri = 1;
  g.tri_off = shard * TILE;
  g.nblk_stride = nshards;
  g.cblk_stride = packed ? 1 : 0;
  g.alpha = 1.0;
  g.beta = 0.0;
  return launch_gemm(e, g, 4);
}

int grad_reduce(gmb_engine* e, int shard, int nshards, const double* Z, int64_t ldz, bool packed,
                std::vector<double>& h) {
  const gmb_kernel_spec& s = e->spec;
  const int nt = (int)(e->Np / TILE);
  int rc;
  if (!e->gacc_zeroed) HIP_TRY(e, hipMemsetAsync(e->dgpart, 0, (GACC_DOUBLES + 64) * sizeof(double), e->stream));
  e->gacc_zeroed = false;
  const int n_ls = s.ard ? s.n_cont : 1;
  long long total = 0;
  for (int i = shard; i < nt; i += nshards) total += i + 1;
  // persistent workgroups: six per compute unit, each reduces a contiguous run of `per` tiles in registers / LDS
  // and writes ONE partial vector; a second launch adds the vectors in a fixed order (bit-reproducible results)
  const long long want = std::max<long long>(1, 3 * e->wg_slots);  // 6 workgroups per compute unit: one resident round
  const int per = (int)std::max<long long>(1, (total + want - 1) / want);
  const int grid = (int)((total + per - 1) / per);
  const int ncp = e->nc_pad;
  bool merged = false;  // the pass ends in grad_finish_kernel (set inside the one term's iteration)
  GradFinishArgs fin{};
  for (size_t t = 0; t < e->terms.size(); ++t) {
    const gmb_engine::Term& tr = e->terms[t];
    if (t > 0 && (rc = prep_points(e, e->dX, e->N, e->D, e->Nr, e->xs, e->xl, e->cat, &tr.pa))) return rc;
    GradArgs a{};
    a.p = tr.cp;
    a.pts = train_set(e);
    a.Z = Z;
    a.ldz = ldz;
    a.z_packed = packed ? 1 : 0;
    a.alpha = e->dalpha;
    a.tiles = nt;
    a.ard = s.ard;
    a.nc_real = s.n_cont;
    for (int k = 0; k < 16; ++k) a.inv_ls[k] = k < s.n_cont ? tr.pa.inv_ls[k] : 0.0;
    a.eta = tr.eta;
    a.row_first = shard;
    a.row_stride = nshards;
    a.total_tiles = total;
    a.per = per;
    const int n_small = ncp + 2 + tr.cp.n_lin;
    a.part_stride = n_small + tr.cp.n_tab * 64;
    double* acc = e->dgpart + (int64_t)t * GACC_REGION;
    int tab_acc_off[MAX_TABS] = {0};
    int off = 64, big = 0;
    for (int j = 0; j < tr.cp.n_tab; ++j) {
      tab_acc_off[j] = off;
      const int LL = tr.cp.tab_levels[j] * tr.cp.tab_levels[j];
      off += LL;
      a.big_off[j] = big;
      if (tr.cp.tab_levels[j] > 8) big += LL;
    }
    a.big_stride = big;
    // smooth stationary terms: the full tiles strictly below the diagonal go to grad_interior_kernel (gradient.hpp)
    a.split = (tr.cp.kind == GMB_EXPQUAD || tr.cp.kind == GMB_MATERN52 || tr.cp.kind == GMB_MATERN32) && tr.cp.n_lin == 0 &&
                      tr.cp.n_tab == 0 && !e->naive_leaf
                  ? 1 : 0;
    // the tiles the interior launch leaves: the diagonal tile of every owned block row + a ragged last row's other tiles
    a.n_owned_rows = shard < nt ? (nt - shard + nshards - 1) / nshards : 0;
    const bool last_ragged_owned = e->Np > e->N && (nt - 1 - shard) % nshards == 0 && nt - 1 >= shard;
    a.general_tiles = a.n_owned_rows + (last_ragged_owned ? nt - 1 : 0);
    if (a.general_tiles == 0) a.split = 0;
    // small matrices: every tile through the direct loops of ONE launch, four workgroups per tile (N = 392: the interior launch's
    // 10.8 us behind the general tiles' 10.3 saved; from 129 tiles on the matrix-pipe form wins)
    if (total <= 128) a.split = 0;
    // small launches: four workgroups per direct-loop tile (GradArgs::gsplit)
    const int ndirect = a.split ? a.general_tiles : grid;  // runs / list tiles of grad_tile_kernel
    a.gsplit = ndirect <= 512 ? 4 : 1;
    const int nvec = a.split ? ndirect * a.gsplit + grid : ndirect * a.gsplit;  // partial vectors the pass leaves
    if (grid > 0) {
      if ((rc = ensure(e, &e->dgred, &e->cap_gred, (int64_t)nvec * a.part_stride))) return rc;
      a.part = e->dgred;
      if (big > 0) {
        if ((rc = ensure(e, &e->dgbig, &e->cap_gbig, (int64_t)ndirect * a.gsplit * 4 * big))) return rc;
        HIP_TRY(e, hipMemsetAsync(e->dgbig, 0, (size_t)ndirect * a.gsplit * 4 * big * sizeof(double), e->stream));
        a.big = e->dgbig;
      }
      switch (tr.cp.kind) {
        case GMB_EXPQUAD: rc = launch_grad_nc<0>(e, a, grid); break;
        case GMB_MATERN52: rc = launch_grad_nc<1>(e, a, grid); break;
        case GMB_MATERN32: rc = launch_grad_nc<2>(e, a, grid); break;
        case GMB_MATERN12: rc = launch_grad_nc<3>(e, a, grid); break;
        default: rc = launch_grad_nc<4>(e, a, grid); break;
      }
      if (rc) return rc;
      // second stage: dense slots of the partial vectors -> the term's accumulator region
      GradRanges r{};
      auto add = [&](int dense, int count, int dst) {
        r.dense[r.n] = dense;
        r.count[r.n] = count;
        r.dst[r.n] = dst;
        ++r.n;
      };
      double* tmp = e->dgpart + GACC_DOUBLES;  // scratch behind the accumulators
      merged = e->light && e->terms.size() == 1 && s.ard && big == 0 && nvec <= 8192 && e->N <= 16384;
      if (s.ard) add(0, s.n_cont, 0);
      else add(0, s.n_cont, (int)(tmp - acc));               // per-dimension sums, folded into the one parameter below
      add(ncp, 2 + tr.cp.n_lin, n_ls);                        // eta | tau | c..
      for (int j = 0; j < tr.cp.n_tab; ++j)
        if (tr.cp.tab_levels[j] <= 8) add(n_small + j * 64, tr.cp.tab_levels[j] * tr.cp.tab_levels[j], tab_acc_off[j]);
      if (merged) {
        // small evaluation: the second stage, the diagonal terms and the landing are ONE launch (grad_finish_kernel)
        fin.part = e->dgred;
        fin.nparts = nvec;
        fin.nslots = a.part_stride;
        fin.stride = a.part_stride;
        fin.r = r;
        fin.acc = acc;
      } else {
        hipLaunchKernelGGL(grad_sum_partials_kernel, dim3(a.part_stride), dim3(256), 0, e->stream, e->dgred, nvec,
                           (int64_t)a.part_stride, r, acc);
      }
      if (!s.ard) hipLaunchKernelGGL(grad_fold_ls_kernel, dim3(1), dim3(64), 0, e->stream, acc, s.n_cont, tmp);
      if (big > 0) {
        GradRanges rb{};
        for (int j = 0; j < tr.cp.n_tab; ++j)
          if (tr.cp.tab_levels[j] > 8) {
            rb.dense[rb.n] = a.big_off[j];
            rb.count[rb.n] = tr.cp.tab_levels[j] * tr.cp.tab_levels[j];
            rb.dst[rb.n] = tab_acc_off[j];
            ++rb.n;
          }
        hipLaunchKernelGGL(grad_sum_partials_kernel, dim3(big), dim3(256), 0, e->stream, e->dgbig, ndirect * a.gsplit * 4, (int64_t)big, rb, acc);
      }
      HIP_TRY(e, hipGetLastError());
    }
    if (t == 0) {  // diagonal terms (sigma, noise table) with the global term's categories in place
      const double sigma = e->theta[n_ls + 1];
      if (merged) {
        fin.Z = Z;
        fin.ldz = ldz;
        fin.alpha = e->dalpha;
        fin.pts = train_set(e);
        fin.p = tr.cp;
        fin.sigma = sigma;
        fin.diag_out = e->dgpart + off;
      } else {
        hipLaunchKernelGGL(grad_diag_kernel, dim3(1), dim3(1024), 0, e->stream, Z, ldz, e->dalpha, train_set(e),
                           tr.cp, sigma, e->dgpart + off, shard, nshards, packed ? 1 : 0);
        HIP_TRY(e, hipGetLastError());
      }
    }
  }
  if (e->terms.size() > 1 &&
      (rc = prep_points(e, e->dX, e->N, e->D, e->Nr, e->xs, e->xl, e->cat, &e->terms[0].pa)))
    return rc;
  h.assign(GACC_DOUBLES, 0.0);
  if (e->light) {
    // gmb_evaluate on the fused launch: scalars, failure index, abort word and the used head of every accumulator region go to
    // the pinned landing in one launch; grad_accumulate hands them on after its synchronisation
    EvalLandArgs la{};
    la.scal = e->dscal;
    la.info = e->dinfo;
    la.abort = e->ct_used ? e->dct + 1 : nullptr;
    la.gacc = e->dgpart;
    la.nterms = (int)e->terms.size();
    la.region = GACC_REGION;
    for (size_t t = 0; t < e->terms.size(); ++t) {
      int used = 64;
      for (int j = 0; j < e->terms[t].cp.n_tab; ++j) used += e->terms[t].cp.tab_levels[j] * e->terms[t].cp.tab_levels[j];
      la.used[t] = std::min(used + 64, GACC_REGION);  // (+ the diagonal terms' slots behind the tables)
    }
    la.out_scal = e->hl_dev->scal;
    la.out_info = &e->hl_dev->info;
    la.out_abort = &e->hl_dev->abort;
    la.out_gacc = e->hl_dev->gacc;
    if (merged) {
      fin.land = la;
      hipLaunchKernelGGL(grad_finish_kernel, dim3(1), dim3(1024), 0, e->stream, fin);
    } else hipLaunchKernelGGL(eval_land_kernel, dim3(1), dim3(256), 0, e->stream, la);
    HIP_TRY(e, hipGetLastError());
    return GMB_OK;
  }
  HIP_TRY(e, hipMemcpyAsync(h.data(), e->dgpart, GACC_DOUBLES * sizeof(double), hipMemcpyDeviceToHost,
                            e->stream));
  return GMB_OK;
}

// matrices whose inverse / Sigma^-1 run as tile tasks (eval_tiles.hpp): the sizes the tile Cholesky factors
bool grad_by_tiles(const gmb_engine* e) {
  const int nt = (int)(e->Np / TILE);
  if (e->naive_leaf || e->grad_scheme == 0) return false;
  if (e->grad_scheme > 0) return nt >= 1 && nt <= 0x7fff;
  return nt >= e->tiles_min_blocks && nt <= e->tiles_max_blocks;
}

// Single-GPU gradient: W = L^-1 and U = L^-T by recursive block inversion, alpha = W^T v, Sigma^-1 = U U^T
// into dW, reductions.  The factor is consumed.
int grad_accumulate(gmb_engine* e, std::vector<double>& h) {
  HIP_TRY(e, hipSetDevice(e->device));
  if (e->factor_consumed)
    return fail(e, GMB_EINVAL, "the factor was already consumed by a gradient call; refactorize");
  int rc;
  gmb_timings& tm = e->tm;
  tm.grad_ms = tm.grad_gemm_ms = tm.grad_gemm_flops = 0.0;
  if ((rc = ensure(e, &e->dW, &e->cap_W, e->Np * e->Np))) return rc;
  if ((rc = grad_workspace(e))) return rc;
  const int nt = (int)(e->Np / TILE);
  if (e->et_fused || grad_by_tiles(e)) {
    // Sigma^-1 and the alpha parts come from the persistent evaluation launch: already enqueued with the factorisation
    // (gmb_evaluate), or launched here behind a factor that is final.  L stays intact: nothing to save or restore.
    const bool light = e->light && e->et_fused;  // (gmb_evaluate: results land through eval_land_kernel, see grad_reduce)
    const bool timed = !(light && !e->fe_recorded);
    std::unique_ptr<PhaseTimer> tgt(timed ? new PhaseTimer(e) : nullptr);
    bool own_launch = false;
    if (!e->et_fused) {
      e->cur = e->stream;
      if ((rc = eval_tiles(e, false))) return rc;
      own_launch = true;
    }
    e->et_fused = false;
    e->light = light;
    rc = grad_reduce(e, 0, 1, e->dW, e->Np, false, h);
    e->light = false;
    if (rc) return rc;
    if (tgt) tgt->stop();
    uint32_t ab = 0;
    if (own_launch) HIP_TRY(e, hipMemcpyAsync(&ab, e->dct + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    if (tgt) tm.grad_ms = tgt->ms();
    ev_collect(e);
    if (light) {
      for (size_t t = 0; t < e->terms.size(); ++t)
        std::copy(e->hl->gacc + t * GACC_REGION, e->hl->gacc + (t + 1) * GACC_REGION, h.begin() + t * GACC_REGION);
      e->have_alpha = true;
      e->u_valid = true;  // (gmb_evaluate takes both back if factorize_finish reports a failure)
      return GMB_OK;  // (abort word, failure index: factorize_finish, which gmb_evaluate calls next)
    }
    if (ab != 0) return fail(e, GMB_EHIP, "tile inverse: a workgroup waited longer than its time-out for a tile (launch abandoned)");
    e->have_alpha = true;
    e->u_valid = true;
    return GMB_OK;
  }
  if ((rc = ensure(e, &e->dDiagSave, &e->cap_diag, (int64_t)nt * TILE * TILE))) return rc;
  e->u_valid = e->linv_valid = false;
  PhaseTimer tg(e);
  // the diagonal blocks of L are put back at the end: U takes their place in between
  hipLaunchKernelGGL(diag_blocks_copy_kernel, dim3(nt), dim3(256), 0, e->stream, e->dA, e->ld, e->dDiagSave, 1);
  // 0. inverses of all diagonal factor blocks, one workgroup each (kept off the Cholesky's chain)
  {
    InvArgs ia;
    ia.L = e->dA;
    ia.lda = e->ld;
    ia.blk_stride = (int64_t)TILE * (e->ld + 1);
    ia.dinv16 = e->dDinv16;
    ia.invL = nullptr;
    ia.n = e->N;
    ia.W = e->dW;
    ia.ldw = e->Np;
    ia.U = e->dA;
    ia.ldu = e->ld;
    hipLaunchKernelGGL(leaf_invert_kernel, dim3(nt), dim3(256), 0, e->stream, ia);
    HIP_TRY(e, hipGetLastError());
  }
  // 1. W = L^-1 (dW, lower) and U = L^-T (factor buffer, upper); the factor is consumed from here on
  e->factor_consumed = true;
  if (!e->fact_in_flight) e->sync_next = 0;  // (gmb_evaluate: the factorisation's cross-stream waits may still be pending on pooled events)
  if (e->par_inverse) {
    if ((rc = winv_levels(e, nt))) return rc;
  } else if ((rc = winv_cols(e, 0, nt))) {
    return rc;
  }
  // 2. alpha = W^T v = Sigma^-1 y   (before Sigma^-1 overwrites W)
  hipLaunchKernelGGL(wt_v_kernel, dim3((unsigned)((e->N + 3) / 4)), dim3(256), 0, e->stream, e->dW, e->Np,
                     e->dv, e->N, e->dalpha);
  if (e->Np > e->N)
    hipLaunchKernelGGL(reset_pad_cols_kernel, dim3((unsigned)((e->Np + 255) / 256)), dim3(256), 0, e->stream,
                       e->dA, e->ld, e->N, e->Np);
  HIP_TRY(e, hipGetLastError());
  // 3. Sigma^-1 = U U^T (lower triangle) into dW -- the last reader of U: the factor's diagonal blocks go back
  if ((rc = grad_sigma_inv_rows(e, 0, 1, e->dW, e->Np, false))) return rc;
  hipLaunchKernelGGL(diag_blocks_copy_kernel, dim3(nt), dim3(256), 0, e->stream, e->dA, e->ld, e->dDiagSave, 0);
  HIP_TRY(e, hipGetLastError());
  // 4. fused trace reductions
  if ((rc = grad_reduce(e, 0, 1, e->dW, e->Np, false, h))) return rc;
  tg.stop();
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  tm.grad_ms = tg.ms();
  ev_collect(e);
  e->factor_consumed = false;  // lower triangle = L again (the upper triangle keeps U: scratch as far as L goes)
  e->have_alpha = true;
  return GMB_OK;
}

// Host part: chain rule from the (summed) accumulators back to the packed natural-scale parameters.
int grad_chain_rule(gmb_engine* e, const std::vector<double>& h, double* grad) {
  const gmb_kernel_spec& s = e->spec;
  const int n_ls = s.ard ? s.n_cont : 1;
  const double* th = e->theta.data();
  auto region = [&](size_t t) { return h.data() + t * GACC_REGION; };
  auto table_slot = [&](size_t t, int j) {  // j-th table of term t inside its region
    int off = 64;
    for (int q = 0; q < j; ++q) off += e->terms[t].cp.tab_levels[q] * e->terms[t].cp.tab_levels[q];
    return region(t) + off;
  };
  auto kernel_block = [&](size_t t, double* g) {  // [ls | eta] and, after a gap the caller fills, [c | tau]
    const double* r = region(t);
    int k = 0;
    for (int i = 0; i < n_ls; ++i) g[k++] = r[i];
    g[k++] = r[n_ls];
    return k;
  };
  auto table_grad = [&](const std::vector<double>& G, int L, const double* W, double* g) {
    for (int x = 0; x < L; ++x)
      for (int q = 0; q < 2; ++q) {
        double acc = 0.0;
        for (int b = 0; b < L; ++b) acc += (G[x * L + b] + G[b * L + x]) * W[2 * b + q];
        g[2 * x + q] = acc;
      }
    for (int x = 0; x < L; ++x) g[2 * L + x] = G[x * L + x];
  };
  // global term: ls | eta | sigma | c | tau
  int k = kernel_block(0, grad);
  const int i_sigma = k++;
  if (s.n_lin > 0) {
    for (int i = 0; i < s.n_lin; ++i) grad[k++] = region(0)[n_ls + 2 + i];
    grad[k++] = region(0)[n_ls + 1];  // tau
  }
  // coregion tables of the categorical dims, then the output table
  const int ntab = spec_ntab(s);
  for (int j = 0; j < ntab; ++j) {
    const bool is_out = (j == s.n_coreg);
    const int L = is_out ? s.n_out : s.coreg_levels[j];
    std::vector<double> G((size_t)L * L, 0.0);
    if (!s.additive) {
      const double* src = table_slot(0, j);
      for (int i = 0; i < L * L; ++i) G[i] = src[i];
    } else if (!is_out) {
      const double* src = table_slot(1 + j, 0);  // the dim's own term carries its table first
      for (int i = 0; i < L * L; ++i) G[i] = src[i];
    } else {
      for (size_t t = 0; t < e->terms.size(); ++t) {  // every term is multiplied by the output table
        const double* src = table_slot(t, e->terms[t].cp.n_tab - 1);
        for (int i = 0; i < L * L; ++i) G[i] += src[i];
      }
    }
    table_grad(G, L, th + k, grad + k);
    k += 3 * L;
  }
  // noise: region 0 keeps [sigma, noise table ...] after the global term's tables
  int diag_off = 64;
  for (int j = 0; j < e->terms[0].cp.n_tab; ++j) diag_off += e->terms[0].cp.tab_levels[j] * e->terms[0].cp.tab_levels[j];
  const double* hd = region(0) + diag_off;
  grad[i_sigma] = hd[0];
  if (s.out_col >= 0 && s.hetero_noise) {
    const int P = s.n_out;
    const double* W = th + k;
    for (int x = 0; x < P; ++x) {
      const double gd = hd[1 + x];
      grad[k + 2 * x] = 2.0 * gd * W[2 * x];
      grad[k + 2 * x + 1] = 2.0 * gd * W[2 * x + 1];
      grad[k + 2 * P + x] = gd;
    }
    k += 3 * P;
  }
  // additive terms: ls | eta | c | tau per coregion dim
  if (s.additive)
    for (int j = 0; j < s.n_coreg; ++j) {
      k += kernel_block(1 + j, grad + k);
      if (s.n_lin > 0) {
        for (int i = 0; i < s.n_lin; ++i) grad[k++] = region(1 + j)[n_ls + 2 + i];
        grad[k++] = region(1 + j)[n_ls + 1];
      }
    }
  if (k != (int)e->theta.size()) return fail(e, GMB_EINVAL, "internal: gradient packing mismatch");
  return GMB_OK;
}

// -1 = by size (default), 0 = plain recursion, 2 = masked look-ahead streams, 3 = persistent tile kernel
inline bool chol_scheme_valid(int s) { return s == -1 || s == 0 || s == 2 || s == 3 || s == 4; }

int require_ready(gmb_engine* e, bool need_factor) {
  if (!e) return GMB_EINVAL;
  if (e->N <= 0) return fail(e, GMB_EINVAL, "gmb_set_data has not been called");
  if (!e->have_spec) return fail(e, GMB_EINVAL, "gmb_set_kernel has not been called");
  if (!e->have_theta) return fail(e, GMB_EINVAL, "gmb_set_theta has not been called");
  if (need_factor && !e->factored)
    return fail(e, GMB_EINVAL, "no valid factorisation: call gmb_factorize first");
  return GMB_OK;
}

// The calls that read the COMPLETE factor out of dA (single-engine gradient / prediction / factor windows, the replicated
// multi-GPU passes) against a factorisation of the capacity driver, which leaves none there -- and the reverse: the capacity
// passes against anything else than a capacity factorisation (their dAown would be an older theta's).
int require_full_factor(gmb_engine* e, const char* what) {
  if (e->factor_kind == gmb_engine::FK_CAPACITY || !e->dA || e->cap_A <= 0)
    return fail(e, GMB_EINVAL, "%s needs the complete factor on this device, but the resident factorisation is a capacity-mode one "
                               "(gmb_dist_set_mode(e, 1): no rank holds the factor) -- use the gmb_dist_* calls, or gmb_factorize first", what);
  return GMB_OK;
}

int require_capacity_factor(gmb_engine* e, const char* what) {
  if (e->factor_kind != gmb_engine::FK_CAPACITY || !e->dAown)
    return fail(e, GMB_EINVAL, "%s in capacity mode needs a factorisation made by gmb_dist_factorize in capacity mode; the resident "
                               "one was made by %s", what, e->factor_kind == gmb_engine::FK_SINGLE ? "gmb_factorize / gmb_evaluate" : "the replicated multi-GPU driver");
  return GMB_OK;
}

}  // namespace

namespace {

// ONE CU-masked bulk stream per process and device, created on first use and never destroyed (masked
// streams that were created and destroyed with every engine made kernel timings depend on the queue
// history of the process).  It leaves `part_cus / 8` compute units of every XCD free for the latency-bound
// panel chain of chol_lookahead_masked; the runtime applies a CU mask symmetrically to all XCDs (bit i <->
// XCD i % 8, compute unit i / 8 of it -- tools/probes/cumask_probe.hip).  Engines on one device share it,
// i.e. their trailing updates serialise on it; every engine joins it back into its own stream before
// gmb_factorize returns.
hipStream_t shared_masked_stream(int device, int part_cus, int* ncu_out) {
  static std::mutex mu;
  static hipStream_t streams[64] = {};
  static int ncus[64] = {};
  std::lock_guard<std::mutex> lock(mu);
  hipStream_t& st = streams[device & 63];
  if (!st) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return nullptr;
    const int ncu = prop.multiProcessorCount;
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    for (int i = part_cus; i < ncu; ++i) mask[i / 32] |= 1u << (i % 32);
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
      st = nullptr;
      (void)hipGetLastError();
      return nullptr;
    }
    ncus[device & 63] = ncu;
  }
  if (ncu_out) *ncu_out = ncus[device & 63];
  return st;
}

int gmb_create_impl(gmb_engine** out, int32_t device, void* stream, const gmb_engine* peer = nullptr) {
  if (!out) return GMB_EINVAL;
  *out = nullptr;
  int n = gmb_device_count();
  if (n < 0) return GMB_ENODEVICE;
  if (device < 0 || device >= n) return GMB_EINVAL;
  if (hipSetDevice(device) != hipSuccess) return GMB_EHIP;
  gmb_engine* e = new gmb_engine();
  e->device = device;
  if (stream) {
    e->stream = (hipStream_t)stream;
  } else {
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) {
      delete e;
      return GMB_EHIP;
    }
    e->own_stream = true;
  }
  auto flag = [](const char* name, bool dflt) {
    const char* v = getenv(name);
    return v && v[0] ? v[0] != '0' : dflt;
  };
  // Environment switches of the PRODUCT library: only the ones a test or a documented tools/ A-B uses.
  //   GMB_LEAF_NAIVE=1   the reference diagonal-block kernel (tests/test_gpu_parity.py::test_potrf_leaf_block)
  //   GMB_CHOL_SCHEME    0 = plain recursion (launches only), 2 = masked look-ahead, 3 = persistent tile kernel, 4 = plain recursion
  //                      with its bottom panels on the tile kernel (what large matrices take by size), for every size
  //                      (tests/test_gpu_parity.py::test_cholesky_schedules_agree); default: by size
  //   GMB_TRACE_FILE     per-launch event log while profiling (tools/gpu_timeline_run.py)
  //   GMB_GEMM_DMA=0     the 128 x 128 GEMM with register staging instead of the LDS-DMA ring (tools/gpu_ab_gemm_dma.py)
  // Everything else is compiled in only with -DGMB_TUNING (GUMBI_BUILD_TUNING=1 python -m gumbi_amd.build writes
  // gumbi_amd/lib/libgumbi_hip_tuning.so; tools/README.md).
  e->naive_leaf = flag("GMB_LEAF_NAIVE", false);
  e->gemm_dma = flag("GMB_GEMM_DMA", true);
  const char* cs = getenv("GMB_CHOL_SCHEME");
  if (cs && cs[0]) {
    const int v = atoi(cs);
    if (chol_scheme_valid(v)) e->chol_scheme = v;
    else fprintf(stderr, "[gumbi_hip] GMB_CHOL_SCHEME=%s ignored: valid schemes are -1 (by size), 0, 2, 3, 4\n", cs);
  }
  int part = 32;  // compute units the masked bulk stream leaves to the panel chain
#ifdef GMB_TUNING
  e->small_tiles = flag("GMB_SMALL_TILES", true);
  const char* gv = getenv("GMB_GEMM_VARIANT");  // pins the tile shape of every out-of-place GEMM: 0 .. 3
  if (gv && gv[0] >= '0' && gv[0] <= '4') {
    e->gemm_variant = gv[0] - '0';
    e->force_variant = true;
  }
  e->lookahead = flag("GMB_LOOKAHEAD", true);
  e->lpt_order = flag("GMB_LPT_ORDER", true);
  if (const char* lm = getenv("GMB_LPT_MAX_TILES")) e->lpt_max_tiles = atoll(lm);
  const char* ts = getenv("GMB_TILE_STRIP");
  if (ts) e->tile_strip = std::max(0, atoi(ts));
  const char* cst = getenv("GMB_COV_STRIP");
  if (cst) e->cov_strip = std::max(0, std::min(64, atoi(cst)));
  e->batch_inverse = flag("GMB_BATCH_INVERSE", true);
  e->par_inverse = flag("GMB_PAR_INVERSE", true);
  const char* pb = getenv("GMB_PANEL_BLOCKS");
  if (pb && atoi(pb) > 0) {
    e->panel_blocks = atoi(pb);
    e->panel_auto = false;
  }
  const char* mb = getenv("GMB_MASKED_MAX_BLOCKS");
  if (mb) e->masked_max_blocks = atoi(mb);
  if (const char* pw = getenv("GMB_PANEL_TILES_W")) e->panel_tiles_w = std::max(0, std::min(64, atoi(pw)));  // (tools/gpu_panel_tiles_ab.py)
  const char* pc = getenv("GMB_PART_CUS");
  if (pc) part = atoi(pc);
#endif
  e->cur = e->stream;
  // Exactly four streams per engine (beyond four hardware queues the runtime multiplexes streams and every
  // kernel of the process slows down -- measured): the caller's, aux[0] / aux[1] at the highest queue
  // priority (panel chain of the unmasked look-ahead; independent merges of the triangular inverse), and
  // aux[2] = the bulk stream of the trailing updates: the process-wide CU-masked stream (GMB_PART_CUS
  // compute units left free, default 32; tuning builds: 0 = an ordinary lowest-priority stream).
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  for (int a = 0; a < 3 && peer; ++a) {  // sibling: the peer's streams (its engines work one after the other)
    e->aux[a] = peer->aux[a];
    e->aux_borrowed = true;
    e->aux_shared = peer->aux_shared;
    e->part_cus = peer->part_cus;
  }
  for (int a = 0; a < 3 && !peer; ++a) {
    hipError_t st2 = hipSuccess;
    int ncu = 0;
    hipStream_t masked = (a == 2 && part > 0) ? shared_masked_stream(device, part, &ncu) : nullptr;
    if (masked) {
      e->aux[a] = masked;
      e->aux_shared = true;
      e->part_cus = part;
    } else {
      st2 = hipStreamCreateWithPriority(&e->aux[a], hipStreamNonBlocking, a == 2 ? prio_lo : prio_hi);
    }
    if (st2 != hipSuccess) {
      gmb_destroy(e);
      return GMB_EHIP;
    }
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) e->wg_slots = 2LL * prop.multiProcessorCount;
  if (hipMalloc((void**)&e->dscal, (size_t)SCAL_DOUBLES * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&e->dstat, (size_t)(DIST_HEADER + DIST_MAX_PAYLOAD) * (1 + DIST_MAX_WORLD) * sizeof(double)) != hipSuccess) {
    gmb_destroy(e);
    return GMB_ENOMEM;
  }
  e->dinfo = reinterpret_cast<int32_t*>(e->dscal + SCAL_INFO_AT);
  e->dgpart = e->dscal + SCAL_GACC_AT;
  *out = e;
  return GMB_OK;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int gmb_abi_version(void) { return GMB_ABI_VERSION; }

int gmb_device_count(void) {
  int n = 0;
  hipError_t s = hipGetDeviceCount(&n);
  if (s != hipSuccess || n <= 0) return GMB_ENODEVICE;
  return n;
}

int gmb_create(gmb_engine** out, int32_t device, void* stream) { return gmb_create_impl(out, device, stream); }

int gmb_create_sibling(gmb_engine** out, const gmb_engine* peer) {
  if (!peer) return GMB_EINVAL;
  return gmb_create_impl(out, peer->device, (void*)peer->stream, peer);
}

void gmb_destroy(gmb_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  // the engine's own work on every stream it used (the shared bulk stream is joined into e->stream at the
  // end of every factorisation, but an error return may have skipped that)
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  for (int a = 0; a < 3; ++a)
    if (e->aux[a]) (void)hipStreamSynchronize(e->aux[a]);
  void* ptrs[] = {e->dct, e->dct_trace, e->dstat, e->dDiagSave, e->dsend, e->drecv, e->dplan_gemm, e->dplan_tr, e->dDinv16, e->dX, e->dy, e->dA, e->xs, e->xl, e->cat, e->dtabs,
                  e->dnoise, e->dscal, e->dv, e->dV, e->dXs, e->txs, e->txl,
                  e->tcat, e->dkss, e->dpart, e->dmean, e->dvar, e->dW, e->dalpha, e->dgred, e->dgbig, e->det_tasks, e->dUdiag, e->dApart,
                  e->dAown, e->dPanel, e->dV2, e->dsend2, e->drecv2};
  if (e->cap_A == 0) ptrs[11] = nullptr;  // (e->dA may be a virtual base of the capacity driver: nothing of ours to free)
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (int a = 0; a < 3; ++a)
    if (e->aux[a] && !e->aux_borrowed && !(a == 2 && e->aux_shared)) (void)hipStreamDestroy(e->aux[a]);
  for (auto ev : e->fe)
    if (ev) (void)hipEventDestroy(ev);
  if (e->hl) (void)hipHostFree(e->hl);
  for (auto ev : e->sync_pool) (void)hipEventDestroy(ev);
  for (auto ev : e->time_pool) (void)hipEventDestroy(ev);
  if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

const char* gmb_last_error(const gmb_engine* e) { return e ? e->err.c_str() : "null engine"; }
void* gmb_stream(const gmb_engine* e) { return e ? (void*)e->stream : nullptr; }

int gmb_theta_size(const gmb_kernel_spec* s) {
  if (!s) return GMB_EINVAL;
  int n = (s->ard ? s->n_cont : 1) + 2;
  if (s->n_lin > 0) n += s->n_lin + 1;
  for (int t = 0; t < s->n_coreg; ++t) n += 3 * s->coreg_levels[t];
  if (s->out_col >= 0) {
    n += 3 * s->n_out;
    if (s->hetero_noise) n += 3 * s->n_out;
  }
  if (s->additive) {
    const int blk = (s->ard ? s->n_cont : 1) + 1 + (s->n_lin > 0 ? s->n_lin + 1 : 0);
    n += s->n_coreg * blk;
  }
  return n;
}

int gmb_set_data(gmb_engine* e, const double* X, int64_t N, int32_t D, int64_t ldx, const double* y,
                 int32_t memspace) {
  if (!e) return GMB_EINVAL;
  if (!X || !y || N < 1 || D < 1 || ldx < D) return fail(e, GMB_EINVAL, "bad X/y/N/D/ldx");
  if (e->have_spec) {
    int rc = validate_spec(e, e->spec, D);
    if (rc) return rc;
  }
  HIP_TRY(e, hipSetDevice(e->device));
  // not ready until every allocation below has succeeded; workspaces sized for the previous N / D are dropped
  e->factored = false;
  e->factor_kind = gmb_engine::FK_NONE;  // (no factorisation of THIS data has been attempted: gmb_debug_assume_factored must refuse)
  e->have_theta = false;
  e->N = 0;
  e->cap_pts = 0;
  e->Mt_cap = 0;
  e->Mt2_cap = 0;
  e->u_valid = e->linv_valid = false;
  e->cap_part = 0;
  e->plan_N = -1;
  e->D = D;
  e->Np = round_up(N, TILE);
  e->Nr = round_up(N + 1, TILE);
  e->ld = e->Nr;
#ifdef GMB_TUNING
  if (const char* lp = getenv("GMB_LD_PAD")) e->ld = e->Nr + atoi(lp);  // probe: leading dimension off the 1 KiB grid (tools/gpu_ld_pad.py)
#endif
  int rc;
  if ((rc = alloc(e, &e->dX, N * (int64_t)D))) return rc;
  if ((rc = alloc(e, &e->dy, e->Np))) return rc;
  // (the factor buffer itself -- Nr x Np doubles -- is allocated by the first factorisation that needs it: in the multi-GPU
  // driver's capacity mode no rank ever holds it)
  if (e->cap_A < e->ld * e->Np && e->dA) {
    release(e, e->dA);
    e->dA = nullptr;
    e->cap_A = 0;
  }
  if ((rc = ensure(e, &e->dDinv16, &e->cap_dinv16, (e->Np / TILE) * (int64_t)8 * 256))) return rc;
  if ((rc = alloc(e, &e->dv, e->Np))) return rc;
  const hipMemcpyKind kind = memspace == GMB_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  HIP_TRY(e, hipMemcpy2DAsync(e->dX, D * sizeof(double), X, ldx * sizeof(double), D * sizeof(double), N,
                              kind, e->stream));
  HIP_TRY(e, hipMemsetAsync(e->dy, 0, e->Np * sizeof(double), e->stream));
  HIP_TRY(e, hipMemcpyAsync(e->dy, y, N * sizeof(double), kind, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  e->N = N;
  return GMB_OK;
}

int gmb_set_y(gmb_engine* e, const double* y, int32_t memspace) {
  if (!e) return GMB_EINVAL;
  if (e->N <= 0) return fail(e, GMB_EINVAL, "gmb_set_data must precede gmb_set_y");
  if (!y) return fail(e, GMB_EINVAL, "y is null");
  HIP_TRY(e, hipSetDevice(e->device));
  const hipMemcpyKind kind = memspace == GMB_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  e->factored = false;  // the y row of the factor (v = L^-1 y) belongs to the old observations
  e->factor_kind = gmb_engine::FK_NONE;
  HIP_TRY(e, hipMemcpyAsync(e->dy, y, e->N * sizeof(double), kind, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));  // the caller may reuse its buffer
  return GMB_OK;
}

int gmb_set_kernel(gmb_engine* e, const gmb_kernel_spec* spec) {
  if (!e || !spec) return GMB_EINVAL;
  int rc = validate_spec(e, *spec, e->D);
  if (rc) return rc;
  e->spec = *spec;
  e->have_spec = true;
  e->have_theta = false;
  e->factored = false;
  e->factor_kind = gmb_engine::FK_NONE;
  e->nc_pad = pick_nc(spec->n_cont);
  e->cap_pts = 0;
  return GMB_OK;
}

int gmb_set_theta(gmb_engine* e, const double* theta, int32_t n) {
  if (!e || !theta) return GMB_EINVAL;
  if (!e->have_spec) return fail(e, GMB_EINVAL, "gmb_set_kernel must precede gmb_set_theta");
  if (e->N <= 0) return fail(e, GMB_EINVAL, "gmb_set_data must precede gmb_set_theta");
  if (n != gmb_theta_size(&e->spec))
    return fail(e, GMB_EINVAL, "theta has %d entries, kernel needs %d", n, gmb_theta_size(&e->spec));
  for (int i = 0; i < n; ++i)
    if (!std::isfinite(theta[i])) return fail(e, GMB_EINVAL, "theta[%d] is not finite", i);
  HIP_TRY(e, hipSetDevice(e->device));
  e->theta.assign(theta, theta + n);
  e->factored = false;
  e->factor_kind = gmb_engine::FK_NONE;
  int rc = apply_theta(e);
  if (rc) return rc;
  if (e->cap_pts < e->Nr) {
    if ((rc = alloc(e, &e->xs, (int64_t)17 * e->Nr))) return rc;
    if ((rc = alloc(e, &e->xl, (int64_t)MAX_LIN * e->Nr))) return rc;
    if ((rc = alloc(e, &e->cat, (int64_t)MAX_TABS * e->Nr))) return rc;
    e->cap_pts = e->Nr;
  }
  e->prezeroed = false;
  if (e->eval_call && eval_will_fuse(e) && e->ct_lose == 0) {
    // gmb_evaluate: this launch also clears the scalar block and the tile launch's control words (factorize_enqueue, eval_tiles)
    const int64_t words = eval_tiles_words((int)(e->Np / TILE), (int)(e->Nr / TILE));
    if ((rc = ensure(e, &e->dct, &e->cap_ct, words + 2))) return rc;
    e->prep_zero[0] = reinterpret_cast<unsigned long long*>(e->dscal);
    e->prep_zero_words[0] = SCAL_DOUBLES;
    e->prep_zero[1] = reinterpret_cast<unsigned long long*>(e->dct);
    e->prep_zero_words[1] = (words + 1) / 2;
    e->prezeroed = true;
  }
  // small evaluations: the covariance build of this same gmb_evaluate prepares the points itself (one launch less)
  e->prep_deferred = e->prezeroed && e->terms.size() == 1 && e->Nr <= e->inline_prep_max_rows && !e->naive_leaf;
  if (!e->prep_deferred && (rc = prep_points(e, e->dX, e->N, e->D, e->Nr, e->xs, e->xl, e->cat))) return rc;
  e->have_theta = true;
  return GMB_OK;
}

int gmb_set_profiling(gmb_engine* e, int32_t on) {
  if (!e) return GMB_EINVAL;
  e->profiling = on != 0;
  if (on) {
    e->tm.total_gemm_ms = e->tm.total_gemm_flops = 0.0;
    e->tm.masked_gemm_ms = e->tm.masked_gemm_flops = 0.0;
    e->tm.total_gemm_wall_ms = 0.0;
    e->tm.total_chol_gemm_ms = e->tm.total_chol_gemm_flops = e->tm.total_chol_gemm_wall_ms = 0.0;
    e->tm.total_chol_gemm_launches = 0;
    e->tm.total_chol_panel_gemm_ms = e->tm.total_chol_panel_gemm_flops = 0.0;
    e->tm.total_chol_tile_ms = e->tm.total_chol_tile_flops = 0.0;
    e->tm.total_chol_tile_launches = 0;
    e->tm.total_chol_update_all_ms = e->tm.total_chol_update_all_flops = 0.0;
    e->tm.total_chol_update_all_launches = 0;
    e->tm.total_chol_panel_tile_ms = e->tm.total_chol_panel_tile_flops = 0.0;
    e->tm.total_chol_panel_tile_launches = 0;
    e->tm.masked_cus = e->aux_shared ? e->wg_slots / 2 - e->part_cus : 0;
    e->tm.total_gemm_launches = 0;
    e->tm.total_kbuild_ms = e->tm.total_kbuild_bytes = 0.0;
    e->tm.total_kbuild_launches = 0;
  }
  return GMB_OK;
}

int gmb_timings_get(const gmb_engine* e, gmb_timings* out) {
  if (!e || !out) return GMB_EINVAL;
  *out = e->tm;
  return GMB_OK;
}

int64_t gmb_notpd_index(const gmb_engine* e) { return e ? e->notpd : -1; }

int gmb_factor_valid(const gmb_engine* e) { return (e && e->N > 0 && e->have_theta && e->factored && !e->factor_consumed) ? 1 : 0; }

namespace {

// Everything of gmb_factorize up to (not including) the host's look at the results: K-build, Cholesky, v = L^-1 y, and the
// asynchronous copies of log-det / |v|^2 / failure index / abort word into the engine.  Nothing is synchronised.
int factorize_enqueue(gmb_engine* e, bool with_grad = false) {
  int rc = require_ready(e, false);
  if (rc) return rc;
  HIP_TRY(e, hipSetDevice(e->device));
  e->factored = false;
  e->factor_kind = gmb_engine::FK_SINGLE;
  e->factor_consumed = false;
  e->have_alpha = false;
  e->u_valid = e->linv_valid = false;
  e->notpd = -1;
  if ((rc = ensure_factor_buffer(e))) return rc;
  gmb_timings& tm = e->tm;
  tm.kbuild_ms = tm.chol_ms = tm.chol_gemm_ms = tm.chol_gemm_flops = 0.0;
  tm.chol_leaf_ms = tm.chol_trsm_ms = 0.0;
  tm.chol_gemm_launches = 0;
  for (auto& ev : e->fe)
    if (!ev) HIP_TRY(e, hipEventCreate(&ev));
  if (e->panel_auto) {
    // wider panels for larger matrices: a k = 1024 trailing update pays its C read-modify-write and
    // epilogue per 1024 of contraction; measured at N = 60k: 8 blocks 58.0, 16: 62.2, 32: 63.9 TF/s
    // (N = 10k: 4 .. 16 within 2 %)
    const int nct = (int)(e->Np / TILE);
    e->panel_blocks = std::max(8, ((nct / 16 + 4) / 8) * 8);
  }
  const int nblocks = (int)(e->Np / TILE);
  const bool tiles = !e->naive_leaf && (e->chol_scheme == 3 || (e->chol_scheme < 0 && nblocks >= e->tiles_min_blocks &&
                                                                 nblocks <= e->tiles_max_blocks));
  const bool masked = !tiles && e->lookahead && e->aux_shared && e->Np / TILE > e->panel_blocks &&
                      (e->chol_scheme == 2 || (e->chol_scheme < 0 && e->Np / TILE <= e->masked_max_blocks));
  // the fused evaluation launch pays at every size (per evaluation, stream schedules -> fused: N = 100: 0.21 -> 0.14 ms, N = 392:
  // 0.44 -> 0.27, N = 1000: 0.76 -> 0.45; tools/gpu_small_eval.py), the tile Cholesky on its own only from six block columns
  const bool fused = with_grad && eval_will_fuse(e);
  // gmb_evaluate on the fused launch: everything the host wants lands through eval_land_kernel at the end of the gradient
  e->light = e->eval_call && fused;
  const bool events = !(e->light && !e->profiling && e->Np < 4096);
  e->fe_recorded = events;
  // scalars + failure index (+ the gradient's accumulators): one memset -- or none, when the evaluation's prep_points launch
  // has cleared them already (gmb_evaluate)
  e->prezeroed = e->prezeroed && e->light;
  if (!e->prezeroed) HIP_TRY(e, hipMemsetAsync(e->dscal, 0, (size_t)(e->light ? SCAL_DOUBLES : SCAL_GACC_AT) * sizeof(double), e->stream));
  e->gacc_zeroed = e->light;

  // 1. covariance build: lower-triangular tiles of Sigma, y row, identity padding
  if (events) HIP_TRY(e, hipEventRecord(e->fe[0], e->stream));
  if ((rc = build_sigma(e, e->dA, e->ld))) return rc;
  if (events) HIP_TRY(e, hipEventRecord(e->fe[1], e->stream));
  // 2. Cholesky
  if (events) HIP_TRY(e, hipEventRecord(e->fe[2], e->stream));
  e->chol_update_kind = 7;
  e->ct_used = false;
  e->ct_traced = false;
  e->cur = e->stream;
  e->et_fused = false;
  if (fused) {
    // the gradient's inverse and Sigma^-1 ride in the factorisation's launch (eval_tiles.hpp); v = L^-1 y (row N of the factor)
    // and |v|^2 come out of the launch that adds up alpha
    if ((rc = eval_tiles(e, true))) return rc;
    e->et_fused = true;
  } else {
    e->ct_panels = false;
    if (tiles) {
      rc = chol_tiles(e);
    } else if (masked) {
      rc = chol_lookahead_masked(e);
    } else {
      // the plain recursion; its bottom panels as launches of the tile kernel (chol_tiles_panel)
      e->panel_tiles_on = !e->naive_leaf && !e->panel_tiles_veto && e->panel_tiles_w >= 2 && e->chol_scheme != 0;
      if (e->panel_tiles_on) {
        if ((rc = ensure(e, &e->dct, &e->cap_ct, 4 + (int64_t)(e->panel_tiles_w + 3) * (e->Nr / TILE)))) return rc;
        HIP_TRY(e, hipMemsetAsync(e->dct, 0, 4 * sizeof(uint32_t), e->stream));  // (the abort word: once per factorisation)
      }
      rc = chol_cols(e, 0, nblocks, (int)(e->Nr / TILE));
      e->panel_tiles_on = false;
    }
    if (rc) return rc;
    // 3. v = L^-1 y is row N of the factor
    hipLaunchKernelGGL(extract_v_kernel, dim3(EXTRACT_V_BLOCKS), dim3(TILE), 0, e->stream, e->dA, e->ld, e->N, e->dv,
                       e->dscal + 1);
  }
  if (events) HIP_TRY(e, hipEventRecord(e->fe[3], e->stream));
  HIP_TRY(e, hipGetLastError());
  if (!e->hl) {
    HIP_TRY(e, hipHostMalloc((void**)&e->hl, sizeof(gmb_engine::HostLanding), hipHostMallocDefault));
    HIP_TRY(e, hipHostGetDevicePointer((void**)&e->hl_dev, e->hl, 0));
  }
  e->hl->info = 0;
  e->hl->abort = 0;
  if (!e->light) {
    HIP_TRY(e, hipMemcpyAsync(e->hl->scal, e->dscal, 2 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipMemcpyAsync(&e->hl->info, e->dinfo, sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    if (e->ct_used) HIP_TRY(e, hipMemcpyAsync(&e->hl->abort, e->dct + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
  }
  e->fact_in_flight = true;
  return GMB_OK;
}

// The host's look at an enqueued factorisation: synchronise, timings, failure checks; sets e->factored.
int factorize_finish(gmb_engine* e) {
  gmb_timings& tm = e->tm;
  e->factored = false;
  e->fact_in_flight = false;
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  if (!e->hl) return fail(e, GMB_EINVAL, "internal: no factorisation enqueued");
  if (e->hl->abort != 0) {
    ev_collect(e);
    return fail(e, GMB_EHIP, "tile Cholesky: a workgroup waited longer than its time-out for a tile (launch abandoned)");
  }
  float t = 0.f;
  if (e->fe_recorded && hipEventElapsedTime(&t, e->fe[0], e->fe[1]) == hipSuccess) tm.kbuild_ms = t;
  if (e->fe_recorded && hipEventElapsedTime(&t, e->fe[2], e->fe[3]) == hipSuccess) tm.chol_ms = t;
  tm.kbuild_bytes = 8.0 * (double)e->N * (double)(e->N + 1) / 2.0 +
                    8.0 * (double)e->N * (double)(e->spec.n_cont + 1);
  ev_collect(e);
  if (e->profiling) {
    tm.total_kbuild_ms += tm.kbuild_ms;
    tm.total_kbuild_bytes += tm.kbuild_bytes;
    tm.total_kbuild_launches += 1;
  }
  if (e->hl->info != 0) {
    e->notpd = (int64_t)e->hl->info - 1;
    return fail(e, GMB_ENOTPD, "covariance matrix is not positive definite at row %lld",
                (long long)e->notpd);
  }
  e->logdet = e->hl->scal[0];
  e->vnorm2 = e->hl->scal[1];
  if (!std::isfinite(e->logdet) || !std::isfinite(e->vnorm2)) {
    e->notpd = 0;
    return fail(e, GMB_ENOTPD, "factorisation produced non-finite values");
  }
  e->factored = true;
  return GMB_OK;
}

}  // namespace

// A tile-kernel launch that gave up waiting (abort word set: a lost flag, or a GPU shared with a process that starved it
// for seconds) says nothing about the matrix: factor it once more with the plain recursion before reporting an error.
// Injected faults (gmb_debug_chol_lose_tickets) are reported as they are -- the tests want to see the bounded wait.
static int factorize_retry_after_abort(gmb_engine* e, int rc) {
  if (rc != GMB_EHIP || !e->ct_used || !e->hl || e->hl->abort == 0 || e->ct_injected) return rc;
  fprintf(stderr, "libgumbi_hip: tile Cholesky abandoned after a time-out; refactorising with the plain recursion\n");
  const int scheme = e->chol_scheme;
  e->chol_scheme = 0;  // (scheme 0 = launches only: no tile kernel anywhere, the recursion's bottom panels included)
  int rc2 = factorize_enqueue(e);
  if (!rc2) rc2 = factorize_finish(e);
  e->chol_scheme = scheme;
  return rc2;
}

int gmb_factorize(gmb_engine* e) {
  int rc = factorize_enqueue(e);
  if (!rc) rc = factorize_retry_after_abort(e, factorize_finish(e));
  return rc;
}

int gmb_evaluate(gmb_engine* e, const double* theta, int32_t n, double* nlml, double* grad) {
  if (!e || !nlml) return GMB_EINVAL;
  e->eval_call = grad != nullptr;
  int rc = gmb_set_theta(e, theta, n);
  if (rc) {
    e->eval_call = false;
    return rc;
  }
  rc = factorize_enqueue(e, grad != nullptr);
  e->eval_call = false;
  e->prezeroed = false;
  if (e->prep_deferred) {  // the covariance build was not reached (an allocation failed): the points are prepared now
    const int rc2 = flush_deferred_prep(e);
    if (!rc) rc = rc2;
  }
  if (rc) {
    e->light = false;
    return rc;
  }
  int rc_grad = GMB_OK;
  std::vector<double> h;
  if (grad) {
    // the gradient's launches follow the factorisation's on the same stream with no host round trip in between; if the
    // factorisation turns out to have failed they ran on garbage and their result is dropped below
    e->factored = true;
    const bool light = e->light;
    rc_grad = grad_accumulate(e, h);
    e->light = false;
    if (rc_grad && light) {  // nothing has landed on the host: there is no factorisation to look at
      (void)hipStreamSynchronize(e->stream);
      e->factored = false;
      e->fact_in_flight = false;
      e->et_fused = false;
      return rc_grad;
    }
  }
  if ((rc = factorize_finish(e))) {
    // the gradient ran on garbage: nothing of it may outlive this call (gmb_copy_alpha checks have_alpha only)
    e->have_alpha = false;
    e->u_valid = e->linv_valid = false;
    e->factor_consumed = false;
    e->et_fused = false;
    if (factorize_retry_after_abort(e, rc) != GMB_OK) return rc;
    rc_grad = grad ? grad_accumulate(e, h) : GMB_OK;  // (an abandoned tile launch: the recursion's factor, the gradient again)
  }
  if (rc_grad) return rc_grad;
  *nlml = 0.5 * (double)e->N * std::log(2.0 * M_PI) + e->logdet + 0.5 * e->vnorm2;
  return grad ? grad_chain_rule(e, h, grad) : GMB_OK;
}

}  // extern "C"

namespace {

// M is tiled so that the solved cross-covariance (Mt x Np doubles) stays within ~8 GiB
int64_t predict_tile_rows(const gmb_engine* e, int64_t M) {
  int64_t mt_max = (int64_t)(8.0 * 1024 * 1024 * 1024 / 8.0 / (double)e->Np);
  mt_max = std::max<int64_t>(TILE, std::min<int64_t>(mt_max / TILE * TILE, 32768));
  return std::min<int64_t>(round_up(M, TILE), mt_max);
}

// Every allocation of gmb_predict for M-tiles of Mt rows.  The capacity driver calls it BEFORE its ranks agree to start a
// prediction: a rank that ran out of memory here would otherwise leave gmb_predict before the solve hook, i.e. without
// issuing the pass's all-gathers, with its peers blocked inside them.
int predict_workspace(gmb_engine* e, int64_t Mt) {
  int rc;
  if (e->Mt_cap < Mt) {
    e->Mt_cap = 0;  // (a failure half-way leaves no capacity claimed)
    if ((rc = alloc(e, &e->dV, Mt * e->Np))) return rc;
    if ((rc = alloc(e, &e->dXs, Mt * (int64_t)e->D))) return rc;
    if ((rc = alloc(e, &e->txs, (int64_t)17 * Mt))) return rc;
    if ((rc = alloc(e, &e->txl, (int64_t)MAX_LIN * Mt))) return rc;
    if ((rc = alloc(e, &e->tcat, (int64_t)MAX_TABS * Mt))) return rc;
    if ((rc = alloc(e, &e->dkss, Mt))) return rc;
    if ((rc = alloc(e, &e->dmean, Mt))) return rc;
    if ((rc = alloc(e, &e->dvar, Mt))) return rc;
    e->Mt_cap = Mt;
  }
  const int nchunk = (int)((e->N + RED_CHUNK - 1) / RED_CHUNK);
  return ensure(e, &e->dpart, &e->cap_part, (int64_t)2 * nchunk * e->Mt_cap);
}

}  // namespace

extern "C" {

int gmb_nlml(gmb_engine* e, double* nlml, double* grad) {
  int rc = require_ready(e, true);
  if (rc) return rc;
  if (!nlml) return fail(e, GMB_EINVAL, "nlml output pointer is null");
  *nlml = 0.5 * (double)e->N * std::log(2.0 * M_PI) + e->logdet + 0.5 * e->vnorm2;
  if (grad) {
    if ((rc = require_full_factor(e, "gmb_nlml with a gradient"))) return rc;
    std::vector<double> h;
    if ((rc = grad_accumulate(e, h))) return rc;
    return grad_chain_rule(e, h, grad);
  }
  return GMB_OK;
}

int gmb_predict(gmb_engine* e, const double* Xs, int64_t M, int64_t ldxs, int32_t with_noise,
                double* mean, double* var, int32_t memspace) {
  int rc = require_ready(e, true);
  if (rc) return rc;
  if (M < 0 || (M > 0 && (!Xs || !mean || !var)) || ldxs < e->D)
    return fail(e, GMB_EINVAL, "bad Xs/M/ldxs/mean/var");
  if (M == 0) return GMB_OK;
  // (the capacity driver's passes come through here with their own triangular solve; everybody else reads the factor in dA)
  if ((rc = e->solve_hook ? require_capacity_factor(e, "gmb_predict") : require_full_factor(e, "gmb_predict"))) return rc;
  if (e->factor_consumed)
    return fail(e, GMB_EINVAL, "the factor was consumed by a gradient call: call gmb_factorize again");
  HIP_TRY(e, hipSetDevice(e->device));
  gmb_timings& tm = e->tm;
  tm.predict_ms = tm.predict_gemm_ms = tm.predict_gemm_flops = 0.0;
  tm.predict_gemm_launches = 0;

  const int64_t Mt = predict_tile_rows(e, M);
  if ((rc = predict_workspace(e, Mt))) return rc;
  const int nchunk = (int)((e->N + RED_CHUNK - 1) / RED_CHUNK);
  // the GEMM form needs the inverse factor the tile path's gradient leaves behind, and a second Mt x Np buffer
  const bool use_gemm_form = !e->solve_hook && e->predict_form != 0 && e->u_valid && e->factor_kind == gmb_engine::FK_SINGLE &&
                             e->dW && e->dUdiag && e->cap_W >= e->Np * e->Np;
  if (use_gemm_form && e->Mt2_cap < Mt) {
    e->Mt2_cap = 0;
    if ((rc = alloc(e, &e->dV2, Mt * e->Np))) return rc;
    e->Mt2_cap = Mt;
  }
  bool gemm_form = false;
  tm.predict_gemm_form = use_gemm_form ? 1 : 0;

  const hipMemcpyKind in_kind = memspace == GMB_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  const hipMemcpyKind out_kind = memspace == GMB_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  PhaseTimer tp(e);
  for (int64_t m0 = 0; m0 < M; m0 += Mt) {
    const int64_t mc = std::min<int64_t>(Mt, M - m0);
    const int64_t mpad = round_up(mc, TILE);
    HIP_TRY(e, hipMemcpy2DAsync(e->dXs, e->D * sizeof(double), Xs + m0 * ldxs, ldxs * sizeof(double),
                                e->D * sizeof(double), mc, in_kind, e->stream));
    if ((rc = prep_points(e, e->dXs, mc, e->D, mpad, e->txs, e->txl, e->tcat))) return rc;
    const PointSet test{e->txs, e->txl, e->tcat, mc, mpad};
    {  // cross-covariance W[m + i*mpad] = k(x*_m, x_i)
      CovTileArgs a{};
      a.p = e->cp;
      a.rows = test;
      a.cols = train_set(e);
      a.out = e->dV;
      a.ldo = mpad;
      a.ti = (int)(mpad / TILE);
      a.tj = (int)(e->Np / TILE);
      a.mode = COV_CROSS;
      if ((rc = launch_cov(e, a))) return rc;
    }
    KssArgs k{};
    k.p = e->cp;
    k.pts = test;
    k.with_noise = with_noise;
    k.kss = e->dkss;
    hipLaunchKernelGGL(kss_kernel, dim3((unsigned)((mpad + 255) / 256)), dim3(256), 0, e->stream, k);
    for (size_t t = 1; t < e->terms.size(); ++t) {  // additive models: one accumulating pass per further term
      const gmb_engine::Term& tr = e->terms[t];
      if ((rc = prep_points(e, e->dX, e->N, e->D, e->Nr, e->xs, e->xl, e->cat, &tr.pa))) return rc;
      if ((rc = prep_points(e, e->dXs, mc, e->D, mpad, e->txs, e->txl, e->tcat, &tr.pa))) return rc;
      CovTileArgs a{};
      a.p = tr.cp;
      a.rows = test;
      a.cols = train_set(e);
      a.out = e->dV;
      a.ldo = mpad;
      a.ti = (int)(mpad / TILE);
      a.tj = (int)(e->Np / TILE);
      a.mode = COV_CROSS;
      a.accumulate = 1;
      if ((rc = launch_cov(e, a))) return rc;
      k.p = tr.cp;
      k.with_noise = 0;
      k.accumulate = 1;
      hipLaunchKernelGGL(kss_kernel, dim3((unsigned)((mpad + 255) / 256)), dim3(256), 0, e->stream, k);
    }
    if (e->terms.size() > 1 &&
        (rc = prep_points(e, e->dX, e->N, e->D, e->Nr, e->xs, e->xl, e->cat, &e->terms[0].pa)))
      return rc;
    {
      // the triangular solve V <- V L^-T: one persistent launch for the matrices the tile Cholesky factors, else the recursion
      const int nblocks = (int)(e->Np / TILE);
      // (measured window, r03: faster than the recursion whenever the launch has enough tile tasks to fill the chip --
      // row tiles x block columns >= 576: 1000 test points at N = 10k 4.05 -> 3.25 ms, 2500 at N = 5k 2.15 -> 1.86, 640 at
      // N = 16k 6.55 -> 5.24, 384 at N = 28k 12.7 -> 8.9; 5 - 10 % for the standard 10^4-point grid -- and slower below
      // that (a few row tiles walk the block columns as a chain either way, the recursion's launches are lighter), and for
      // 4 x 10^4 points, where the recursion's large GEMMs run at 73 TF/s against the persistent loop's 69;
      // scheme 3 forces it for any size; tools/gpu_chol_tiles.py CT_PREDICT=1)
      const int ntm = (int)(mpad / TILE);
      const bool by_size = nblocks >= e->tiles_trsm_min_blocks && nblocks <= e->tiles_max_blocks &&
                           (ntm >= 32 || (long long)ntm * nblocks >= 576);
      const bool tiles = !e->naive_leaf && ntm <= 96 && (e->chol_scheme == 3 || (e->chol_scheme < 0 && by_size));
      e->cur = e->stream;
      e->tt_used = false;
      gemm_form = false;
      if (e->solve_hook) {
        if ((rc = e->solve_hook(e->dV, mpad, (int)(mpad / TILE)))) return rc;
      } else if (use_gemm_form) {
        // U = L^-T of this factor is resident (the fit's last evaluation left it): V^T = L^-1 K(X, X*) as ONE product with a
        // triangular operand (predict_form.hpp) instead of a solve
        if (!e->linv_valid) {
          LinvArgs la{};
          la.U = e->dA;
          la.ldu = e->ld;
          la.udiag = e->dUdiag;
          la.W = e->dW;
          la.ldw = e->Np;
          la.nct = nblocks;
          hipLaunchKernelGGL(linv_from_u_kernel, dim3((unsigned)((long long)nblocks * (nblocks + 1) / 2)), dim3(256), 0, e->stream, la);
          HIP_TRY(e, hipGetLastError());
          e->linv_valid = true;
        }
        GemmArgs g{};
        g.C = e->dV2;
        g.ldc = e->Np;
        g.A = e->dV;
        g.lda = mpad;
        g.B = e->dW;
        g.ldb = e->Np;
        g.mt = ntm;
        g.nt = nblocks;
        g.k = (int)e->Np;
        g.alpha = 1.0;
        g.beta = 0.0;
        g.khi_n = 1;
        if ((rc = launch_gemm(e, g, 3))) return rc;
        gemm_form = true;
      } else if ((rc = tiles ? trsm_tiles(e, e->dV, mpad, (int)(mpad / TILE), 3)
                             : trsm_cols(e, e->dV, mpad, (int)(mpad / TILE), 0, nblocks, 3, 6)))
        return rc;
      if (e->tt_used) {  // a launch that gave up waiting must not be mistaken for a prediction
        uint32_t ab = 0;
        HIP_TRY(e, hipMemcpyAsync(&ab, e->dct + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(e, hipStreamSynchronize(e->stream));
        if (ab != 0) return fail(e, GMB_EHIP, "tile triangular solve: a workgroup waited longer than its time-out for a tile (launch abandoned)");
      }
    }
    if (gemm_form) {
      hipLaunchKernelGGL(predict_rows_kernel, dim3((unsigned)mc), dim3(256), 0, e->stream, e->dV2, e->Np, e->dv, e->N, e->dkss, e->dmean, e->dvar);
      HIP_TRY(e, hipGetLastError());
    } else {
      double* pmu = e->dpart;
      double* ps = e->dpart + (int64_t)nchunk * mpad;
      hipLaunchKernelGGL(predict_partial_kernel, dim3((unsigned)(mpad / 256 + (mpad % 256 ? 1 : 0)), nchunk),
                         dim3(256), 0, e->stream, e->dV, mpad, e->dv, e->N, pmu, ps, mpad);
      hipLaunchKernelGGL(predict_final_kernel, dim3((unsigned)((mc + 255) / 256)), dim3(256), 0, e->stream,
                         pmu, ps, nchunk, mpad, e->dkss, mc, e->dmean, e->dvar);
      HIP_TRY(e, hipGetLastError());
    }
    HIP_TRY(e, hipMemcpyAsync(mean + m0, e->dmean, mc * sizeof(double), out_kind, e->stream));
    HIP_TRY(e, hipMemcpyAsync(var + m0, e->dvar, mc * sizeof(double), out_kind, e->stream));
  }
  tp.stop();
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  tm.predict_ms = tp.ms();
  ev_collect(e);
  return GMB_OK;
}

int gmb_copy_factor(const gmb_engine* ce, int64_t r0, int64_t nr, int64_t c0, int64_t nc, double* out) {
  gmb_engine* e = const_cast<gmb_engine*>(ce);
  if (!e || !out) return GMB_EINVAL;
  if (e->factored && require_full_factor(e, "gmb_copy_factor")) return GMB_EINVAL;
  if (!e->dA || e->cap_A <= 0 || r0 < 0 || c0 < 0 || nr < 0 || nc < 0 || r0 + nr > e->Nr || c0 + nc > e->Np)
    return fail(e, GMB_EINVAL, "factor window out of range");
  if (nr == 0 || nc == 0) return GMB_OK;
  std::vector<double> tmp((size_t)nr * nc);  // column-major window
  HIP_TRY(e, hipMemcpy2D(tmp.data(), nr * sizeof(double), e->dA + r0 + c0 * e->ld, e->ld * sizeof(double),
                         nr * sizeof(double), nc, hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < nr; ++i)
    for (int64_t j = 0; j < nc; ++j) out[i * nc + j] = tmp[(size_t)j * nr + i];
  return GMB_OK;
}

int gmb_copy_v(const gmb_engine* ce, double* out) {
  gmb_engine* e = const_cast<gmb_engine*>(ce);
  int rc = require_ready(e, true);
  if (rc) return rc;
  HIP_TRY(e, hipMemcpy(out, e->dv, e->N * sizeof(double), hipMemcpyDeviceToHost));
  return GMB_OK;
}

int gmb_copy_alpha(const gmb_engine* ce, double* out) {
  gmb_engine* e = const_cast<gmb_engine*>(ce);
  int rc = require_ready(e, false);
  if (rc) return rc;
  if (!out) return fail(e, GMB_EINVAL, "null output");
  if (!e->have_alpha || !e->dalpha)
    return fail(e, GMB_EINVAL, "alpha exists only after a gradient evaluation (gmb_nlml with grad)");
  HIP_TRY(e, hipMemcpy(out, e->dalpha, e->N * sizeof(double), hipMemcpyDeviceToHost));
  return GMB_OK;
}

int gmb_ls_limits(int32_t device, const double* X, int64_t N, int32_t n_cols, int64_t ldx, int32_t ard,
                  double* lower, double* upper) {
  if (!X || !lower || !upper || N < 1 || n_cols < 1 || n_cols > GMB_MAX_DIMS || ldx < n_cols)
    return GMB_EINVAL;
  int nd = gmb_device_count();
  if (nd < 0) return GMB_ENODEVICE;
  if (device < 0 || device >= nd) return GMB_EINVAL;
  if (hipSetDevice(device) != hipSuccess) return GMB_EHIP;
  const int64_t npad = round_up(N, TILE);
  const int ngroups = ard ? n_cols : 1, gw = ard ? 1 : n_cols;
  std::vector<double> soa((size_t)n_cols * npad, 0.0);
  for (int64_t i = 0; i < N; ++i)
    for (int k = 0; k < n_cols; ++k) soa[(size_t)k * npad + i] = X[i * ldx + k];
  double* dpts = nullptr;
  unsigned long long *dmn = nullptr, *dmx = nullptr;
  int rc = GMB_OK;
  std::vector<unsigned long long> hmn(ngroups, ~0ull), hmx(ngroups, 0ull);
  if (hipMalloc((void**)&dpts, soa.size() * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&dmn, ngroups * sizeof(unsigned long long)) != hipSuccess ||
      hipMalloc((void**)&dmx, ngroups * sizeof(unsigned long long)) != hipSuccess) {
    rc = GMB_ENOMEM;
  } else if (hipMemcpy(dpts, soa.data(), soa.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
             hipMemcpy(dmn, hmn.data(), ngroups * sizeof(unsigned long long), hipMemcpyHostToDevice) != hipSuccess ||
             hipMemcpy(dmx, hmx.data(), ngroups * sizeof(unsigned long long), hipMemcpyHostToDevice) != hipSuccess) {
    rc = GMB_EHIP;
  } else {
    const int tiles = (int)(npad / TILE);
    const int64_t npairs = (int64_t)tiles * (tiles + 1) / 2;
    hipLaunchKernelGGL(ls_limits_kernel, dim3((unsigned)npairs, ngroups), dim3(256), 0, 0, dpts, N, npad, gw,
                       dmn, dmx, tiles);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(hmn.data(), dmn, ngroups * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(hmx.data(), dmx, ngroups * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess)
      rc = GMB_EHIP;
  }
  if (dpts) (void)hipFree(dpts);
  if (dmn) (void)hipFree(dmn);
  if (dmx) (void)hipFree(dmx);
  if (rc) return rc;
  for (int g = 0; g < ngroups; ++g) {
    if (hmn[g] == ~0ull) {
      lower[g] = -1.0;
      upper[g] = -1.0;
    } else {
      double lo, hi;
      memcpy(&lo, &hmn[g], 8);
      memcpy(&hi, &hmx[g], 8);
      lower[g] = std::sqrt(lo);
      upper[g] = std::sqrt(hi);
    }
  }
  return GMB_OK;
}

// Register-only MFMA loop, launched back to back until `seconds` have passed (at least once): per-launch HIP
// events give the rate, block 0's s_memtime span over its own loop gives the shader clock.
static int mfma_f64_run(int32_t device, double seconds, int iters, double* mean_tf, double* min_tf, double* mhz,
                        double* cyc_per_mfma, int64_t* launches) {
  int nd = gmb_device_count();
  if (nd < 0) return GMB_ENODEVICE;
  if (device < 0 || device >= nd) return GMB_EINVAL;
  if (hipSetDevice(device) != hipSuccess) return GMB_EHIP;
  double* sink = nullptr;
  if (hipMalloc((void**)&sink, 16) != hipSuccess) return GMB_ENOMEM;
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  const int blocks = 256 * 2;  // two workgroups per compute unit, as the GEMM runs
  hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, 0, sink, 100, 1.0);  // warm-up
  const double flops = (double)blocks * 4.0 * (double)iters * 16.0 * 2.0 * 16 * 16 * 4;
  double sum_ms = 0.0, worst_ms = 0.0, sum_mhz = 0.0, sum_cyc = 0.0;
  int64_t n = 0;
  int rc = GMB_OK;
  do {
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, 0, sink, iters, 1.0);
    (void)hipEventRecord(b, 0);
    if (hipEventSynchronize(b) != hipSuccess || hipGetLastError() != hipSuccess) {
      rc = GMB_EHIP;
      break;
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    double hsink[2] = {0.0, 0.0};
    if (hipMemcpy(hsink, sink, 16, hipMemcpyDeviceToHost) != hipSuccess) {
      rc = GMB_EHIP;
      break;
    }
    sum_ms += ms;
    worst_ms = std::max(worst_ms, (double)ms);
    // block 0's loop spans (almost) the whole launch.  s_memtime ticks once per TWO shader cycles on gfx950
    // (measured: 1187 MHz of counter beside 77.8 TFLOP/s, i.e. 2375 MHz of matrix-pipe issue at 64 cycles per MFMA)
    sum_mhz += 2.0 * hsink[1] / ((double)ms * 1e3);
    sum_cyc += hsink[1] / ((double)iters * 16.0);
    ++n;
  } while (sum_ms < seconds * 1e3);
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  (void)hipFree(sink);
  if (rc) return rc;
  if (mean_tf) *mean_tf = flops * (double)n / (sum_ms * 1e-3) / 1e12;
  if (min_tf) *min_tf = flops / (worst_ms * 1e-3) / 1e12;
  if (mhz) *mhz = sum_mhz / (double)n;
  if (cyc_per_mfma) *cyc_per_mfma = sum_cyc / (double)n;
  if (launches) *launches = n;
  return GMB_OK;
}

int gmb_mfma_f64_peak(int32_t device, double* tflops, double* cycles_per_mfma) {
  if (!tflops) return GMB_EINVAL;
  return mfma_f64_run(device, 0.0, 2000, tflops, nullptr, nullptr, cycles_per_mfma, nullptr);
}

int gmb_mfma_f64_sustained(int32_t device, double seconds, double* mean_tflops, double* min_tflops, double* shader_mhz,
                           int64_t* launches) {
  if (!mean_tflops || !(seconds >= 0.0) || seconds > 60.0) return GMB_EINVAL;
  return mfma_f64_run(device, seconds, 40000, mean_tflops, min_tflops, shader_mhz, nullptr, launches);  // ~37 ms per launch
}

// ---- block-level operations (multi-GPU driver) ---------------------------------------------
// Host-only (no device): the tile list of one GEMM launch as the kernels enumerate it -- gemm_schedule's
// grid, then gemm_decode_tile for every block.  out receives (block, tm, tn) triples of the blocks that
// compute a tile; returns their number (or the negative status).  Tests check that every computed tile of
// the (triangular, strided, strip-ordered ...) launch appears exactly once.
int64_t gmb_debug_tile_list(int32_t mt, int32_t nt, int32_t bm, int32_t bn, int32_t k, int32_t tri, int32_t tri_off,
                            int32_t nblk_stride, int32_t klo_n, int32_t khi_n, int32_t order, int32_t strip,
                            int32_t* out, int64_t cap, int32_t* grid) {
  if (mt < 1 || nt < 1 || k < KT || nblk_stride < 1 || (bm != 128 && bm != 64) || (bn != 128 && bn != 64 && bn != 32))
    return GMB_EINVAL;
  GemmArgs g{};
  g.mt = mt;
  g.nt = nt;
  g.k = k;
  g.tri = tri;
  g.tri_off = tri_off;
  g.nblk_stride = nblk_stride;
  g.klo_n = klo_n;
  g.khi_n = khi_n;
  g.order = order;
  g.strip = order == 0 ? strip : 0;
  double flops = 0.0;
  const int nblocks = gemm_schedule(g, bm, bn, &flops);
  if (grid) *grid = nblocks;
  int64_t n = 0;
  for (int b = 0; b < nblocks; ++b) {
    int tm = -1, tn = -1;
    if (!gemm_decode_tile(g, bm, bn, b, tm, tn)) continue;
    if (out && n < cap) {
      out[3 * n] = b;
      out[3 * n + 1] = tm;
      out[3 * n + 2] = tn;
    }
    ++n;
  }
  return n;
}

int64_t gmb_debug_cov_grid(int32_t ti, int32_t tj, int32_t strip, int32_t tri_grid, int32_t row_first, int32_t row_stride,
                           int32_t keep_order, int32_t* out, int64_t cap, int64_t* grid) {
  if (ti < 1 || tj < 1 || strip < 1 || row_first < 0 || row_stride < 0) return GMB_EINVAL;
  const long long nb = cov_grid_blocks(ti, tj, strip, tri_grid, row_first, row_stride);
  if (grid) *grid = nb;
  int64_t n = 0;
  for (long long b = 0; b < nb; ++b) {
    int tix, lo, hi;
    if (!cov_decode_block(ti, tj, strip, tri_grid, row_first, row_stride, keep_order != 0, b, &tix, &lo, &hi)) continue;
    for (int t = lo; t < hi; ++t) {
      if (out && n < cap) {
        out[3 * n] = (int32_t)b;
        out[3 * n + 1] = tix;
        out[3 * n + 2] = t;
      }
      ++n;
    }
  }
  return n;
}

int64_t gmb_debug_chol_task(int32_t t, int32_t nct, int32_t nrt, int32_t* I, int32_t* J) {
  if (nct < 1 || nrt < nct) return GMB_EINVAL;
  const int n = ct_task_count(nct, nrt);
  int i = -1, j = -1;
  if (t >= 0 && t < n) ct_decode(t, nct, nrt, i, j);
  if (I) *I = i;
  if (J) *J = j;
  return n;
}

int64_t gmb_chol_task_trace(gmb_engine* e, int32_t enable, uint64_t* out, int64_t cap_tasks) {
  if (!e) return GMB_EINVAL;
  if (enable >= 0) e->ct_trace = enable != 0;
  // only the stamps of the LAST factorisation count: 0 when it did not run on the tile kernel or ran untraced
  const int64_t n = (e->ct_used && e->ct_traced && e->dct_trace && e->cap_ct_trace >= 4 * (int64_t)e->ct_ntasks) ? e->ct_ntasks : 0;
  if (out && n > 0) {
    HIP_TRY(e, hipSetDevice(e->device));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    HIP_TRY(e, hipMemcpy(out, e->dct_trace, (size_t)std::min<int64_t>(n, cap_tasks) * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  }
  return n;
}

int gmb_debug_chol_lose_tickets(gmb_engine* e, int32_t n) {
  if (!e || n < 0) return GMB_EINVAL;
  e->ct_lose = n;
  return GMB_OK;
}

int gmb_debug_assume_factored(gmb_engine* e) {
  if (!e) return GMB_EINVAL;
  // a door for the timing tools (tools/gpu_dist_emulate.py), never for a caller of the product: it takes an explicit opt-in
  // in the environment
  const char* door = getenv("GUMBI_HIP_DEBUG_DOORS");
  if (!door || door[0] != '1') return fail(e, GMB_EINVAL, "gmb_debug_assume_factored needs GUMBI_HIP_DEBUG_DOORS=1 in the environment");
  if (e->factor_kind == gmb_engine::FK_NONE || !e->have_theta) return fail(e, GMB_EINVAL, "no factorisation has been attempted at this theta");
  e->factored = true;
  e->factor_consumed = false;
  e->notpd = -1;
  if (!std::isfinite(e->logdet)) e->logdet = 0.0;
  if (!std::isfinite(e->vnorm2)) e->vnorm2 = 0.0;
  return GMB_OK;
}

int gmb_set_chol_scheme(gmb_engine* e, int32_t scheme) {
  if (!e) return GMB_EINVAL;
  if (!chol_scheme_valid(scheme)) return fail(e, GMB_EINVAL, "Cholesky scheme %d: valid are -1 (by size), 0, 2, 3, 4", scheme);
  const int old = e->chol_scheme;
  e->chol_scheme = scheme;
  return old + 1;  // previous scheme + 1: "by size" (-1) comes back as 0, so no valid answer collides with a (negative) status
}

int gmb_set_grad_scheme(gmb_engine* e, int32_t scheme, int32_t lag) {
  if (!e || scheme < -1 || scheme > 2) return GMB_EINVAL;
  const int old = e->grad_scheme;
  e->grad_scheme = scheme;
  if (lag >= 0) e->et_lag = lag;
  return old + 1;
}

int gmb_reserve(gmb_engine* e, int32_t gradient, int64_t M) {
  int rc = require_ready(e, false);
  if (rc) return rc;
  if (M < 0) return fail(e, GMB_EINVAL, "M must be >= 0");
  HIP_TRY(e, hipSetDevice(e->device));
  if ((rc = ensure_factor_buffer(e))) return rc;
  if (gradient) {
    if ((rc = ensure(e, &e->dW, &e->cap_W, e->Np * e->Np))) return rc;
    if ((rc = grad_workspace(e))) return rc;
  }
  if (M > 0 && (rc = predict_workspace(e, predict_tile_rows(e, M)))) return rc;
  HIP_TRY(e, hipDeviceSynchronize());
  return GMB_OK;
}

int gmb_set_predict_form(gmb_engine* e, int32_t form) {
  if (!e || form < -1 || form > 1) return GMB_EINVAL;
  const int old = e->predict_form;
  e->predict_form = form;
  return old + 1;
}

int gmb_set_eval_pairs(gmb_engine* e, int32_t mode) {
  if (!e || mode < -1 || mode > 1) return GMB_EINVAL;
  const int old = e->et_pairs;
  e->et_pairs = mode;
  return old + 1;
}

int64_t gmb_debug_eval_tasks(int32_t nct, int32_t nrt, int32_t with_chol, int32_t lag, uint32_t* out, int64_t cap) {
  if (nct < 1 || nrt < nct || nrt > 0x3fff || lag < 0 || with_chol < 0 || with_chol > 3) return GMB_EINVAL;
  std::vector<uint32_t> list;
  et_build_tasks(nct, nrt, (with_chol & 1) != 0, lag, list, (with_chol & 2) != 0);
  if (out)
    for (int64_t i = 0; i < (int64_t)list.size() && i < cap; ++i) out[i] = list[(size_t)i];
  return (int64_t)list.size();
}

int gmb_blk_covariance(gmb_engine* e, double* out, int64_t ldo) {
  int rc = require_ready(e, false);
  if (rc) return rc;
  if (!out || ldo < e->Nr) return fail(e, GMB_EINVAL, "gmb_blk_covariance: out is null or ldo < %lld", (long long)e->Nr);
  // the interior tiles go out as 16-byte stores (covariance.hpp: cov_interior_tile): two rows per lane
  if ((((uintptr_t)out) | (uintptr_t)(ldo * (int64_t)sizeof(double))) & 15u)
    return fail(e, GMB_EINVAL, "gmb_blk_covariance: out must be 16-byte aligned and ldo even (16-byte stores)");
  HIP_TRY(e, hipSetDevice(e->device));
  if ((rc = build_sigma(e, out, ldo))) return rc;
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return GMB_OK;
}

int gmb_blk_potrf(gmb_engine* e, double* Akk, int64_t lda, int32_t nvalid, double* dinv16,
                  double* logdet_accum, int32_t* info) {
  if (!e || !Akk || nvalid < 1 || nvalid > TILE || lda < TILE) return fail(e, GMB_EINVAL, "bad potrf block");
  HIP_TRY(e, hipSetDevice(e->device));
  LeafArgs a{};
  a.A = Akk;
  a.lda = lda;
  a.nvalid = nvalid;
  a.dinv16 = dinv16;
  a.logdet = logdet_accum;
  a.info = info ? info : e->dinfo;
  a.row0 = 0;
#ifdef GMB_TUNING
  a.dbg = getenv("GMB_LEAF_DBG") ? logdet_accum + 1 : nullptr;  // stamps after the log-det slot (tools/gpu_leaf_timing.py)
#else
  a.dbg = nullptr;
#endif
  return launch_leaf(e, a);
}

int gmb_blk_invert(gmb_engine* e, const double* Lkk, int64_t lda, int32_t nvalid, const double* dinv16,
                   double* invLkk) {
  if (!e || !Lkk || !dinv16 || !invLkk || nvalid < 1 || nvalid > TILE || lda < TILE)
    return fail(e, GMB_EINVAL, "bad invert block");
  HIP_TRY(e, hipSetDevice(e->device));
  InvArgs ia{};
  ia.L = Lkk;
  ia.lda = lda;
  ia.blk_stride = 0;
  ia.dinv16 = dinv16;
  ia.invL = invLkk;
  ia.n = nvalid;
  hipLaunchKernelGGL(leaf_invert_kernel, dim3(1), dim3(256), 0, e->stream, ia);
  HIP_TRY(e, hipGetLastError());
  return GMB_OK;
}

int gmb_blk_trsm(gmb_engine* e, double* B, int64_t ldb, int64_t nrows, const double* Lkk, int64_t ldl,
                 const double* dinv16, int32_t nvalid) {
  if (!e || !B || !Lkk || !dinv16 || nrows < 0 || nrows % 16 || ldb < nrows || ldl < TILE || nvalid < 1 ||
      nvalid > TILE)
    return fail(e, GMB_EINVAL, "bad trsm arguments (rows must be a multiple of 16)");
  HIP_TRY(e, hipSetDevice(e->device));
  return launch_trsm_strip(e, B, ldb, nrows, Lkk, ldl, dinv16, nvalid, 5);
}

int gmb_blk_gemm_nt(gmb_engine* e, double* C, int64_t ldc, const double* A, int64_t lda, const double* B,
                    int64_t ldb, int64_t m, int64_t n, int64_t k, double alpha, double beta, int32_t tri,
                    int64_t tri_shift) {
  if (!e || !C || !A || !B) return fail(e, GMB_EINVAL, "null gemm operand");
  if (m % TILE || n % TILE || k % KT || m < 0 || n < 0 || k < 0)
    return fail(e, GMB_EINVAL, "gemm sizes must be multiples of 128 (m, n) and 16 (k)");
  HIP_TRY(e, hipSetDevice(e->device));
  GemmArgs g{};
  g.C = C;
  g.ldc = ldc;
  g.A = A;
  g.lda = lda;
  g.B = B;
  g.ldb = ldb;
  g.mt = (int)(m / TILE);
  g.nt = (int)(n / TILE);
  g.k = (int)k;
  g.alpha = alpha;
  g.beta = beta;
  g.tri = tri;
  g.tri_off = (int)(tri_shift * TILE);
  return launch_gemm(e, g, 0);
}

}  // extern "C"

#include "dist_driver.hpp"
#include "dist_capacity.hpp"

extern "C" {

// 0 = replicated factor (default: every rank ends with the complete factor), 1 = capacity (dist_capacity.hpp: owned block rows
// + panel buffers only).  Takes effect with the next gmb_dist_factorize; returns the previous mode or a negative status.
int gmb_dist_set_mode(gmb_engine* e, int32_t mode) {
  if (!e || mode < 0 || mode > 1) return GMB_EINVAL;
  const int old = e->dist_mode;
  if (old != mode) {
    e->factored = false;
    e->factor_kind = gmb_engine::FK_NONE;
  }
  e->dist_mode = mode;
  if (mode == 1) {
    // the point of the mode is that no rank keeps an Nr x Np buffer: one left behind by an earlier single-engine or replicated
    // factorisation goes now (its factorisation was just invalidated, or is of no use to the capacity passes)
    if (e->dA && e->cap_A > 0) {
      if (hipSetDevice(e->device) == hipSuccess) (void)hipStreamSynchronize(e->stream);
      release(e, e->dA);
      e->factored = false;
      e->factor_kind = gmb_engine::FK_NONE;
    }
    e->dA = nullptr;
    e->cap_A = 0;
  }
  return old;
}

// device bytes this engine holds through its own allocations right now (peak != 0: the largest value so far)
int64_t gmb_resident_bytes(const gmb_engine* e, int32_t peak) {
  if (!e) return GMB_EINVAL;
  return peak ? e->bytes_peak : e->bytes_resident;
}

int gmb_dist_factorize(gmb_engine* e, const gmb_comm* comm, int32_t panel_blocks) {
  if (!e) return GMB_EINVAL;
  return e->dist_mode == 1 ? cap_factorize(e, comm, panel_blocks) : dist_factorize(e, comm, panel_blocks);
}

int gmb_dist_nlml(gmb_engine* e, const gmb_comm* comm, double* nlml, double* grad) {
  if (!e) return GMB_EINVAL;
  return e->dist_mode == 1 ? cap_nlml(e, comm, nlml, grad) : dist_nlml(e, comm, nlml, grad);
}

int gmb_dist_predict(gmb_engine* e, const gmb_comm* comm, const double* Xs, int64_t M, int64_t ldxs, int32_t with_noise,
                     double* mean, double* var, int32_t memspace) {
  if (!e) return GMB_EINVAL;
  return e->dist_mode == 1 ? cap_predict(e, comm, Xs, M, ldxs, with_noise, mean, var, memspace)
                           : dist_predict(e, comm, Xs, M, ldxs, with_noise, mean, var, memspace);
}

}  // extern "C"
