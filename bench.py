#!/usr/bin/env python
"""Headline benchmark: GP fit + predict on synthetic N x d fp64 tables (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c2|c1|c5] [--map-evals E]

One GPU (default; BASELINE.json configs[2] = C3, the largest single-GPU configuration: N = 50k, d = 8,
Matern-5/2 ARD, M = 10^4 grid).  A *step* is one pass of the hot path the reference reaches through
``gp.fit()`` + ``gp.predict_grid()``, inputs already resident in HBM:

    find_MAP  -- L-BFGS-B on -(log-lik + log-priors); every objective / gradient evaluation = covariance
                 build -> Cholesky -> v = L^-1 y, log-det -> L^-T, Sigma^-1, fused trace reductions
    refactor at the MAP, predict mean / variance on the M-point grid                (all in libgumbi_hip.so)

``value`` = algorithmic GFLOP/s of the whole step; ``ms_per_step`` = fit + predict wall time.  The same JSON
line carries: ``roofline`` (the Cholesky's trailing-update SYRK/GEMM launches alone -- the kernel the north
star's MFMA target is stated on -- from per-launch HIP events of one extra step, plus the same figure over
every GEMM launch), ``kbuild`` (HBM GB/s of the covariance build), ``phases``, ``end_to_end`` (the user-level
``DataSet -> GP.fit() -> prepare_grid() -> predict_grid()`` wall time, host transfers included),
``c5_single_gpu`` (the N = 100k problem of the multi-GPU runs on this one GPU: their strong-scaling base) and
``cpu_baseline`` (the oracle on a bounded sample on the host cores).

Several GPUs (``--gpus N``, launched by torch.distributed.run, one rank per GPU): ONE GP -- BASELINE.json
configs[4] = C5, N = 100k, d = 8, RBF-ARD -- factored block-cyclically over all ranks by the native driver
(gumbi_amd/csrc/dist_driver.hpp; all-gathers on RCCL over xGMI), hyper-parameters fixed.  A step = one MAP
objective + gradient evaluation (factorise + row-partitioned gradient) + re-factorisation (``fit`` at fixed
theta) + prediction of the 10^4 grid sharded over the ranks; the three parts are reported separately under
``phases``; ``scaling`` is "strong" (the same problem at every N; ``python bench.py --config c5`` is its N = 1
point).  The barrier / max-over-ranks timing contract is the same.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

FP64_MFMA_PEAK_TFLOPS = 78.6  # AMD MI355X datasheet (vector = matrix FP64); MI355X_MICROARCH.md lists none
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

CONFIGS = {
    # name: (N, d, kernel, M-grid resolution)
    "c1": dict(N=392, d=1, kernel="ExpQuad", res=100, label="C1-like synthetic N=392 d=1 RBF"),
    "c2": dict(N=10_000, d=4, kernel="ExpQuad", res=100, label="synthetic N=10k d=4 RBF-ARD fp64, M=10^4 grid"),
    "c3": dict(N=50_000, d=8, kernel="Matern52", res=100, label="synthetic N=50k d=8 Matern-5/2 ARD fp64, M=10^4 grid"),
    "c5": dict(N=100_000, d=8, kernel="ExpQuad", res=100,
               label="synthetic N=100k d=8 RBF-ARD fp64, fixed theta, M=10^4 grid, ONE GP block-cyclic over the GPUs"),
}
# test hook: shrink the one-GP problem (the contract tests run it on a shared GPU in seconds)
if os.environ.get("GUMBI_BENCH_DIST_N"):
    CONFIGS["c5"]["N"] = int(os.environ["GUMBI_BENCH_DIST_N"])
    CONFIGS["c5"]["label"] = CONFIGS["c5"]["label"].replace("N=100k", f"N={CONFIGS['c5']['N']}")


def synthetic_table(N, d, seed=2021, sigma=0.2):
    """SURVEY.md section 8d generator (same draws as oracle/gp_oracle.py:synthetic_table)."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, d))
    ls = np.geomspace(0.7, 2.0, d) if d > 1 else np.array([1.0])
    f = np.sum(np.sin(X / ls), axis=1) / np.sqrt(d)
    y = f + sigma * rng.standard_normal(N)
    y = (y - y.mean()) / y.std(ddof=1)
    return X, y, ls


def synthetic_grid(d, res=100, lim=2.4):
    g = np.linspace(-lim, lim, res)
    if d == 1:
        return g[:, None].copy()
    G0, G1 = np.meshgrid(g, g, indexing="ij")
    Xs = np.zeros((res * res, d))
    Xs[:, 0], Xs[:, 1] = G0.ravel(), G1.ravel()
    return Xs


def make_dataset(cfg):
    import pandas as pd

    import gumbi_amd as gmb

    X, y, _ = synthetic_table(cfg["N"], cfg["d"])
    cols = [f"x{k}" for k in range(cfg["d"])]
    df = pd.DataFrame(X, columns=cols)
    df["y"] = y
    return gmb.DataSet(df, outputs=["y"]), cols


def build_gp(cfg, device):
    """Front-end objects exactly as a Gumbi user builds them (DataSet -> GP -> build_model)."""
    import gumbi_amd as gmb

    ds, cols = make_dataset(cfg)
    gp = gmb.GP(ds, outputs=["y"], device=device)
    gp.specify_model(continuous_dims=cols)
    gp.build_model(continuous_kernel=cfg["kernel"])
    return gp


def step_flops(N, M, n_eval, n_refactor=1):
    """Algorithmic flops of one step: per evaluation N^3/3 (Cholesky) +
    2N^3/3 (inverse + Sigma^-1); N^3/3 per re-factorisation at the MAP (none when the optimiser's last
    evaluation was at the MAP: its factorisation is still resident); predict N^2 M + 4 N M."""
    n3 = float(N) ** 3
    return n_eval * n3 + n_refactor * n3 / 3.0 + float(N) ** 2 * M + 4.0 * N * M


def pmc_traffic(config, kernel_filter):
    """HBM bytes per launch of the named kernels from the committed rocprofv3 PMC summary of this same bench
    command (profiles/*_pmc_bench_<config>_summary.csv, made by tools/gpu_pmc_bench.sh: separate --pmc passes
    for FETCH_SIZE and WRITE_SIZE; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950,
    WRITE_SIZE as counted).  None when no summary is committed."""
    import csv
    import glob

    files = sorted(glob.glob(os.path.join(str(ROOT), "profiles", f"*_pmc_bench_{config}_summary.csv")))
    if not files:
        return None
    launches, gbytes, flops = 0, 0.0, 0.0
    with open(files[-1]) as fh:
        for row in csv.DictReader(fh):
            if kernel_filter in row["Kernel"]:
                launches += int(row["Launches"])
                gbytes += float(row["FetchGB(x2 corrected)"]) + float(row["WriteGB(raw)"])
                # flops of these launches from SQ_INSTS_VALU_MFMA_MOPS_F64 (rate x duration of the same pass)
                flops += float(row["MFMA_F64_TFLOPs"]) * 1e12 * float(row["TotalMs(pass1)"]) * 1e-3
    if launches == 0:
        return None
    return {"bytes_per_launch": gbytes * 1e9 / launches, "source": os.path.relpath(files[-1], str(ROOT)), "launches": launches,
            "flops_per_launch": flops / launches}


def cpu_baseline(cfg, target_seconds=20.0):
    """Oracle (numpy + LAPACK through scipy) timed on a bounded sample of the same workload:
    one MAP objective+gradient evaluation and one grid prediction at a reduced N chosen (from a
    short probe, cost ~ N^3) to take about ``target_seconds`` on this host."""
    from oracle import gp_oracle as O

    d = cfg["d"]
    Xs = O.synthetic_grid(d, cfg["res"])
    spec = O.make_spec(d, range(d), kind=cfg["kernel"])

    def run(n):
        X, y, ls = O.synthetic_table(n, d)
        theta = O.pack_theta(spec, ls, 1.0, 0.2)
        t0 = time.perf_counter()
        O.nlml_and_grad(spec, theta, X, y, dist_mode="gemm")
        O.predict(spec, theta, X, y, Xs, with_noise=True)
        return time.perf_counter() - t0

    # grow the sample (cost ~ N^3, but BLAS efficiency also grows with N, so re-scale from each
    # measurement) until one evaluation takes >= 10 s or the full N is reached
    target_seconds = float(os.environ.get("GUMBI_BENCH_CPU_SECONDS", target_seconds))  # tests shorten it
    Ns = min(cfg["N"], 1500)
    dt = run(Ns)
    for _ in range(3):
        if dt >= 0.5 * target_seconds or Ns >= cfg["N"]:
            break
        nxt = int(Ns * (target_seconds / max(dt, 1e-3)) ** (1.0 / 3.0))
        nxt = min(cfg["N"], max(Ns + 128, nxt // 128 * 128))
        Ns = nxt
        dt = run(Ns)
    flops = step_flops(Ns, len(Xs), 1)
    cores = os.cpu_count() or 1
    try:
        from threadpoolctl import threadpool_info

        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        pass
    return {
        "value": round(flops / dt / 1e9, 2),
        "unit": "GFLOP/s",
        "cores": int(cores),
        "kind": "port",
        "sample": f"1 MAP objective+gradient evaluation + predict(M={len(Xs)}) at N={Ns}, d={d}, {cfg['kernel']} "
                  f"(numpy/LAPACK oracle, {dt:.1f} s)",
        "seconds": round(dt, 2),
    }


def roofline_block(tm, config):
    """``roofline`` from the engine's per-launch HIP events (gmb_timings, cumulative since profiling was
    switched on): the trailing-update launches of the Cholesky alone, and every GEMM launch beside it."""
    chol_tf = tm["total_chol_gemm_flops"] / max(tm["total_chol_gemm_ms"], 1e-9) / 1e9
    all_tf = tm["total_gemm_flops"] / max(tm["total_gemm_ms"], 1e-9) / 1e9
    n_chol = max(int(tm["total_chol_gemm_launches"]), 1)
    out = {
        "bound": "mfma",
        "kernel": "gemm_f64_kernel, the Cholesky's bulk trailing-update launches U1 / U2 (v_mfma_f64_16x16x4_f64 SYRK/GEMM)",
        "achieved": round(chol_tf, 3),
        "peak": FP64_MFMA_PEAK_TFLOPS,
        "unit": "TFLOP/s",
        "frac": round(chol_tf / FP64_MFMA_PEAK_TFLOPS, 4),
        "traffic": None,
        "measured_over": "one extra step of the same workload after the timed region (per-launch HIP events on the "
                         "launch's own stream)",
        "launches": int(tm["total_chol_gemm_launches"]),
        "avg_launch_ms": round(tm["total_chol_gemm_ms"] / n_chol, 5),
        "flops_per_launch": round(tm["total_chol_gemm_flops"] / n_chol, 1),
        # launches of the look-ahead schedule's two streams overlap: the same flops over the WALL time with at
        # least one of these launches in flight (union of the launch intervals)
        "achieved_over_wall_time": round(tm["total_chol_gemm_flops"] / max(tm.get("total_chol_gemm_wall_ms", 0.0), 1e-9) / 1e9, 3),
        # the small products INSIDE the panels (the latency-bound chain that runs beside U2): share of the Cholesky's flops
        "in_panel_products": {"flops_share_of_cholesky": round(tm.get("total_chol_panel_gemm_flops", 0.0) / max(
            tm.get("total_chol_panel_gemm_flops", 0.0) + tm["total_chol_gemm_flops"], 1.0), 4),
            "sum_of_launch_ms": round(tm.get("total_chol_panel_gemm_ms", 0.0), 3)},
        "all_gemm_launches": {  # trailing updates + in-panel products + triangular solves + inverse + Sigma^-1 + predict
            "achieved": round(all_tf, 3),
            "frac": round(all_tf / FP64_MFMA_PEAK_TFLOPS, 4),
            "launches": int(tm["total_gemm_launches"]),
            "avg_launch_ms": round(tm["total_gemm_ms"] / max(tm["total_gemm_launches"], 1), 5),
            "flops_per_launch": round(tm["total_gemm_flops"] / max(tm["total_gemm_launches"], 1), 1),
            "achieved_over_wall_time": round(tm["total_gemm_flops"] / max(tm.get("total_gemm_wall_ms", 0.0), 1e-9) / 1e9, 3),
        },
    }
    if tm.get("masked_gemm_flops", 0.0) > 0.0:
        # trailing updates of the masked look-ahead schedule run on masked_cus of the chip's compute units BY
        # DESIGN (the rest serves the concurrent panel chain): their share and rate are reported separately
        ncu = 256
        mtf = tm["masked_gemm_flops"] / max(tm["masked_gemm_ms"], 1e-9) / 1e9
        out["cu_masked_launches"] = {
            "compute_units": int(tm["masked_cus"]), "of": ncu,
            "flops_share": round(tm["masked_gemm_flops"] / tm["total_gemm_flops"], 4),
            "achieved": round(mtf, 3),
            "frac_of_their_share_of_peak": round(mtf / (FP64_MFMA_PEAK_TFLOPS * tm["masked_cus"] / ncu), 4),
        }
    pt = pmc_traffic(config, "gemm_f64_kernel<2, 2, 4, 4")  # the 128 x 128 instantiation the bulk updates run
    if pt is not None and pt["flops_per_launch"] > 0:
        # the PMC passes count every launch of the 128x128 instantiation (bulk updates, solves, inverse, Sigma^-1,
        # predict: other launch sizes than the ones `achieved` is quoted on), so the bytes are scaled by flops
        bytes_per_flop = pt["bytes_per_launch"] / pt["flops_per_launch"]
        out["traffic"] = round(bytes_per_flop * tm["total_chol_gemm_flops"] / n_chol, 1)
        out["traffic_unit"] = ("HBM-side bytes per bulk-update launch: (2*FETCH_SIZE + WRITE_SIZE) per flop of the 128x128 "
                               "gemm_f64_kernel instantiation over one bench step (rocprofv3 PMC passes), times this launch's flops")
        out["traffic_flop_per_byte"] = round(1.0 / bytes_per_flop, 2)
        out["traffic_source"] = pt["source"]
        out["traffic_source_per_launch"] = {"launches": pt["launches"], "bytes": round(pt["bytes_per_launch"], 1),
                                            "flops": round(pt["flops_per_launch"], 1)}
    return out


def kbuild_block(tm):
    kb_gbs = tm["total_kbuild_bytes"] / max(tm["total_kbuild_ms"], 1e-9) / 1e6
    return {"bound": "hbm", "kernel": "cov_tile_kernel (lower-triangle covariance build)", "achieved": round(kb_gbs, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(kb_gbs / HBM_PEAK_GBS, 4),
            "launches": int(tm["total_kbuild_launches"]),
            "bytes_per_launch": round(tm["total_kbuild_bytes"] / max(tm["total_kbuild_launches"], 1), 1)}


class Clock:
    """barrier + device synchronisation on both sides of a timed region; max over ranks."""

    def __init__(self, dist, dev):
        self.dist, self.dev = dist, dev

    def sync(self):
        import torch

        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, values):
        import torch

        if self.dist is None:
            return [float(v) for v in values]
        on_dev = self.dist.get_backend() == "nccl"
        t = torch.tensor(list(values), dtype=torch.float64, device=self.dev if on_dev else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(v) for v in t.cpu()]


# -----------------------------------------------------------------------------------------------------
# workload A: MAP fit + grid prediction of one table on ONE GPU (C1 / C2 / C3)
# -----------------------------------------------------------------------------------------------------
def map_fit_workload(cfg, config_name, local_rank, steps, warmup, map_evals, clock):
    import torch

    t_build = time.perf_counter()
    gp = build_gp(cfg, device=local_rank)
    t_build = time.perf_counter() - t_build
    eng = gp.engine
    Xs = synthetic_grid(cfg["d"], cfg["res"])
    M = len(Xs)
    dev = torch.device("cuda", local_rank)
    xs_dev = torch.tensor(Xs, dtype=torch.float64, device=dev)  # X* resident in HBM
    mean_dev = torch.empty(M, dtype=torch.float64, device=dev)
    var_dev = torch.empty(M, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    maxeval = map_evals if map_evals > 0 else 200

    def one_step():
        gp.find_MAP(maxeval=maxeval)
        eng.predict_device(xs_dev.data_ptr(), M, cfg["d"], mean_dev.data_ptr(), var_dev.data_ptr(), True)
        return gp.n_eval, gp.n_refactor

    for _ in range(warmup):
        one_step()
    clock.sync()
    t0 = time.perf_counter()
    counts = [one_step() for _ in range(steps)]
    n_evals = [c[0] for c in counts]
    clock.sync()
    elapsed = time.perf_counter() - t0
    # Roofline pass: ONE more step of the same workload with a HIP event pair around every GEMM launch (on
    # the stream it is launched on).  Kept out of the timed region: event pairs cost host time per launch.
    eng.set_profiling(True)  # resets the totals
    one_step()
    torch.cuda.synchronize()
    tm = eng.timings()
    eng.set_profiling(False)

    # Phases of one evaluation timed separately, wall clock, unprofiled, at the
    # fitted hyper-parameters; outside the timed region.
    def wall_ms(fn, reps=3):
        best = float("inf")
        for _ in range(reps):
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t_)
        return round(1e3 * best, 3)

    def fact_then_grad():
        eng.factorize()
        eng.nlml(grad=True)

    phases = {
        "specify_plus_build_model_s": round(t_build, 3),  # DataSet -> specify_model -> build_model (ls priors, H2D copy)
        "factorize_ms": wall_ms(eng.factorize),          # K-build + Cholesky + L^-1 y + log-det
        "factorize_plus_gradient_ms": wall_ms(fact_then_grad),  # one MAP objective evaluation
    }
    eng.factorize()
    phases["predict_ms"] = wall_ms(lambda: eng.predict_device(xs_dev.data_ptr(), M, cfg["d"], mean_dev.data_ptr(),
                                                                var_dev.data_ptr(), True))
    phases["ls_limits_ms"] = wall_ms(lambda: gp._prepare_lengthscales(gp.model.X, ARD=True))
    N = cfg["N"]
    phases["rates_tflops"] = {
        "factorize": round(N**3 / 3.0 / phases["factorize_ms"] / 1e9, 2),
        "map_evaluation": round(float(N) ** 3 / phases["factorize_plus_gradient_ms"] / 1e9, 2),
        "predict": round(float(N) ** 2 * M / phases["predict_ms"] / 1e9, 2),
    }
    phases["profiled_last_evaluation_ms"] = {k: round(tm[k], 3) for k in
                                             ("kbuild_ms", "chol_ms", "chol_leaf_ms", "chol_trsm_ms", "chol_gemm_ms",
                                              "grad_ms", "grad_gemm_ms", "predict_ms", "predict_gemm_ms")}
    mean_host = mean_dev.cpu().numpy()
    var_host = var_dev.cpu().numpy()
    finite = bool(np.all(np.isfinite(mean_host)) and np.all(var_host > 0))
    # the engine's streams go away before any other section creates its own: more than four live HIP
    # streams per process slow every kernel down on this stack
    eng.close()
    gp.engine = None
    return dict(elapsed=elapsed, n_evals=n_evals, n_refactor=[c[1] for c in counts], M=M, tm=tm, phases=phases, finite=finite,
                flops=sum(step_flops(N, M, n, r) for n, r in counts))


def end_to_end_fit(cfg, local_rank, map_evals):
    """The user-level call sequence of the reference (DataSet -> GP.fit -> prepare_grid -> predict_grid), wall
    clock, once: host-side plumbing, lengthscale priors, host -> device copies and result wrapping included."""
    import gumbi_amd as gmb

    t0 = time.perf_counter()
    ds, cols = make_dataset(cfg)
    t1 = time.perf_counter()
    gp = gmb.GP(ds, outputs=["y"], device=local_rank)
    gp.fit(continuous_dims=cols, continuous_kernel=cfg["kernel"], MAP_kwargs={"maxeval": map_evals if map_evals > 0 else 200})
    t2 = time.perf_counter()
    if cfg["d"] > 2:
        gp.prepare_grid(at=gp.parray(**{c: 0.0 for c in cols[2:]}, stdzd=True), resolution=cfg["res"])
    else:
        gp.prepare_grid(resolution=cfg["res"])
    pred = gp.predict_grid()
    t3 = time.perf_counter()
    ok = bool(np.all(np.isfinite(np.asarray(pred.μ))))
    gp.engine.close()
    gp.engine = None
    return {"dataset_s": round(t1 - t0, 3), "fit_s": round(t2 - t1, 3), "prepare_and_predict_grid_s": round(t3 - t2, 3),
            "total_s": round(t3 - t0, 3), "map_evals": int(gp.n_eval), "results_finite": ok,
            "note": "gp.fit() = specify_model + build_model + find_MAP, host <-> device transfers included"}


# -----------------------------------------------------------------------------------------------------
# workload B: ONE GP at fixed hyper-parameters over `world` GPUs (C5)
# -----------------------------------------------------------------------------------------------------
def one_gp_workload(cfg, world, local_rank, dist, steps, warmup, clock):
    import torch

    from gumbi_amd import engine as E

    N, d = cfg["N"], cfg["d"]
    X, y, ls = synthetic_table(N, d)
    Xs = synthetic_grid(d, cfg["res"])
    M = len(Xs)
    theta = np.concatenate([ls, [1.0, 0.2]])
    spec = E.KernelSpec(D=d, idx_cont=list(range(d)), kind=cfg["kernel"])
    if world == 1:
        eng = E.Engine(local_rank)
        transport = "single engine"
    else:
        from gumbi_amd.distributed import DistributedEngine

        eng = DistributedEngine(local_rank)
        transport = eng.comm.kind
    eng.set_data(X, y)
    eng.set_kernel(spec)
    eng.set_theta(theta)
    part = {"map_eval": 0.0, "fit_fixed_theta": 0.0, "predict": 0.0}
    last = {}

    def one_step(record=False):
        t0 = time.perf_counter()
        eng.factorize()
        val, grad = eng.nlml(grad=True)      # one MAP objective + gradient evaluation
        t1 = time.perf_counter()
        eng.factorize()                      # fit at fixed theta: K-build + Cholesky + v + log-det
        nl = eng.nlml()
        t2 = time.perf_counter()
        mu, var = eng.predict(Xs)            # 10^4-point grid (sharded over the ranks)
        t3 = time.perf_counter()
        if record:
            part["map_eval"] += t1 - t0
            part["fit_fixed_theta"] += t2 - t1
            part["predict"] += t3 - t2
        last.update(val=val, grad=grad, nl=nl, mu=mu, var=var)

    for _ in range(warmup):
        one_step()
    clock.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step(record=True)
    clock.sync()
    elapsed = time.perf_counter() - t0
    eng.set_profiling(True)
    one_step()
    torch.cuda.synchronize()
    tm = eng.timings()
    eng.set_profiling(False)
    eng.close()
    per = clock.max_over_ranks([part["map_eval"] / steps, part["fit_fixed_theta"] / steps, part["predict"] / steps])
    n3 = float(N) ** 3
    phases = {
        "map_eval_s": round(per[0], 4), "fit_fixed_theta_s": round(per[1], 4), "predict_s": round(per[2], 4),
        "rates_tflops_whole_job": {"map_eval": round(n3 / per[0] / 1e12, 2), "fit_fixed_theta": round(n3 / 3.0 / per[1] / 1e12, 2),
                                   "predict": round(float(N) ** 2 * M / per[2] / 1e12, 2)},
        "profiled_last_step_ms_rank0": {k: round(tm[k], 3) for k in ("kbuild_ms", "chol_ms", "chol_gemm_ms", "grad_ms",
                                                                      "grad_gemm_ms", "predict_ms")},
        "nlml": float(last["nl"]),
    }
    finite = bool(np.all(np.isfinite(last["mu"])) and np.all(last["var"] > 0) and np.all(np.isfinite(last["grad"])))
    return dict(elapsed=elapsed, M=M, tm=tm, phases=phases, finite=finite, transport=transport,
                flops=steps * step_flops(N, M, 1))


def run_with_deadline(fn, seconds):
    """fn() on a helper thread; (result, finished).  A collective that never completes (a peer died, a
    link is down) must not hang the run for ever: the caller prints what it has and leaves with os._exit."""
    import threading

    box = {}

    def target():
        try:
            box["value"] = fn()
        except Exception as err:
            import traceback

            box["value"] = {"error": f"{type(err).__name__}: {err}"[:400], "traceback": traceback.format_exc()[-1500:]}

    th = threading.Thread(target=target, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        return {"error": f"no result after {seconds:.0f} s (collective did not complete)"}, False
    return box["value"], True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="default: c3 on one GPU, c5 (ONE GP over all ranks) on several")
    ap.add_argument("--map-evals", type=int, default=0,
                    help="cap on L-BFGS objective evaluations per fit (0 = run to convergence, cap 200)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    config_name = args.config or ("c3" if world == 1 else "c5")
    cfg = CONFIGS[config_name]
    if world > 1 and config_name != "c5":
        raise SystemExit("several GPUs run the one-GP workload: --config c5")

    dist = None
    import torch

    if os.environ.get("GUMBI_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0  # test hook: several ranks share GPU 0 (needs GUMBI_BENCH_BACKEND=gloo)
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        backend = os.environ.get("GUMBI_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from gumbi_amd import engine

    if engine.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: libgumbi_hip has no CPU fallback")
    dev = torch.device("cuda", local_rank)
    clock = Clock(dist, dev)
    healthy = True

    if config_name == "c5":
        def section():
            torch.cuda.set_device(local_rank)  # the current device is per-thread state
            return one_gp_workload(cfg, world, local_rank, dist, args.steps, args.warmup, clock)

        res, healthy = run_with_deadline(section, float(os.environ.get("GUMBI_BENCH_DEADLINE", "1500")))
    else:
        res = map_fit_workload(cfg, config_name, local_rank, args.steps, args.warmup, args.map_evals, clock)

    out = None
    if "error" in res:
        if rank == 0:
            out = {"metric": "fit+predict achieved GFLOP/s (fp64 exact GP)", "value": None, "unit": "GFLOP/s", "n_gpus": world,
                   "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                   "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                   "config": {"workload": cfg["label"]}, "error": res}
    else:
        (elapsed,) = clock.max_over_ranks([res["elapsed"]]) if healthy else (res["elapsed"],)
        if rank == 0:
            tm = res["tm"]
            out = {
                "metric": "fit+predict achieved GFLOP/s (fp64 exact GP: MAP fit + grid prediction)" if config_name != "c5"
                          else "fit+predict achieved GFLOP/s (fp64 exact GP: one MAP evaluation + fit at fixed theta + grid prediction)",
                "value": round(res["flops"] / elapsed / 1e9, 2),
                "unit": "GFLOP/s",
                "n_gpus": world,
                "steps": args.steps,
                "warmup": args.warmup,
                "ms_per_step": round(1e3 * elapsed / args.steps, 3),
                "higher_is_better": True,
                "scaling": "strong" if config_name == "c5" else "weak",
                "vs_baseline": None,
                "dtype": "f64",
                "data": "synthetic",
                "config": {
                    "workload": cfg["label"],
                    "N": cfg["N"], "d": cfg["d"], "kernel": cfg["kernel"], "M": res["M"],
                    "parallelism": "1 GPU" if world == 1 else
                                   f"ONE GP, 128-row blocks dealt block-cyclically over {world} GPUs; panel all-gathers: {res['transport']}",
                },
                "fit_predict_seconds": round(elapsed / args.steps, 4),
                "roofline": roofline_block(tm, config_name),
                "kbuild": kbuild_block(tm),
                "results_finite": res["finite"],
                "phases": res["phases"],
            }
            if "n_evals" in res:
                out["config"]["map_evals_per_step"] = res["n_evals"]
                out["config"]["refactorizations_at_the_map_per_step"] = res["n_refactor"]
            try:
                tf, cyc = engine.mfma_f64_peak(local_rank)
                out["roofline"]["mfma_only_microbench_tflops"] = round(tf, 2)  # sustained ceiling under DVFS
            except Exception:
                pass

    # side sections of the one-GPU run (outside the timed steps)
    if world == 1 and out is not None and "error" not in out and config_name != "c5":
        if os.environ.get("GUMBI_BENCH_NO_E2E") != "1":
            try:
                out["end_to_end"] = end_to_end_fit(cfg, local_rank, args.map_evals)
            except Exception as err:
                out["end_to_end"] = {"error": f"{type(err).__name__}: {err}"[:300]}
        if os.environ.get("GUMBI_BENCH_NO_DIST") != "1":
            def c5_side():
                torch.cuda.set_device(local_rank)
                r = one_gp_workload(CONFIGS["c5"], 1, local_rank, None, 2, 1, clock)
                return {"workload": CONFIGS["c5"]["label"], "note": "strong-scaling base of the multi-GPU runs (python bench.py --config c5)",
                        "value": round(r["flops"] / r["elapsed"] / 1e9, 2), "unit": "GFLOP/s", "steps": 2, "warmup": 1,
                        "ms_per_step": round(1e3 * r["elapsed"] / 2, 3), "phases": r["phases"], "results_finite": r["finite"],
                        "trailing_update_tflops": roofline_block(r["tm"], "c5")["achieved"]}

            out["c5_single_gpu"], _ = run_with_deadline(c5_side, 900.0)
        if not args.no_cpu_baseline and os.environ.get("GUMBI_BENCH_NO_CPU") != "1":
            out["cpu_baseline"] = cpu_baseline(cfg)
    if rank == 0 and out is not None:
        print(json.dumps(out, ensure_ascii=False), flush=True)
    if dist is not None:
        def leave():
            torch.cuda.set_device(local_rank)
            dist.barrier()
            dist.destroy_process_group()

        if healthy:
            _, healthy = run_with_deadline(leave, 180.0)
        if not healthy:
            sys.stderr.write(f"[bench rank {rank}] leaving without the final barrier\n")
            sys.stderr.flush()
            os._exit(0 if out is None or "error" not in out else 1)


if __name__ == "__main__":
    main()
