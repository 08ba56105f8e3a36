#!/usr/bin/env python
"""Headline benchmark: GP fit + predict on synthetic N x d fp64 tables (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c2|c1|c4|c5] [--map-evals E]

One GPU (default; BASELINE.json configs[2] = C3, the largest single-GPU configuration: N = 50k, d = 8,
Matern-5/2 ARD, M = 10^4 grid).  A *step* is one pass of the hot path the reference reaches through
``gp.fit()`` + ``gp.predict_grid()``, inputs already resident in HBM:

    find_MAP  -- L-BFGS-B on -(log-lik + log-priors) TO CONVERGENCE (scipy defaults, what pm.find_MAP runs);
                 every objective / gradient evaluation = covariance build -> Cholesky -> v = L^-1 y, log-det
                 -> L^-T, Sigma^-1, fused trace reductions
    predict   -- mean / variance on the M-point grid against the resident factor     (all in libgumbi_hip.so)

The model is declared as a Gumbi user who knows the scale of the inputs would: ``ls_bounds`` (the reference's own
knob, pymc/GP.py:630-650, built with ``make_deltas_parray``) says "length scales of at least LS_LOWER_Z standard
deviations".  With the PyMC DEFAULTS the lengthscale prior comes from the smallest pairwise gap (0.01 at these
N), L-BFGS-B starts at l = 0.023 where K = I and -- at C3 -- stops after 8 evaluations on that white-noise
plateau having learnt nothing; that run is kept as the side figure ``default_start``.
``fit_quality`` says what the timed fits found: RMSE / correlation of the grid mean against the generator's
noise-free f, sigma-hat against the generator's, the final NLML, evaluations, scipy's convergence flag.

``value`` = algorithmic GFLOP/s of the whole step; ``ms_per_step`` = fit + predict wall time.  Warm-up steps run
the same code path with the evaluation budget capped at WARMUP_EVALS (they are untimed; a full fit is ~30
evaluations = 55 s at C3).  The same JSON line carries ``roofline`` (the Cholesky's trailing-update SYRK/GEMM
launches alone -- the kernel the north star's MFMA target is stated on -- from per-launch HIP events of a few
profiled evaluations after the timed region, plus the same figure over every GEMM launch, plus the register-only
MFMA ceiling sampled for >= 1 s BEFORE and AFTER the timed region), ``kbuild`` (HBM GB/s of the covariance build),
``phases``, ``cpu_baseline`` (the oracle on a bounded sample on the host cores) and, while the time budget
(GUMBI_BENCH_BUDGET_S, default 1400 s of process time: a quarter of the driver's 1800 s stays free) allows, ``c2_single_gpu`` (the N = 10k fit whose factorisation is ONE
launch of the persistent tile kernel: step, phases, the launch's roofline), ``strong_scaling_base_gflops`` / ``c5_single_gpu``
(the N = 100k problem of the multi-GPU runs on this one GPU), ``default_start`` and ``end_to_end`` (the user-level
``DataSet -> GP.fit() -> prepare_grid() -> predict_grid()`` wall time, host transfers included).

Several GPUs (``--gpus N``: under torch.distributed.run when the driver launches it, one rank per GPU; started WITHOUT a
launcher -- no WORLD_SIZE in the environment -- bench.py re-runs itself as N ranks under torch.distributed.run on 127.0.0.1;
fewer than N devices, or a launcher whose WORLD_SIZE differs from --gpus, is an error line + non-zero exit, never a silent
one-GPU run): ONE GP -- BASELINE.json
configs[4] = C5, N = 100k, d = 8, RBF-ARD -- factored block-cyclically over all ranks by the native driver
(gumbi_amd/csrc/dist_driver.hpp; all-gathers on RCCL over xGMI).  Default: hyper-parameters fixed, a step = one
MAP objective + gradient evaluation (factorise + row-partitioned gradient) + re-factorisation (``fit`` at fixed
theta) + prediction of the 10^4 grid sharded over the ranks.  ``--map-evals k``: a step = a distributed
``GP(distributed=True).find_MAP(maxeval=k)`` from the same declared model + the grid prediction.  ``scaling`` is
"strong" (the same problem at every N); the line carries ``comm`` (transport, rank count as RCCL reports it,
per-phase collective time and how much of it nothing hid -- gmb_timings.dist_*) and
``strong_scaling_base_gflops``: the same step on ONE GPU of the same node, run by rank 0 after the timed region.
The barrier / max-over-ranks timing contract is the same.
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

LS_LOWER_Z = 0.5     # ls_bounds of the headline fit: length scales >= half a standard deviation of the input
WARMUP_EVALS = 3     # evaluation budget of an (untimed) warm-up step
PROFILE_EVALS = 4    # evaluations of the profiled pass behind `roofline` (per-launch HIP events)
C2_HOST_CAPPED_EVALS = 4  # evaluation cap of the HOST's C2 fit when the time budget is too short for the whole fit (cpu_baseline)
T_PROCESS_START = time.perf_counter()

CONFIGS = {
    # name: (N, d, kernel, M-grid resolution)
    "c1": dict(N=392, d=1, kernel="ExpQuad", res=100, label="C1-like synthetic N=392 d=1 RBF"),
    "c2": dict(N=10_000, d=4, kernel="ExpQuad", res=100, label="synthetic N=10k d=4 RBF-ARD fp64, M=10^4 grid"),
    "c3": dict(N=50_000, d=8, kernel="Matern52", res=100, label="synthetic N=50k d=8 Matern-5/2 ARD fp64, M=10^4 grid"),
    "c4": dict(N=20_000, d=4, kernel="ExpQuad", res=100, P=2,
               label="2-output coregionalized GP (ICM), N=20k rows x 2 outputs (stacked 40k), d=4 RBF-ARD fp64, Kronecker path, "
                     "M=10^4 grid per output"),
    "c5": dict(N=100_000, d=8, kernel="ExpQuad", res=100,
               label="synthetic N=100k d=8 RBF-ARD fp64, fixed theta, M=10^4 grid, ONE GP block-cyclic over the GPUs"),
}
# test hook: shrink the one-GP problem (the contract tests run it on a shared GPU in seconds)
if os.environ.get("GUMBI_BENCH_DIST_N"):
    CONFIGS["c5"]["N"] = int(os.environ["GUMBI_BENCH_DIST_N"])
    CONFIGS["c5"]["label"] = CONFIGS["c5"]["label"].replace("N=100k", f"N={CONFIGS['c5']['N']}")


if os.environ.get("GUMBI_BENCH_C4_N"):  # test hook: shrink the two-output problem
    CONFIGS["c4"]["N"] = int(os.environ["GUMBI_BENCH_C4_N"])
    CONFIGS["c4"]["label"] = CONFIGS["c4"]["label"].replace("N=20k", f"N={CONFIGS['c4']['N']}")


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


def synthetic_truth(cfg):
    """What the generator knows and the fit does not: the noise-free f on the prediction grid and the noise
    level, both in the z-scored units of y (the units the engine works in)."""
    N, d = cfg["N"], cfg["d"]
    rng = np.random.default_rng(2021)
    X = rng.standard_normal((N, d))
    ls = np.geomspace(0.7, 2.0, d) if d > 1 else np.array([1.0])
    y = np.sum(np.sin(X / ls), axis=1) / np.sqrt(d) + 0.2 * rng.standard_normal(N)
    Xs = synthetic_grid(d, cfg["res"])
    f_grid = (np.sum(np.sin(Xs / ls), axis=1) / np.sqrt(d) - y.mean()) / y.std(ddof=1)
    return f_grid, 0.2 / y.std(ddof=1)


def fit_quality(gp, mean, cfg):
    """Did the fit fit?  Grid mean against the generator's f, sigma-hat against the generator's sigma."""
    f_grid, sigma_z = synthetic_truth(cfg)
    mean = np.asarray(mean, dtype=float)
    res = getattr(gp, "opt_result", None)
    sig = float(np.asarray(gp.MAP["σ"]))
    return {
        "corr": round(float(np.corrcoef(mean, f_grid)[0, 1]), 5) if np.std(mean) > 0 else 0.0,
        "rmse": round(float(np.sqrt(np.mean((mean - f_grid) ** 2))), 5),
        "rmse_of_predicting_zero": round(float(np.sqrt(np.mean(f_grid**2))), 5),
        "sigma_hat": round(sig, 5), "sigma_true": round(float(sigma_z), 5), "sigma_rel_err": round(abs(sig / sigma_z - 1.0), 4),
        "nlml_final": round(float(gp.nlml_trace[-1]), 3) if gp.nlml_trace else None,
        "n_eval": int(gp.n_eval), "rejected_evals": int(gp._rejected),
        "converged": bool(res.success) if res is not None else None,
        "optimizer_message": (res.message if isinstance(res.message, str) else res.message.decode())[:80] if res is not None else None,
        "ls": [round(float(v), 3) for v in np.atleast_1d(gp.MAP["ls_total"])], "eta": round(float(np.asarray(gp.MAP["η_total"])), 4),
    }


def make_dataset(cfg):
    import pandas as pd

    import gumbi_amd as gmb

    X, y, _ = synthetic_table(cfg["N"], cfg["d"])
    cols = [f"x{k}" for k in range(cfg["d"])]
    df = pd.DataFrame(X, columns=cols)
    df["y"] = y
    return gmb.DataSet(df, outputs=["y"]), cols


def ls_bounds_for(ds, cols, ls_lower):
    """``ls_bounds`` as a Gumbi user states them (make_deltas_parray, gumbi/array_utils.py:8-33): a lower limit
    of ``ls_lower`` standard deviations on every input's length scale, the upper limit left to the default."""
    import gumbi_amd as gmb

    if not ls_lower:
        return None  # PyMC defaults: limits from the pairwise distances (gp_utils.py:34-46)
    return gmb.make_deltas_parray(stdzr=ds.stdzr, scale="standardized", **{c: [float(ls_lower), None] for c in cols})


def build_gp(cfg, device, ls_lower=LS_LOWER_Z, distributed=None):
    """Front-end objects exactly as a Gumbi user builds them (DataSet -> GP -> build_model)."""
    import gumbi_amd as gmb

    ds, cols = make_dataset(cfg)
    gp = gmb.GP(ds, outputs=["y"], device=device, distributed=distributed)
    gp.specify_model(continuous_dims=cols)
    gp.build_model(continuous_kernel=cfg["kernel"], ls_bounds=ls_bounds_for(ds, cols, ls_lower))
    return gp


class Budget:
    """Wall-clock budget of the whole process (the driver gives a run 1800 s): side sections are skipped, and
    say so in the JSON, when starting them would overrun it; the timed steps never are."""

    def __init__(self):
        self.total = float(os.environ.get("GUMBI_BENCH_BUDGET_S", "1400"))

    def used(self):
        return time.perf_counter() - T_PROCESS_START

    def allows(self, estimate_s):
        return self.used() + estimate_s <= self.total

    def skipped(self, estimate_s):
        return {"skipped": f"time budget: {self.used():.0f} s used of {self.total:.0f} s, this section needs ~{estimate_s:.0f} s "
                           f"(GUMBI_BENCH_BUDGET_S)"}


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
    out = {"bytes_per_launch": gbytes * 1e9 / launches, "source": os.path.relpath(files[-1], str(ROOT)), "launches": launches,
           "flops_per_launch": flops / launches, "kernel_sources_sha16_then": None, "kernel_sources_sha16_now": kernel_sources_sha16()}
    meta = files[-1].replace("_summary.csv", "_summary.meta.json")
    if os.path.exists(meta):  # written by tools/gpu_pmc_bench.sh: the kernel sources the counters were taken on
        with open(meta) as fh:
            out["kernel_sources_sha16_then"] = json.load(fh).get("kernel_sources_sha16")
    out["stale"] = out["kernel_sources_sha16_then"] != out["kernel_sources_sha16_now"]
    return out


def kernel_sources_sha16():
    """sha256 over the CODE of gumbi_amd/csrc/* (sorted; `//` comments and white space stripped, so that rewording a
    comment does not disown a counter summary): ties a committed PMC summary to the kernels it was taken on."""
    import hashlib
    import re

    h = hashlib.sha256()
    for path in sorted((ROOT / "gumbi_amd" / "csrc").glob("*.h*")):
        h.update(path.name.encode())
        code = re.sub(r"//[^\n]*", "", path.read_text())
        h.update("".join(code.split()).encode())
    return h.hexdigest()[:16]


def cpu_baseline(cfg, target_seconds=20.0):
    """Oracle (numpy + LAPACK through scipy) timed on a bounded sample of the same workload:
    one MAP objective+gradient evaluation and one grid prediction at a reduced N chosen (from a
    short probe, cost ~ N^3) to take about ``target_seconds`` on this host."""
    from oracle import gp_oracle as O

    d = cfg["d"]
    Xs = O.synthetic_grid(d, cfg["res"])
    spec = O.make_spec(d, range(d), kind=cfg["kernel"])

    split = {}

    def run(n):
        X, y, ls = O.synthetic_table(n, d)
        theta = O.pack_theta(spec, ls, 1.0, 0.2)
        t0 = time.perf_counter()
        O.nlml_and_grad(spec, theta, X, y, dist_mode="gemm", inverse="potri")
        t1 = time.perf_counter()
        O.predict(spec, theta, X, y, Xs, with_noise=True)
        split["evaluation"], split["predict"] = t1 - t0, time.perf_counter() - t1
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
    cores, blas = host_blas()
    return {
        "value": round(flops / dt / 1e9, 2),
        "unit": "GFLOP/s",
        "cores": int(cores),
        "blas": blas,
        "kind": "port",
        "sample": f"1 MAP objective+gradient evaluation + predict(M={len(Xs)}) at N={Ns}, d={d}, {cfg['kernel']} "
                  f"(numpy/LAPACK oracle, {dt:.1f} s)",
        "seconds": round(dt, 2),
        "evaluation_seconds": round(split["evaluation"], 2), "predict_seconds": round(split["predict"], 2),
    }


def host_blas():
    """(threads, "vendor version (threading layer)") of the BLAS numpy / scipy call on this host."""
    cores, blas = os.cpu_count() or 1, "unknown"
    try:
        from threadpoolctl import threadpool_info

        pools = [p for p in threadpool_info() if p.get("user_api") == "blas"] or threadpool_info()
        cores = max([p.get("num_threads", 1) for p in pools] + [1])
        if pools:
            p = pools[0]
            blas = f"{p.get('internal_api', '?')} {p.get('version', '?')} ({p.get('threading_layer', '?')}, {p.get('architecture', '?')})"
    except Exception:
        pass
    return cores, blas


def cpu_config_size_sections(budget, gpu_c2=None):
    """The host beside the GPU at the CONFIG's own sizes, each only while the time budget lasts:

    c2  -- the oracle behind the SAME host code (HipGP.find_MAP / predict with tests/oracle_engine.OracleEngine standing in
           for the HIP engine: same priors, same L-BFGS-B, same declaration): one objective + gradient evaluation and one
           grid prediction at N = 10k always (~12 s); the whole fit to convergence + prediction in a SUBPROCESS when the ~3.5
           minutes it takes are left (the evaluation count is then the optimiser's own); with less time a fit CAPPED at
           C2_HOST_CAPPED_EVALS evaluations is measured instead and the whole fit is that per-evaluation time x the GPU fit's
           evaluation count + one prediction, labelled as an extrapolation;
    c3  -- LAPACK dpotrf alone on a 50,000 x 50,000 SPD matrix (20 GB; the factorisation of ONE of the fit's ~30
           evaluations);
    c5  -- N = 100k: never run, the N^3 extrapolation of the c3 figure, labelled as one."""
    import subprocess

    def host_fit(maxeval, est):
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--cpu-fit", "c2"], cwd=str(ROOT), capture_output=True, text=True,
                           timeout=est * 1.5 + 60.0, env=dict(os.environ, GUMBI_BENCH_NO_CPU="1", GUMBI_BENCH_CPU_FIT_MAXEVAL=str(maxeval)))
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and line:
            return json.loads(line[-1])
        return {"error": (r.stderr or r.stdout)[-300:]}

    out = {}
    cfg = CONFIGS["c2"]
    est_eval = 25.0
    if budget.allows(est_eval):
        one = cpu_baseline(cfg, target_seconds=1e9)  # (target beyond reach: grows straight to the full N)
        out["c2_one_evaluation_plus_predict_seconds"] = one["seconds"]
        out["c2_one_evaluation_seconds"] = one["evaluation_seconds"]
        out["c2_sample"] = one["sample"]
        n_eval = int(gpu_c2["n_eval"]) if gpu_c2 and gpu_c2.get("n_eval") else 29
        # a fit = its evaluations + ONE prediction (+ the subprocess's start-up and table)
        est_fit = 1.08 * (n_eval + 1) * one["evaluation_seconds"] + one["predict_seconds"] + 25.0
        est_capped = 1.08 * (C2_HOST_CAPPED_EVALS + 2) * one["evaluation_seconds"] + one["predict_seconds"] + 25.0
        out["c2_seconds_estimate"] = round(n_eval * one["evaluation_seconds"] + one["predict_seconds"], 1)
        out["c2_seconds_estimate_note"] = f"{n_eval} evaluations (the GPU fit's count) x the measured evaluation + one prediction; NOT a measured fit"
        out["c2_seconds"] = None
        try:
            if budget.allows(est_fit):
                fit = host_fit(200, est_fit)
                out["c2_fit"] = fit
                out["c2_seconds"] = fit.get("fit_predict_seconds")
            elif budget.allows(est_capped):
                fit = host_fit(C2_HOST_CAPPED_EVALS, est_capped)
                out["c2_fit_capped"] = fit
                out["c2_fit"] = budget.skipped(est_fit)
                if fit.get("n_eval"):
                    per_eval = fit["fit_seconds"] / fit["n_eval"]  # (declaration and priors included: what the fit pays per evaluation)
                    out["c2_seconds_extrapolated"] = round(per_eval * n_eval + fit["predict_seconds"], 1)
                    out["c2_seconds_extrapolated_note"] = (
                        f"EXTRAPOLATION, not a measured fit: a host fit capped at {fit['n_eval']} evaluations measured {fit['fit_seconds']} s "
                        f"= {per_eval:.2f} s per evaluation, x the GPU fit's {n_eval} evaluations + the measured prediction "
                        f"({fit['predict_seconds']} s); the time budget did not allow the ~{est_fit:.0f} s of the whole host fit "
                        "(profiles/r05_bench_c3_driver_like.json holds a measured one: 173 s)")
            else:
                out["c2_fit"] = budget.skipped(est_capped)
        except Exception as err:  # noqa: BLE001
            out["c2_fit"] = {"error": f"{type(err).__name__}: {err}"[:300]}
    else:
        out["c2_seconds"] = None
        out["c2_fit"] = budget.skipped(est_eval)
    est_c3 = 75.0  # (measured: 43 - 45 s for the factorisation + ~15 s to fill 20 GB)
    if budget.allows(est_c3):
        try:
            out.update(cpu_dpotrf_seconds(CONFIGS["c3"]["N"]))
        except Exception as err:  # noqa: BLE001
            out["c3_cholesky_seconds"] = None
            out["c3_cholesky"] = {"error": f"{type(err).__name__}: {err}"[:300]}
    else:
        out["c3_cholesky_seconds"] = None
        out["c3_cholesky"] = budget.skipped(est_c3)
    if out.get("c3_cholesky_seconds"):
        out["c5_cholesky_seconds_extrapolated"] = round(out["c3_cholesky_seconds"] * (CONFIGS["c5"]["N"] / CONFIGS["c3"]["N"]) ** 3, 1)
        out["c5_note"] = "N = 100k is not run on the host: c3's dpotrf time x (100k / 50k)^3, an extrapolation (BASELINE.md section 3)"
    return out


def cpu_dpotrf_seconds(N):
    """LAPACK dpotrf (scipy.linalg.cholesky, in place, all BLAS threads) on an N x N SPD matrix -- what PyTensor's Cholesky op
    dispatches to once per objective evaluation.  The values do not matter for the time; the matrix is N I + 0.5."""
    import scipy.linalg

    need = 8.0 * N * N * 1.15
    try:
        import psutil

        avail = float(psutil.virtual_memory().available)
    except Exception:
        avail = need * 2
    if avail < need * 1.3:
        return {"c3_cholesky_seconds": None, "c3_cholesky": {"skipped": f"host memory: {avail / 1e9:.0f} GB available, {need / 1e9:.0f} GB needed"}}
    W = np.full((2048, 2048), 0.5, order="F")  # wake the BLAS threads up first
    W[np.diag_indices(2048)] += 2048.0
    scipy.linalg.cholesky(W, lower=True, overwrite_a=True, check_finite=False)
    A = np.full((N, N), 0.5, order="F")
    A[np.diag_indices(N)] += float(N)
    t0 = time.perf_counter()
    scipy.linalg.cholesky(A, lower=True, overwrite_a=True, check_finite=False)
    dt = time.perf_counter() - t0
    del A
    cores, blas = host_blas()
    return {"c3_cholesky_seconds": round(dt, 2),
            "c3_cholesky": {"N": N, "gflops": round(N**3 / 3.0 / dt / 1e9, 1), "cores": cores, "blas": blas,
                            "what": f"scipy.linalg.cholesky(lower=True, overwrite_a=True) = LAPACK dpotrf on a {8.0 * N * N / 1e9:.1f} GB SPD matrix, once"}}


def cpu_fit_subprocess(config_name):
    """``python bench.py --cpu-fit c2``: the whole step -- fit to convergence + grid prediction -- of the config on the HOST: the
    product's host code (HipGP: priors, L-BFGS-B, declaration with ls_bounds) with the numpy / LAPACK oracle standing in for
    the HIP engine (tests/oracle_engine.py).  Runs in a process of its own so that nothing of the stand-in can leak into the
    measured GPU path; prints one JSON line."""
    sys.path.insert(0, str(ROOT / "tests"))
    from oracle_engine import HostBaselineEngine

    import gumbi_amd as gmb
    from gumbi_amd.regression import hip_gp

    hip_gp.Engine = HostBaselineEngine
    cfg = CONFIGS[config_name]
    t0 = time.perf_counter()
    ds, cols = make_dataset(cfg)
    gp = gmb.GP(ds, outputs=["y"], device=0)
    gp.fit(continuous_dims=cols, continuous_kernel=cfg["kernel"], ls_bounds=ls_bounds_for(ds, cols, LS_LOWER_Z), MAP_kwargs={"maxeval": int(os.environ.get("GUMBI_BENCH_CPU_FIT_MAXEVAL", "200"))})
    t_fit = time.perf_counter() - t0
    t1 = time.perf_counter()
    if cfg["d"] > 2:
        gp.prepare_grid(at=gp.parray(**{c: 0.0 for c in cols[2:]}, stdzd=True), resolution=cfg["res"])
    else:
        gp.prepare_grid(resolution=cfg["res"])
    gp.predict_grid()
    t_pred = time.perf_counter() - t1
    cores, blas = host_blas()
    print(json.dumps({"fit_predict_seconds": round(t_fit + t_pred, 2), "fit_seconds": round(t_fit, 2), "predict_seconds": round(t_pred, 2),
                      "n_eval": int(gp.n_eval), "nlml_final": round(float(gp.nlml_trace[-1]), 3), "cores": cores, "blas": blas,
                      "what": "HipGP host code over the numpy/LAPACK oracle (tests/oracle_engine.py: HostBaselineEngine -- PyMC's GEMM distance "
                              "expansion, one factorisation per evaluation), same declaration as the GPU fit"}), flush=True)


def tile_chain_summary(eng):
    """One traced factorisation on the persistent tile kernel (csrc/chol_tiles.hpp; matrices of 16 .. 224 block columns): the
    latency chain diagonal tile -> sub-diagonal tile -> next diagonal tile, per block column, and the contraction time per
    k-block of the bulk tiles, from the per-task wall-clock stamps of gmb_chol_task_trace."""
    eng.chol_task_trace(1)
    try:
        eng.factorize()
        tr = eng.chol_task_trace(0)
    finally:
        eng.chol_task_trace(0)
    if tr is None:
        return None
    return summarize_tile_trace(*tr)


def summarize_tile_trace(tiles, stamps):
    """(tiles, stamps) of Engine.chol_task_trace -> chain step / k-block statistics (pure numpy: tests/test_bench_host.py)."""
    st = np.asarray(stamps, dtype=float) * 1e6  # us: taken / contraction done / solve input ready / published
    tiles = np.asarray(tiles)
    diag = {int(j): st[i] for i, (ii, j) in enumerate(tiles) if ii == j}
    pub = np.array([diag[j][3] for j in sorted(diag)])
    steps = np.diff(pub)
    per_kb = [(st[i, 1] - st[i, 0]) / j for i, (ii, j) in enumerate(tiles) if j >= 8 and ii >= j + 2]
    return {
        "kernel": "chol_tiles_kernel<8> (one launch per factorisation)",
        "block_columns": int(len(diag)), "tasks": int(len(tiles)),
        "launch_us_traced": round(float(st[:, 3].max() - st[:, 0].min()), 1),
        # distance between the publications of consecutive diagonal tiles: p10 ~ the latency chain's own step (the columns at
        # both ends of the matrix, where nothing else limits it); the median sits in the throughput-bound middle
        "chain_step_us": {"p10": round(float(np.percentile(steps, 10)), 1), "median": round(float(np.median(steps)), 1),
                          "min": round(float(steps.min()), 1), "p90": round(float(np.percentile(steps, 90)), 1)} if len(steps) else None,
        "leaf_us_median": round(float(np.median([d[3] - d[2] for d in diag.values()])), 1),
        "contraction_us_per_k_block": {"median": round(float(np.median(per_kb)), 2), "max": round(float(np.max(per_kb)), 2)} if per_kb else None,
        "note": "stamps cost a few per cent: the untraced launch is phases.profiled_last_evaluation_ms.chol_ms",
    }


def roofline_block(tm, config):
    """``roofline`` from the engine's per-launch HIP events (gmb_timings, cumulative since profiling was
    switched on): the trailing-update launches of the Cholesky alone, and every GEMM launch beside it."""
    chol_tf = tm["total_chol_gemm_flops"] / max(tm["total_chol_gemm_ms"], 1e-9) / 1e9
    all_tf = tm["total_gemm_flops"] / max(tm["total_gemm_ms"], 1e-9) / 1e9
    n_chol = max(int(tm["total_chol_gemm_launches"]), 1)
    out = {
        "bound": "mfma",
        "kernel": "gemm_f64_dma_chol_update_kernel (128 x 128 tiles, LDS-DMA staged; the entry point of the Cholesky's bulk "
                  "trailing-update launches, v_mfma_f64_16x16x4_f64 SYRK/GEMM -- rocprofv3 lists exactly this population under that "
                  "name; gemm_f64_dma_kernel is the same body for every other 128 x 128 product)",
        "subset_note": "achieved / frac / avg_launch_ms are SUBSET figures: ONLY the Cholesky's bulk trailing updates (the launches the "
                       "north star states its MFMA target on); all_gemm_launches beside it covers every MFMA GEMM launch of the "
                       "profiled evaluations, and the line's top-level roofline_frac_whole_step = value / peak covers the whole timed step",
        "reproduce_from_profiles": "profiles/*_bench_c3_kernel_stats.csv: AverageNs of gemm_f64_dma_chol_update_kernel; "
                                   "profiles/*_pmc_bench_c3_summary.csv: its MFMA_F64_TFLOPs (SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 / duration) "
                                   "and flops per launch; frac = TFLOP/s / 78.6",
        #  the population behind achieved / frac CHANGED between rounds 4 and 5: r04 = every trailing-update launch of
        # the Cholesky, whatever tile shape it ran (1950 launches of ~107 GF per profiled pass at C3: 0.914); r05 on = only the
        # updates large enough for the 128 x 128 kernel (215 launches of ~944 GF: 0.923).  The figures of the two rounds are NOT
        # like-for-like; `all_trailing_updates` below is the r04 population, measured in this run, and the trend metric across
        # rounds is roofline_frac_whole_step (top level) / all_gemm_launches
        "population": "bulk trailing updates that run the 128 x 128 LDS-DMA kernel (ev kind 7); smaller updates (other tile shapes, "
                      "in-place) are counted under in_panel_products",
        "population_changed_since": "r04 (r04 = all_trailing_updates below; r04's 0.914 and r05's 0.923 are different populations)",
        "achieved": round(chol_tf, 3),
        "peak": FP64_MFMA_PEAK_TFLOPS,
        "unit": "TFLOP/s",
        "frac": round(chol_tf / FP64_MFMA_PEAK_TFLOPS, 4),
        "traffic": None,
        "measured_over": f"{PROFILE_EVALS} evaluations + one prediction of the same workload after the timed region (per-launch HIP "
                         "events on the launch's own stream)",
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
    if tm.get("total_chol_update_all_launches", 0) > 0:
        # every launch the Cholesky's schedules issue AS a trailing update (the population of `roofline` up to round 4)
        a_tf = tm["total_chol_update_all_flops"] / max(tm["total_chol_update_all_ms"], 1e-9) / 1e9
        out["all_trailing_updates"] = {"achieved": round(a_tf, 3), "frac": round(a_tf / FP64_MFMA_PEAK_TFLOPS, 4),
                                       "launches": int(tm["total_chol_update_all_launches"]),
                                       "avg_launch_ms": round(tm["total_chol_update_all_ms"] / tm["total_chol_update_all_launches"], 5),
                                       "flops_per_launch": round(tm["total_chol_update_all_flops"] / tm["total_chol_update_all_launches"], 1)}
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
    if int(tm.get("total_eval_tile_launches", 0)) > 0 and int(tm["total_chol_gemm_launches"]) == 0:
        # N <= ~28k: the matrix work of a MAP evaluation is ONE launch of the persistent evaluation kernel (csrc/eval_tiles.hpp):
        # the tile Cholesky, L^-T by rows and Sigma^-1 = U U^T as 128 x 128 tile tasks from one ticket counter -- the roofline is
        # that whole launch, latency chain, leaves and strip solves included
        n_t = int(tm["total_eval_tile_launches"])
        tf = tm["total_eval_tile_flops"] / max(tm["total_eval_tile_ms"], 1e-9) / 1e9
        n_alg = float(CONFIGS[config]["N"]) if config in CONFIGS else 0.0
        alg_tf = n_alg**3 * n_t / max(tm["total_eval_tile_ms"], 1e-9) / 1e9  # N^3/3 each: Cholesky, inverse, Sigma^-1 (unpadded)
        out.update({
            "kernel": "eval_tiles_kernel<8>, the persistent evaluation launch: tile Cholesky + L^-T by rows + Sigma^-1 = U U^T as "
                      "128 x 128 tile tasks on v_mfma_f64_16x16x4_f64 (LDS-DMA staged contraction, leaves and strip solves inside, "
                      "flags between workgroups) -- ONE launch per MAP evaluation",
            "subset_note": "achieved / frac cover the whole evaluation launch (factorisation + inverse + Sigma^-1), latency chain "
                           "included; flops = the tiles' contractions on the 128-padded matrix, achieved_on_unpadded_N3 = N^3 per "
                           "launch over the same time; all_gemm_launches beside it adds the prediction's launches",
            "achieved": round(tf, 3), "frac": round(tf / FP64_MFMA_PEAK_TFLOPS, 4), "launches": n_t,
            "avg_launch_ms": round(tm["total_eval_tile_ms"] / n_t, 5), "flops_per_launch": round(tm["total_eval_tile_flops"] / n_t, 1),
            "achieved_over_wall_time": round(tf, 3),
            "achieved_on_unpadded_N3": round(alg_tf, 3), "frac_on_unpadded_N3": round(alg_tf / FP64_MFMA_PEAK_TFLOPS, 4),
        })
        out.pop("in_panel_products", None)
        pt = pmc_traffic(config, "eval_tiles_kernel<8>")
        if pt is not None and pt["launches"] > 0:
            out["traffic"] = round(pt["bytes_per_launch"], 1)
            out["traffic_unit"] = ("HBM-side bytes per evaluation launch: (2*FETCH_SIZE + WRITE_SIZE) of eval_tiles_kernel<8> over one "
                                   "bench step (rocprofv3 PMC passes) / its launches; L, U and Sigma^-1 themselves are 3 x 8 N^2 / 2 "
                                   "bytes -- the rest is operand tiles re-read from HBM (the reader sits on another XCD's L2)")
            out["traffic_flop_per_byte"] = round(tm["total_eval_tile_flops"] / n_t / max(pt["bytes_per_launch"], 1.0), 2)
            out["traffic_source"] = pt["source"]
            out["traffic_is_stale"] = pt["stale"]
            out["traffic_kernel_sources_sha16"] = {"then": pt["kernel_sources_sha16_then"], "now": pt["kernel_sources_sha16_now"]}
        return out
    if int(tm.get("total_chol_tile_launches", 0)) > 0 and int(tm["total_chol_gemm_launches"]) == 0:
        # N <= ~20k: the factorisation is ONE launch of the persistent tile kernel (csrc/chol_tiles.hpp) -- there are no
        # separate trailing-update launches to quote; the roofline is the whole launch: contraction, leaves and strip
        # solves, latency chain included (flops = the tiles' contractions)
        n_t = int(tm["total_chol_tile_launches"])
        tf = tm["total_chol_tile_flops"] / max(tm["total_chol_tile_ms"], 1e-9) / 1e9
        out.update({
            "kernel": "chol_tiles_kernel<8>, the persistent tile Cholesky: the WHOLE factorisation in one launch (left-looking "
                      "128 x 128 tile tasks on v_mfma_f64_16x16x4_f64, leaves and strip solves inside, flags between workgroups)",
            "subset_note": "achieved / frac cover the whole factorisation launch, latency chain included -- this size has no separate "
                           "trailing-update launches; all_gemm_launches beside it adds the gradient's and the prediction's GEMM launches",
            "achieved": round(tf, 3), "frac": round(tf / FP64_MFMA_PEAK_TFLOPS, 4), "launches": n_t,
            "avg_launch_ms": round(tm["total_chol_tile_ms"] / n_t, 5), "flops_per_launch": round(tm["total_chol_tile_flops"] / n_t, 1),
            "achieved_over_wall_time": round(tf, 3),
        })
        out.pop("in_panel_products", None)
        pt = pmc_traffic(config, "chol_tiles_kernel<8>")  # every launch of this kernel is one factorisation: bytes per launch as counted
        if pt is not None and pt["launches"] > 0:
            out["traffic"] = round(pt["bytes_per_launch"], 1)
            out["traffic_unit"] = ("HBM-side bytes per factorisation launch: (2*FETCH_SIZE + WRITE_SIZE) of chol_tiles_kernel<8> over one "
                                   "bench step (rocprofv3 PMC passes) / its launches; the lower triangle itself is 8 N^2 / 2 bytes -- the "
                                   "rest is operand tiles re-read from HBM because the reader sits on another XCD's L2")
            out["traffic_flop_per_byte"] = round(tm["total_chol_tile_flops"] / n_t / max(pt["bytes_per_launch"], 1.0), 2)
            out["traffic_source"] = pt["source"]
            out["traffic_is_stale"] = pt["stale"]
            out["traffic_kernel_sources_sha16"] = {"then": pt["kernel_sources_sha16_then"], "now": pt["kernel_sources_sha16_now"]}
        return out
    pt = pmc_traffic(config, "gemm_f64_dma_chol_update_kernel")  # the bulk updates' own entry point (round 5)
    if pt is None or pt["flops_per_launch"] <= 0:
        pt = pmc_traffic(config, "gemm_f64_dma_kernel")  # (summaries of round 4: one name for every 128 x 128 product)
    if pt is not None and pt["flops_per_launch"] > 0:
        # the PMC passes count every launch of the 128x128 instantiation (bulk updates, solves, inverse, Sigma^-1,
        # predict: other launch sizes than the ones `achieved` is quoted on), so the bytes are scaled by flops
        bytes_per_flop = pt["bytes_per_launch"] / pt["flops_per_launch"]
        out["traffic"] = round(bytes_per_flop * tm["total_chol_gemm_flops"] / n_chol, 1)
        out["traffic_unit"] = ("HBM-side bytes per bulk-update launch: (2*FETCH_SIZE + WRITE_SIZE) per flop of "
                               "gemm_f64_dma_chol_update_kernel over one bench step (rocprofv3 PMC passes), times this launch's flops")
        out["traffic_flop_per_byte"] = round(1.0 / bytes_per_flop, 2)
        out["traffic_source"] = pt["source"]
        out["traffic_is_stale"] = pt["stale"]  # the counters were taken on other kernel sources than the ones running now
        out["traffic_kernel_sources_sha16"] = {"then": pt["kernel_sources_sha16_then"], "now": pt["kernel_sources_sha16_now"]}
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

    from gumbi_amd import engine as E

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

    def one_step(budget=maxeval):
        gp.find_MAP(maxeval=budget)
        eng.predict_device(xs_dev.data_ptr(), M, cfg["d"], mean_dev.data_ptr(), var_dev.data_ptr(), True)
        return gp.n_eval, gp.n_refactor

    for _ in range(warmup):
        one_step(min(maxeval, WARMUP_EVALS))
    torch.cuda.synchronize()
    ceiling_before = E.mfma_f64_sustained(local_rank, 1.0)  # register-only MFMA loop, >= 1 s, right before the timed region
    clock.sync()
    t0 = time.perf_counter()
    counts = [one_step() for _ in range(steps)]
    n_evals = [c[0] for c in counts]
    clock.sync()
    elapsed = time.perf_counter() - t0
    ceiling_after = E.mfma_f64_sustained(local_rank, 1.0)   # ... and right after it
    mean_host = mean_dev.cpu().numpy()
    var_host = var_dev.cpu().numpy()
    finite = bool(np.all(np.isfinite(mean_host)) and np.all(var_host > 0))
    quality = fit_quality(gp, mean_host, cfg)
    theta_fit = gp._theta_fitted.copy()
    # Roofline pass: a few more evaluations + one prediction of the same workload with a HIP event pair around
    # every GEMM launch (on the stream it is launched on).  Kept out of the timed region: event pairs cost host
    # time per launch.
    eng.set_profiling(True)  # resets the totals
    one_step(min(maxeval, PROFILE_EVALS))
    torch.cuda.synchronize()
    tm = eng.timings()
    eng.set_profiling(False)

    # Phases of one evaluation timed separately, wall clock, unprofiled, at the
    # fitted hyper-parameters; outside the timed region.
    def wall_ms(fn, reps=2):
        best = float("inf")
        for _ in range(reps):
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t_)
        return round(1e3 * best, 3)

    def fact_then_grad():  # what find_MAP calls per evaluation: ONE C call (gmb_evaluate), for <= 224 block columns ONE tile launch
        eng.evaluate(theta_fit)

    eng.set_theta(theta_fit)
    phases = {
        "specify_plus_build_model_s": round(t_build, 3),  # DataSet -> specify_model -> build_model (ls priors, H2D copy)
        "factorize_ms": wall_ms(eng.factorize),          # K-build + Cholesky + L^-1 y + log-det
        "factorize_plus_gradient_ms": wall_ms(fact_then_grad, reps=3),  # one MAP objective evaluation (gmb_evaluate)
    }
    eng.factorize()
    phases["predict_ms"] = wall_ms(lambda: eng.predict_device(xs_dev.data_ptr(), M, cfg["d"], mean_dev.data_ptr(),
                                                                var_dev.data_ptr(), True))
    # the prediction as the timed step runs it: behind the fit's last evaluation, whose inverse factor is still resident -- one GEMM
    # with a triangular operand instead of the solve for matrices on the tile path (csrc/predict_form.hpp);
    # the first call also transposes the inverse once (what the step pays), the best of two more is the product alone
    eng.evaluate(theta_fit)
    predict = lambda: eng.predict_device(xs_dev.data_ptr(), M, cfg["d"], mean_dev.data_ptr(), var_dev.data_ptr(), True)  # noqa: E731
    phases["predict_after_fit_first_call_ms"] = wall_ms(predict, reps=1)
    phases["predict_after_fit_ms"] = wall_ms(predict)
    phases["predict_after_fit_form"] = "gemm against the resident inverse factor" if eng.timings().get("predict_gemm_form") else "triangular solve"
    phases["ls_limits_ms"] = wall_ms(lambda: gp._prepare_lengthscales(gp.model.X, ARD=True))
    N = cfg["N"]
    phases["rates_tflops"] = {
        "factorize": round(N**3 / 3.0 / phases["factorize_ms"] / 1e9, 2),
        "map_evaluation": round(float(N) ** 3 / phases["factorize_plus_gradient_ms"] / 1e9, 2),
        "predict": round(float(N) ** 2 * M / phases["predict_ms"] / 1e9, 2),
        "predict_after_fit": round(float(N) ** 2 * M / phases["predict_after_fit_ms"] / 1e9, 2),
    }
    phases["at_theta"] = "the MAP of the timed fits"
    phases["profiled_last_evaluation_ms"] = {k: round(tm[k], 3) for k in
                                             ("kbuild_ms", "chol_ms", "chol_leaf_ms", "chol_trsm_ms", "chol_gemm_ms",
                                              "grad_ms", "grad_gemm_ms", "predict_ms", "predict_gemm_ms")}
    phases["tile_cholesky"] = tile_chain_summary(eng)  # None when this size is factored by the stream schedules
    # the engine's streams go away before any other section creates its own: more than four live HIP
    # streams per process slow every kernel down on this stack
    eng.close()
    gp.engine = None
    return dict(elapsed=elapsed, n_evals=n_evals, n_refactor=[c[1] for c in counts], M=M, tm=tm, phases=phases, finite=finite,
                flops=sum(step_flops(N, M, n, r) for n, r in counts), quality=quality,
                ceiling={"before_timed_region": ceiling_before, "after_timed_region": ceiling_after})


def c2_side_section(local_rank, clock):
    """C2 (N = 10k, d = 4, RBF-ARD, M = 10^4) as a side figure of the default run: one timed MAP fit to convergence +
    prediction (after one capped warm-up step), the phases of one evaluation at the MAP and the roofline of its
    factorisation launch.  ``python bench.py --config c2`` is the full-length form."""
    cfg = CONFIGS["c2"]
    res = map_fit_workload(cfg, "c2", local_rank, 1, 1, 0, clock)
    elapsed = res["elapsed"]
    r = roofline_block(res["tm"], "c2")
    ph = res["phases"]
    return {
        "workload": cfg["label"],
        "ms_per_step": round(1e3 * elapsed, 3), "fit_predict_seconds": round(elapsed, 4), "map_evals": res["n_evals"],
        "value": round(res["flops"] / elapsed / 1e9, 2),
        "unit": "GFLOP/s", "fit_quality": {k: res["quality"][k] for k in ("corr", "sigma_rel_err", "converged", "n_eval") if k in res["quality"]},
        "phases": {k: ph[k] for k in ("factorize_ms", "factorize_plus_gradient_ms", "predict_ms", "predict_after_fit_ms", "predict_after_fit_first_call_ms",
                                      "predict_after_fit_form", "rates_tflops", "tile_cholesky") if k in ph},
        "factorisation_roofline": {k: r[k] for k in ("kernel", "achieved", "peak", "frac", "achieved_on_unpadded_N3", "frac_on_unpadded_N3", "launches", "avg_launch_ms", "traffic", "traffic_source", "traffic_is_stale") if k in r},
    }


def default_start_fit(cfg, local_rank):
    """Side figure: the same table fitted with the reference's DEFAULTS -- no ls_bounds, lengthscale prior from the
    smallest pairwise gap (gp_utils.py:34-46), PyMC's initial point (the prior's mode, l = 0.023)."""
    import torch

    gp = build_gp(cfg, device=local_rank, ls_lower=None)
    Xs = synthetic_grid(cfg["d"], cfg["res"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gp.find_MAP(maxeval=200)
    mean, var = gp.predict(Xs)
    dt = time.perf_counter() - t0
    out = {"fit_predict_seconds": round(dt, 3), "start_ls": [round(float(v), 4) for v in gp._initial_theta()[: cfg["d"]]],
           "fit_quality": fit_quality(gp, mean, cfg),
           "note": "the RESTATEMENT's behaviour with the reference's default prior and start (no ls_bounds): L-BFGS-B starts where K = I; "
                   "at C3 it stops there.  PyMC itself cannot run here, so this is not a statement about PyMC (DESIGN.md section 5)"}
    gp.engine.close()
    gp.engine = None
    return out


def end_to_end_fit(cfg, local_rank, map_evals):
    """The user-level call sequence of the reference (DataSet -> GP.fit -> prepare_grid -> predict_grid), wall
    clock, once: host-side plumbing, lengthscale priors, host -> device copies and result wrapping included."""
    import gumbi_amd as gmb

    t0 = time.perf_counter()
    ds, cols = make_dataset(cfg)
    t1 = time.perf_counter()
    gp = gmb.GP(ds, outputs=["y"], device=local_rank)
    gp.fit(continuous_dims=cols, continuous_kernel=cfg["kernel"], ls_bounds=ls_bounds_for(ds, cols, LS_LOWER_Z),
           MAP_kwargs={"maxeval": map_evals if map_evals > 0 else 200})
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
            "note": "gp.fit(ls_bounds=...) = specify_model + build_model + find_MAP, host <-> device transfers included"}


# -----------------------------------------------------------------------------------------------------
# workload C: the two-output coregionalised GP (C4) through the Kronecker path -- MAP fit + grid prediction on ONE GPU
# -----------------------------------------------------------------------------------------------------
C4_W = np.array([[1.0, 0.0], [0.6, 0.8]])
C4_KAPPA = np.array([0.1, 0.1])
C4_NOISE_SD = 0.2 * np.sqrt(np.array([1.0, 2.25]))
C4_SIDE_EVALS = 40   # evaluation budget of the C4 side figure in the default (C3) line


def c4_table(cfg, seed=2021):
    """SURVEY.md section 8d C4 generator (the one tests/test_gpu_configs.py::icm_problem restates): the same X for both
    outputs, y_p = sum_q chol(B)[p, q] f_q + noise_p, B = W W^T + diag(kappa), per-output noise 0.2 / 0.3."""
    n, d = cfg["N"], cfg["d"]
    rng = np.random.default_rng(seed)
    Xc = rng.standard_normal((n, d))
    ls = np.geomspace(0.7, 2.0, d)
    B = C4_W @ C4_W.T + np.diag(C4_KAPPA)
    Lb = np.linalg.cholesky(B)
    f = np.stack([np.sum(np.sin(Xc / ls + p), axis=1) / np.sqrt(d) for p in range(2)])
    Y = Lb @ f + C4_NOISE_SD[:, None] * rng.standard_normal((2, n))
    Xg = synthetic_grid(d, cfg["res"])
    Fg = Lb @ np.stack([np.sum(np.sin(Xg / ls + p), axis=1) / np.sqrt(d) for p in range(2)])
    mean, sd = Y.mean(axis=1), Y.std(axis=1, ddof=1)
    return Xc, Y, ls, Xg, (Fg - mean[:, None]) / sd[:, None]  # (the truth on the grid in each output's z-scored units)


def build_c4_gp(cfg, device, kronecker="auto"):
    """DataSet -> GP(outputs=[y0, y1]) -> specify_model -> build_model, as a Gumbi user declares a multi-output model
    (gumbi/regression/pymc/GP.py:711-729: RBF-ARD x output Coregion + heteroskedastic "Output_noise" Coregion :565-569)."""
    import pandas as pd

    import gumbi_amd as gmb

    Xc, Y, _, _, _ = c4_table(cfg)
    cols = [f"x{k}" for k in range(cfg["d"])]
    df = pd.DataFrame(Xc, columns=cols)
    df["y0"], df["y1"] = Y[0], Y[1]
    ds = gmb.DataSet(df, outputs=["y0", "y1"])
    gp = gmb.GP(ds, outputs=["y0", "y1"], device=device, kronecker=kronecker)
    gp.specify_model(continuous_dims=cols)
    gp.build_model(continuous_kernel=cfg["kernel"], ls_bounds=ls_bounds_for(ds, cols, LS_LOWER_Z))
    return gp


def c4_step_flops(N, P, M_rows, n_eval, n_refactor):
    """Kronecker form: P independent N x N systems per evaluation (N^3 each: Cholesky + inverse + Sigma^-1), P N^3 / 3 per
    re-factorisation at the MAP, every system solved against all M_rows prediction rows."""
    n3 = float(N) ** 3
    return P * (n_eval * n3 + n_refactor * n3 / 3.0 + float(N) ** 2 * M_rows + 4.0 * N * M_rows)


def c4_workload(cfg, local_rank, steps, warmup, map_evals, clock, stacked_beside=True):
    import torch

    from gumbi_amd import engine as E
    from gumbi_amd.regression.icm import IcmEngine

    N, d, P = cfg["N"], cfg["d"], cfg["P"]
    t_build = time.perf_counter()
    gp = build_c4_gp(cfg, local_rank)
    t_build = time.perf_counter() - t_build
    eng = gp.engine
    if not isinstance(eng, IcmEngine):
        raise RuntimeError("C4's table is aligned: HipGP(kronecker='auto') must have chosen the Kronecker engine")
    _, _, _, Xg, Fg = c4_table(cfg)
    Xs = np.vstack([np.column_stack([Xg, np.full(len(Xg), float(p))]) for p in range(P)])  # output-major, task index last
    M_rows = len(Xs)
    maxeval = map_evals if map_evals > 0 else 400
    last = {}

    def one_step(budget=maxeval):
        gp.find_MAP(maxeval=budget)
        last["mu"], last["var"] = eng.predict(Xs)  # (host arrays: 8 M (d + 3) bytes cross PCIe per step -- 1.3 MB)
        return gp.n_eval, gp.n_refactor

    for _ in range(warmup):
        one_step(min(maxeval, WARMUP_EVALS))
    torch.cuda.synchronize()
    ceiling_before = E.mfma_f64_sustained(local_rank, 1.0)
    clock.sync()
    t0 = time.perf_counter()
    counts = [one_step() for _ in range(steps)]
    clock.sync()
    elapsed = time.perf_counter() - t0
    ceiling_after = E.mfma_f64_sustained(local_rank, 1.0)
    mu, var = last["mu"], last["var"]
    finite = bool(np.all(np.isfinite(mu)) and np.all(var > 0))
    res = getattr(gp, "opt_result", None)
    Mg = len(Xg)
    quality = {
        "corr_per_output": [round(float(np.corrcoef(mu[p * Mg:(p + 1) * Mg], Fg[p])[0, 1]), 5) for p in range(P)],
        "rmse_per_output": [round(float(np.sqrt(np.mean((mu[p * Mg:(p + 1) * Mg] - Fg[p]) ** 2))), 5) for p in range(P)],
        "nlml_final": round(float(gp.nlml_trace[-1]), 3) if gp.nlml_trace else None, "n_eval": int(gp.n_eval),
        "rejected_evals": int(gp._rejected), "converged": bool(res.success) if res is not None else None,
        "optimizer_message": (res.message if isinstance(res.message, str) else res.message.decode())[:80] if res is not None else None,
        "ls": [round(float(v), 3) for v in np.atleast_1d(gp.MAP["ls_total"])], "sigma": round(float(np.asarray(gp.MAP["σ"])), 4),
        "n_hyperparameters": int(gp.model.spec.theta_size()),
    }
    theta_fit = gp._theta_fitted.copy()
    eng.set_profiling(True)
    one_step(min(maxeval, PROFILE_EVALS))
    torch.cuda.synchronize()
    tm = eng.timings()
    eng.set_profiling(False)

    def wall_ms(fn, reps=2):
        best = float("inf")
        for _ in range(reps):
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t_)
        return round(1e3 * best, 3)

    def kron_eval():
        eng.set_theta(theta_fit * (1.0 + 1e-9 * (1 + len(last))))  # (a new theta every call: nothing is reused)
        last[len(last)] = None
        eng.factorize()
        eng.nlml(grad=True)

    phases = {"specify_plus_build_model_s": round(t_build, 3), "kronecker_map_evaluation_ms": wall_ms(kron_eval, reps=3)}
    eng.set_theta(theta_fit)
    eng.factorize()
    phases["kronecker_predict_ms"] = wall_ms(lambda: eng.predict(Xs))
    phases["rates_tflops"] = {"map_evaluation": round(P * float(N) ** 3 / phases["kronecker_map_evaluation_ms"] / 1e9, 2),
                              "predict": round(P * float(N) ** 2 * M_rows / phases["kronecker_predict_ms"] / 1e9, 2)}
    spec, Xst, yst = gp.model.spec, gp.model.X, gp.model.y
    eng.close()
    gp.engine = None
    if stacked_beside:
        # the same model the way the reference hands it to PyMC: ONE stacked PN x PN system (P^2 = 4x the flops and bytes)
        st = E.Engine(local_rank)
        try:
            st.set_data(Xst, yst)
            st.set_kernel(spec)
            st.evaluate(theta_fit)
            phases["stacked_map_evaluation_ms"] = wall_ms(lambda: st.evaluate(theta_fit), reps=2)
            st.set_theta(theta_fit)
            st.factorize()
            phases["stacked_predict_ms"] = wall_ms(lambda: st.predict(Xs), reps=1)
            phases["stacked_over_kronecker"] = {"map_evaluation": round(phases["stacked_map_evaluation_ms"] / phases["kronecker_map_evaluation_ms"], 2),
                                                "predict": round(phases["stacked_predict_ms"] / phases["kronecker_predict_ms"], 2)}
        finally:
            st.close()
    return dict(elapsed=elapsed, n_evals=[c[0] for c in counts], n_refactor=[c[1] for c in counts], M=M_rows, tm=tm, phases=phases,
                finite=finite, flops=sum(c4_step_flops(N, P, M_rows, n, r) for n, r in counts), quality=quality,
                ceiling={"before_timed_region": ceiling_before, "after_timed_region": ceiling_after})


def cpu_baseline_c4(cfg, gpu_res=None, target_seconds=20.0):
    """The oracle's STACKED evaluation (what the reference's PyMC model computes: one PN x PN covariance, Cholesky, gradient) +
    prediction on a bounded sample of the C4 workload: n rows per output chosen so that it takes about ``target_seconds``."""
    from oracle import gp_oracle as O

    d, P = cfg["d"], cfg["P"]
    full = dict(cfg)
    Xg = synthetic_grid(d, cfg["res"])
    Xs = np.vstack([np.column_stack([Xg, np.full(len(Xg), float(p))]) for p in range(P)])
    spec = O.make_spec(d + 1, range(d), kind=cfg["kernel"], out_col=d, n_out=P, hetero_noise=True)

    def run(n):
        Xc, Y, ls, _, _ = c4_table(dict(full, N=n))
        X = np.vstack([np.column_stack([Xc, np.full(n, float(p))]) for p in range(P)])
        y = np.concatenate([(Y[p] - Y[p].mean()) / Y[p].std(ddof=1) for p in range(P)])
        theta = O.pack_theta(spec, ls, 1.0, 0.2, W_out=C4_W, kappa_out=C4_KAPPA, W_noise=np.array([[1.0, 0.0], [1.5, 0.0]]), kappa_noise=[1e-3, 1e-3])
        t0 = time.perf_counter()
        O.nlml_and_grad(spec, theta, X, y, dist_mode="gemm", inverse="potri")  # (dpotri, like the C3 baseline)
        O.predict(spec, theta, X, y, Xs, with_noise=True)
        return time.perf_counter() - t0

    target_seconds = float(os.environ.get("GUMBI_BENCH_CPU_SECONDS", target_seconds))
    n = min(cfg["N"], 750)
    dt = run(n)
    for _ in range(3):
        if dt >= 0.5 * target_seconds or n >= cfg["N"]:
            break
        n = min(cfg["N"], max(n + 64, int(n * (target_seconds / max(dt, 1e-3)) ** (1.0 / 3.0)) // 64 * 64))
        dt = run(n)
    # ONE flop basis for host and GPU: the problem's algorithmic flops in the Kronecker form the GPU line counts
    # (P N^3 per evaluation, P N^2 M_rows per prediction); what the host actually executes -- the stacked (PN)^3 system, like the
    # reference's PyMC model -- is P^2 times that and is reported beside it, not as `value`
    flops = c4_step_flops(n, P, len(Xs), 1, 0)
    stacked = float(P * n) ** 3 + float(P * n) ** 2 * len(Xs)
    cores, blas = host_blas()
    out = {"value": round(flops / dt / 1e9, 2), "unit": "GFLOP/s", "cores": int(cores), "blas": blas, "kind": "port",
           "sample": f"1 stacked MAP objective+gradient evaluation + predict(M={len(Xs)}) at N={n} rows x {P} outputs ({P * n} x {P * n} "
                     f"covariance), d={d}, {cfg['kernel']} x Coregion + output noise (numpy/LAPACK oracle, {dt:.1f} s)",
           "seconds": round(dt, 2),
           "executed_stacked_system_gflops": round(stacked / dt / 1e9, 2),
           "note": "value counts the SAME algorithmic flops as the GPU line (Kronecker form: P N^3 per evaluation + P N^2 M per prediction) "
                   "over the host's seconds, so value ratios are wall-time ratios at equal size; the host executes the stacked (PN)^3 "
                   "system (executed_stacked_system_gflops)"}
    if gpu_res is not None and gpu_res.get("phases", {}).get("kronecker_map_evaluation_ms"):
        ph = gpu_res["phases"]
        gpu_s = (ph["kronecker_map_evaluation_ms"] + ph["kronecker_predict_ms"]) / 1e3
        out["gpu_seconds_same_step_at_full_size"] = round(gpu_s, 4)
        out["host_seconds_extrapolated_to_full_size"] = round(dt * (cfg["N"] / n) ** 3, 1)
        out["wall_time_ratio_note"] = ("host: the sample's seconds x (N / n)^3 (an extrapolation, labelled); GPU: one Kronecker evaluation + one "
                                       "prediction at the full N, measured in this run")
    return out


def c4_side_section(local_rank, clock):
    """C4 as a side figure of the default run: one timed MAP fit + prediction through HipGP(kronecker='auto') after one capped
    warm-up step, the stacked engine's evaluation beside it, the roofline of the P N x N evaluation launches."""
    cfg = CONFIGS["c4"]
    res = c4_workload(cfg, local_rank, 1, 1, C4_SIDE_EVALS, clock)
    r = roofline_block(res["tm"], "c4")
    return {
        "workload": cfg["label"],
        "step": f"find_MAP(maxeval={C4_SIDE_EVALS}) + grid prediction -- CAPPED: the fit to convergence is ~220 evaluations = 56 s "
                "(python bench.py --config c4; profiles/r05_bench_c4.json)",
        "ms_per_step": round(1e3 * res["elapsed"], 3), "fit_predict_seconds": round(res["elapsed"], 4),
        "map_evals": res["n_evals"], "seconds_per_map_evaluation": round(res["elapsed"] / max(sum(res["n_evals"]), 1), 4),
        "value": round(res["flops"] / res["elapsed"] / 1e9, 2), "unit": "GFLOP/s", "results_finite": res["finite"],
        "fit_quality": res["quality"], "phases": res["phases"],
        "roofline": {k: r[k] for k in ("kernel", "achieved", "peak", "frac", "achieved_on_unpadded_N3", "frac_on_unpadded_N3", "launches", "avg_launch_ms") if k in r},
    }


# -----------------------------------------------------------------------------------------------------
# workload B: ONE GP over `world` GPUs (C5): fixed hyper-parameters, or a distributed find_MAP(maxeval=k)
# -----------------------------------------------------------------------------------------------------
COMM_KEYS = ("dist_world", "dist_chol_collectives", "dist_chol_comm_bytes", "dist_chol_comm_ms", "dist_chol_comm_exposed_ms",
             "dist_chol_main_wait_ms", "dist_chol_bulk_wait_ms", "dist_grad_collectives", "dist_grad_comm_bytes",
             "dist_grad_comm_ms", "dist_grad_comm_exposed_ms", "dist_lockstep_repairs")


def comm_block(tm, eng, transport, clock):
    """Communication of the LAST factorisation and the LAST gradient evaluation of the run (gmb_timings.dist_*): what
    rank 0 saw, and the maximum over the ranks of the exposed part."""
    comm = getattr(eng, "comm", None)
    ranks = None
    if comm is not None and hasattr(comm, "ranks"):
        try:
            ranks = comm.ranks
        except Exception:
            ranks = None
    chol_exp, grad_exp, chol_ms, grad_ms = clock.max_over_ranks([tm["dist_chol_comm_exposed_ms"], tm["dist_grad_comm_exposed_ms"],
                                                                  tm["dist_chol_comm_ms"], tm["dist_grad_comm_ms"]])
    return {
        "transport": transport, "rccl_ranks": ranks, "world": int(tm["dist_world"]),
        # RCCL runs ONE communicator's collectives in issue order whatever their streams: the transport keeps a second one (split off
        # the first) for the collectives of the communication stream -- None for other transports
        "rccl_two_communicators": bool(comm.two_communicators) if hasattr(comm, "two_communicators") else None,
        "comm_ms_total": round(tm["dist_chol_comm_ms"] + tm["dist_grad_comm_ms"], 3),
        "comm_ms_exposed": round(tm["dist_chol_comm_exposed_ms"] + tm["dist_grad_comm_exposed_ms"], 3),
        "per": "one factorisation + one gradient evaluation (the last of the run), rank 0; *_max = maximum over the ranks",
        # ranks whose redundantly computed (failure index, log-det, |v|^2) differed from rank 0's and were overwritten: 0 expected
        "lockstep_repairs": int(tm["dist_lockstep_repairs"]),
        "factorize": {"collectives": int(tm["dist_chol_collectives"]), "received_GB": round(tm["dist_chol_comm_bytes"] / 1e9, 3),
                      "comm_ms": round(tm["dist_chol_comm_ms"], 3), "comm_ms_exposed": round(tm["dist_chol_comm_exposed_ms"], 3),
                      "comm_ms_max": round(chol_ms, 3), "comm_ms_exposed_max": round(chol_exp, 3),
                      "main_stream_wait_at_joins_ms": round(tm["dist_chol_main_wait_ms"], 3),
                      "bulk_stream_wait_at_forks_ms": round(tm["dist_chol_bulk_wait_ms"], 3),
                      "chol_ms": round(tm["chol_ms"], 3)},
        "gradient": {"collectives": int(tm["dist_grad_collectives"]), "received_GB": round(tm["dist_grad_comm_bytes"] / 1e9, 3),
                     "comm_ms": round(tm["dist_grad_comm_ms"], 3), "comm_ms_exposed": round(tm["dist_grad_comm_exposed_ms"], 3),
                     "comm_ms_max": round(grad_ms, 3), "comm_ms_exposed_max": round(grad_exp, 3), "grad_ms": round(tm["grad_ms"], 3)},
    }


def one_gp_workload(cfg, world, local_rank, dist, steps, warmup, clock, map_evals=0, capacity=False):
    import torch

    from gumbi_amd import engine as E

    N, d = cfg["N"], cfg["d"]
    Xs = synthetic_grid(d, cfg["res"])
    M = len(Xs)
    part = {"map_eval": 0.0, "fit_fixed_theta": 0.0, "predict": 0.0, "find_map": 0.0}
    last = {}
    gp = None
    if map_evals > 0:
        # the front end, as a user would call it: GP(distributed=True) on every rank, same declared model as C3's
        gp = build_gp(cfg, device=local_rank, distributed=(True if world > 1 else None))
        eng = gp.engine
        transport = eng.comm.kind if world > 1 else "single engine"
        counts = []

        def one_step(record=False):
            t0 = time.perf_counter()
            gp.find_MAP(maxeval=map_evals)
            t1 = time.perf_counter()
            mu, var = eng.predict(Xs)
            t2 = time.perf_counter()
            if record:
                part["find_map"] += t1 - t0
                part["predict"] += t2 - t1
                counts.append((gp.n_eval, gp.n_refactor))
            last.update(mu=mu, var=var, grad=np.zeros(1), nl=gp.nlml_trace[-1] if gp.nlml_trace else float("nan"))
    else:
        X, y, ls = synthetic_table(N, d)
        theta = np.concatenate([ls, [1.0, 0.2]])
        spec = E.KernelSpec(D=d, idx_cont=list(range(d)), kind=cfg["kernel"])
        if world == 1:
            eng = E.Engine(local_rank)
            transport = "single engine"
        else:
            from gumbi_amd.distributed import DistributedEngine

            eng = DistributedEngine(local_rank, capacity=capacity)  # capacity: no rank holds the factor (csrc/dist_capacity.hpp)
            transport = eng.comm.kind
        eng.set_data(X, y)
        eng.set_kernel(spec)
        eng.set_theta(theta)
        if world == 1 and warmup == 0:
            # a run without a warm-up step (the C5 side figure of the default line) would time ~160 GB of hipMalloc inside its
            # one step (measured: map_eval 20.1 s instead of 14.0): the buffers are reserved first, like the table is copied first
            eng.reserve(gradient=True, M=M)

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
    # (what the TIMED fit found -- before the profiled pass below runs a two-evaluation fit of its own on the same object)
    quality_timed = fit_quality(gp, last["mu"], cfg) if map_evals > 0 else None
    eng.set_profiling(True)
    if map_evals > 0:
        gp.find_MAP(maxeval=min(map_evals, 2))
        eng.predict(Xs)
    else:
        one_step()
    torch.cuda.synchronize()
    tm = eng.timings()
    eng.set_profiling(False)
    comm = comm_block(tm, eng, transport, clock) if world > 1 else None
    n3 = float(N) ** 3
    if map_evals > 0:
        per = clock.max_over_ranks([part["find_map"] / steps, part["predict"] / steps])
        flops = sum(step_flops(N, M, n, r) for n, r in counts)
        phases = {"find_map_s": round(per[0], 4), "predict_s": round(per[1], 4), "map_evals_per_step": [c[0] for c in counts],
                  "refactorizations_at_the_map_per_step": [c[1] for c in counts],
                  "fit_quality_after_k_evals": quality_timed, "nlml": float(last["nl"])}
    else:
        per = clock.max_over_ranks([part["map_eval"] / steps, part["fit_fixed_theta"] / steps, part["predict"] / steps])
        flops = steps * step_flops(N, M, 1)
        phases = {
            "map_eval_s": round(per[0], 4), "fit_fixed_theta_s": round(per[1], 4), "predict_s": round(per[2], 4),
            "rates_tflops_whole_job": {"map_eval": round(n3 / per[0] / 1e12, 2), "fit_fixed_theta": round(n3 / 3.0 / per[1] / 1e12, 2),
                                       "predict": round(float(N) ** 2 * M / per[2] / 1e12, 2)},
            "nlml": float(last["nl"]),
        }
    phases["profiled_last_step_ms_rank0"] = {k: round(tm[k], 3) for k in ("kbuild_ms", "chol_ms", "chol_gemm_ms", "grad_ms",
                                                                         "grad_gemm_ms", "predict_ms")}
    finite = bool(np.all(np.isfinite(last["mu"])) and np.all(last["var"] > 0) and np.all(np.isfinite(last["grad"])))
    eng.close()
    if gp is not None:
        gp.engine = None
    return dict(elapsed=elapsed, M=M, tm=tm, phases=phases, finite=finite, transport=transport, flops=flops, comm=comm,
                last=dict(nlml=float(last["nl"]), mean=np.asarray(last["mu"], dtype=float).copy(), var=np.asarray(last["var"], dtype=float).copy()))


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


def fail_line(args, n_gpus, message, do_print=True):
    """The contract's JSON line for a run that could not be made: value null, `error` says why; the caller exits non-zero."""
    out = {"metric": "fit+predict achieved GFLOP/s (fp64 exact GP)", "value": None, "unit": "GFLOP/s", "n_gpus": int(n_gpus),
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": {"workload": CONFIGS[args.config or "c5"]["label"]},
           "error": {"error": str(message)[:400]}}
    if do_print:
        print(json.dumps(out, ensure_ascii=False), flush=True)
    sys.stderr.write(f"[bench] {message}\n")
    sys.stderr.flush()
    return out


def enough_devices(n_ranks):
    """(ok, devices): one GPU per rank, except under the tests' hook that puts every rank on GPU 0 (gloo only)."""
    from gumbi_amd import engine

    have = engine.device_count()
    if os.environ.get("GUMBI_BENCH_SINGLE_DEVICE") == "1":
        return have >= 1 and os.environ.get("GUMBI_BENCH_BACKEND", "nccl") != "nccl", have
    return have >= n_ranks, have


def self_launch(args):
    """``python bench.py --gpus N`` started WITHOUT a launcher (no WORLD_SIZE in the environment): re-run this same command line
    as N ranks, one per GPU, under ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1``
    -- exactly the form the driver uses for N > 1 -- and hand its exit code on.  Fewer than N GPUs on the node is an ERROR
    (JSON line with ``error``, exit code 2), never a silent one-GPU run."""
    import socket
    import subprocess

    ok, have = enough_devices(args.gpus)
    if not ok:
        fail_line(args, args.gpus, f"--gpus {args.gpus} needs {args.gpus} MI355X devices (one rank per GPU over RCCL); this node shows {have}")
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    sys.stderr.write("[bench] --gpus %d without a launcher: %s\n" % (args.gpus, " ".join(cmd)))
    sys.stderr.flush()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), GUMBI_BENCH_SELF_LAUNCHED="1")
    return subprocess.call(cmd, env=env, cwd=str(ROOT))


def c5_fit_predict_block(c5_side):
    """fit+predict at the north star's size (N = 100k) on ONE GPU for the default line: an estimate from THIS run's measured
    evaluation (n_eval of the recorded fit x map_eval_s + prediction), labelled, beside the builder-run record of a whole
    find_MAP to convergence (profiles/*_bench_c5_map.json, `python bench.py --config c5 --map-evals 200 --steps 1 --warmup 0`)."""
    import glob

    out = {}
    rec = None
    files = sorted(glob.glob(os.path.join(str(ROOT), "profiles", "*_bench_c5_map.json")))
    if files:
        try:
            with open(files[-1]) as fh:
                rec = json.load(fh)
            rec = rec.get("parsed", rec)
            q = (rec.get("phases") or {}).get("fit_quality_after_k_evals") or {}
            out["measured_record"] = {"source": os.path.relpath(files[-1], str(ROOT)), "fit_predict_seconds": rec.get("fit_predict_seconds"),
                                      "n_eval": q.get("n_eval"), "converged": q.get("converged"), "corr": q.get("corr"),
                                      "what": "find_MAP to convergence (ls_bounds as in the C3 headline) + 10^4-point grid prediction, one step, "
                                              "one MI355X; a builder-run record, not this run"}
        except Exception as err:  # noqa: BLE001
            out["measured_record"] = {"error": f"{type(err).__name__}: {err}"[:200]}
    ph = (c5_side or {}).get("phases") or {}
    n_eval = (out.get("measured_record") or {}).get("n_eval")
    if ph.get("map_eval_s") and n_eval:
        out["c5_fit_predict_seconds_estimate"] = round(n_eval * ph["map_eval_s"] + ph.get("predict_s", 0.0), 1)
        out["estimate_note"] = (f"ESTIMATE ~ n_eval x map_eval_s + predict_s: {n_eval} evaluations (the recorded fit's count) x {ph['map_eval_s']} s "
                                f"(one evaluation at N = 100k, measured in this run) + {ph.get('predict_s')} s")
    return out


def match_against_one_gpu(dist_last, one_last, map_evals):
    """The first multi-GPU line must validate itself: NLML, grid mean and variance of the N-rank step
    against the same step on one GPU (rank 0 runs it after the timed region)."""
    if not dist_last or not one_last:
        return {"results_match_one_gpu": None, "results_match_note": "no one-GPU run to compare with"}
    tol_nl, tol_mu, tol_var = (1e-10, 1e-8, 1e-9) if map_evals <= 0 else (1e-6, 1e-5, 1e-6)
    nl_rel = abs(dist_last["nlml"] - one_last["nlml"]) / max(abs(one_last["nlml"]), 1e-300)
    scale = max(float(np.max(np.abs(one_last["mean"]))), 1e-300)
    mu_rel = float(np.max(np.abs(dist_last["mean"] - one_last["mean"]))) / scale
    var_abs = float(np.max(np.abs(dist_last["var"] - one_last["var"])))
    ok = bool(np.isfinite(nl_rel) and nl_rel <= tol_nl and mu_rel <= tol_mu and var_abs <= tol_var)
    return {"nlml_rel_diff_vs_one_gpu": float(nl_rel), "mean_max_rel_diff_vs_one_gpu": mu_rel, "var_max_abs_diff_vs_one_gpu": var_abs,
            "results_match_one_gpu": ok,
            "results_match_note": f"NLML rel {nl_rel:.2e} (<= {tol_nl:g}), grid mean max rel {mu_rel:.2e} (<= {tol_mu:g}), variance max abs "
                                  f"{var_abs:.2e} (<= {tol_var:g})" + ("; two L-BFGS-B trajectories: looser tolerances" if map_evals > 0 else "")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="default: c3 on one GPU, c5 (ONE GP over all ranks) on several")
    ap.add_argument("--map-evals", type=int, default=0,
                    help="one GPU: cap on L-BFGS objective evaluations per fit (0 = run to convergence, cap 200); c5 / several "
                         "GPUs: 0 = fixed hyper-parameters, k > 0 = a (distributed) find_MAP(maxeval=k) per step")
    ap.add_argument("--capacity", action="store_true",
                    help="several GPUs, fixed hyper-parameters: the multi-GPU driver's CAPACITY mode (a rank keeps its own block rows + panel "
                         "buffers; L is streamed again for the gradient and the prediction) instead of the replicated factor")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-fit", default=None, choices=["c1", "c2"],
                    help="internal (cpu_baseline): the config's whole step on the HOST -- HipGP's host code over the numpy/LAPACK oracle")
    args = ap.parse_args()
    if args.cpu_fit:
        cpu_fit_subprocess(args.cpu_fit)
        return

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))  # `python bench.py --gpus N` by itself: N ranks under torch.distributed.run
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    config_name = args.config or ("c3" if world == 1 else "c5")
    cfg = CONFIGS[config_name]
    if world != args.gpus:
        # a launcher that started another number of ranks than --gpus names: never print a line whose n_gpus is not what was asked
        fail_line(args, args.gpus, f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", rank == 0)
        sys.exit(2)
    if world > 1 and config_name != "c5":
        fail_line(args, world, "several GPUs run the one-GP workload: --config c5", rank == 0)
        sys.exit(2)

    dist = None
    if world > 1:
        ok, have = enough_devices(world)  # before anything touches a device that may not exist
        if not ok:
            fail_line(args, world, f"{world} ranks need {world} MI355X devices (one rank per GPU over RCCL); this node shows {have}", rank == 0)
            sys.exit(2)
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
    budget = Budget()
    healthy = True

    if config_name == "c5":
        def section():
            torch.cuda.set_device(local_rank)  # the current device is per-thread state
            return one_gp_workload(cfg, world, local_rank, dist, args.steps, args.warmup, clock, args.map_evals, args.capacity and world > 1)

        res, healthy = run_with_deadline(section, float(os.environ.get("GUMBI_BENCH_DEADLINE", "1500")))
    elif config_name == "c4":
        res = c4_workload(cfg, local_rank, args.steps, args.warmup, args.map_evals, clock)
    else:
        res = map_fit_workload(cfg, config_name, local_rank, args.steps, args.warmup, args.map_evals, clock)

    base_last = {}

    def c5_on_one_gpu(steps, warmup):
        torch.cuda.set_device(local_rank)
        r = one_gp_workload(CONFIGS["c5"], 1, local_rank, None, steps, warmup, Clock(None, dev), args.map_evals if config_name == "c5" else 0)
        base_last.update(r["last"])
        return {"workload": CONFIGS["c5"]["label"], "note": "the multi-GPU runs' problem on ONE GPU: their strong-scaling base "
                                                           "(python bench.py --config c5)",
                "value": round(r["flops"] / r["elapsed"] / 1e9, 2), "unit": "GFLOP/s", "steps": steps, "warmup": warmup,
                "ms_per_step": round(1e3 * r["elapsed"] / steps, 3), "phases": r["phases"], "results_finite": r["finite"],
                "trailing_update_tflops": roofline_block(r["tm"], "c5")["achieved"]}

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
            if config_name == "c4":
                metric = "fit+predict achieved GFLOP/s (fp64 exact 2-output ICM GP, Kronecker form: MAP fit to convergence + grid prediction)"
            elif config_name != "c5":
                metric = "fit+predict achieved GFLOP/s (fp64 exact GP: MAP fit to convergence + grid prediction)"
            elif args.map_evals > 0:
                metric = f"fit+predict achieved GFLOP/s (fp64 exact GP: find_MAP(maxeval={args.map_evals}) + grid prediction)"
            else:
                metric = "fit+predict achieved GFLOP/s (fp64 exact GP: one MAP evaluation + fit at fixed theta + grid prediction)"
            out = {
                "metric": metric,
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
                # what "parity green" means for these numbers: the HIP path agrees with the numpy restatement
                # of PyMC's formulas to ~1e-13 (tests: <= 1e-8 mean, 1e-9 variance, 1e-10 NLML); the restatement itself meets PyMC
                # only through the outputs the reference's notebooks print
                "parity_pin": "PyMC-5 notebook outputs (Multioutput_Regression.ipynb), 1e-3 means / 5 % variances; 1e-8 vs the restatement "
                              "(oracle/gp_oracle.py; scikit-learn agrees with it to 1e-12 on the stationary kernels)",
            }
            if config_name == "c4":
                out["config"]["outputs"] = cfg["P"]
                out["config"]["form"] = ("Kronecker / ICM: B (x) K + D (x) I as P independent N x N systems (gumbi_amd/regression/icm.py), "
                                         "chosen by HipGP(kronecker='auto'); flops counted as P N^3 per evaluation")
            if "n_evals" in res:
                out["config"]["map_evals_per_step"] = res["n_evals"]
                out["config"]["refactorizations_at_the_map_per_step"] = res["n_refactor"]
                out["config"]["ls_bounds"] = f"lower = {LS_LOWER_Z} standardized units on every input (make_deltas_parray), upper = default"
                out["config"]["warmup_step"] = f"same code path, evaluation budget {WARMUP_EVALS} (untimed)"
                out["fit_quality"] = res["quality"]
                out["seconds_per_map_evaluation"] = round(elapsed / max(sum(res["n_evals"]), 1), 4)
            if config_name == "c5":
                out["config"]["step"] = (f"a distributed find_MAP(maxeval={args.map_evals}) + grid prediction" if args.map_evals > 0 else
                                         "ONE MAP objective + gradient evaluation + fit at fixed theta (re-factorisation) + grid prediction")
                out["config"]["not_like_for_like_with"] = ("the one-GPU default line (config c3: a MAP fit to convergence at N = 50k); the one-GPU "
                                                           "point of THIS curve is strong_scaling_base_gflops / `python bench.py --config c5`")
            if world > 1:
                out["config"]["factor_storage"] = ("capacity mode: own block rows + panel buffers per rank (gmb_dist_set_mode 1)" if args.capacity and not args.map_evals
                                                   else "replicated: every rank ends with the complete factor")
            if res.get("comm") is not None:
                out["comm"] = res["comm"]
                out["transport"], out["rccl_ranks"] = res["comm"]["transport"], res["comm"]["rccl_ranks"]
                out["comm_ms_total"], out["comm_ms_exposed"] = res["comm"]["comm_ms_total"], res["comm"]["comm_ms_exposed"]
                out["launcher"] = "bench.py itself (--gpus N without WORLD_SIZE)" if os.environ.get("GUMBI_BENCH_SELF_LAUNCHED") == "1" else "external (WORLD_SIZE set)"
                if out["transport"] == "rccl" and out["rccl_ranks"] != world:
                    # the library's communicator does not span the ranks that were launched: whatever was timed is not an N-GPU run
                    out["error"] = {"error": f"RCCL communicator reports {out['rccl_ranks']} ranks, {world} were launched"}
            # the two roofline fractions side by side: the dominant kernel's (a subset of the launches) and the
            # whole timed step's on its algorithmic flops
            out["roofline_frac_dominant_kernel"] = out["roofline"]["frac"]
            out["roofline_frac_whole_step"] = round(out["value"] / 1e3 / FP64_MFMA_PEAK_TFLOPS, 4)
            if "ceiling" in res:
                cb, ca = res["ceiling"]["before_timed_region"], res["ceiling"]["after_timed_region"]
                out["roofline"]["mfma_only_ceiling"] = {
                    "kernel": "register-only v_mfma_f64_16x16x4_f64 loop with the GEMM's register pattern, >= 1 s back to back",
                    "before_timed_region": {k: round(v, 2) if isinstance(v, float) else v for k, v in cb.items()},
                    "after_timed_region": {k: round(v, 2) if isinstance(v, float) else v for k, v in ca.items()},
                    "datasheet_tflops_at_2400_mhz": FP64_MFMA_PEAK_TFLOPS}
                out["roofline"]["mfma_only_microbench_tflops"] = round(max(cb["tflops_mean"], ca["tflops_mean"]), 2)
                out["roofline"]["achieved_le_ceiling"] = bool(out["roofline"]["achieved"] <= out["roofline"]["mfma_only_microbench_tflops"])

    # side sections of the one-GPU run (outside the timed steps), most important first, while the time budget lasts;
    # `section_seconds` says what each one took
    sections = {}

    def timed_section(name, fn):
        t_ = time.perf_counter()
        try:
            return fn()
        finally:
            sections[name] = round(time.perf_counter() - t_, 1)

    if world == 1 and out is not None and "error" not in out and config_name != "c5":
        if not args.no_cpu_baseline and os.environ.get("GUMBI_BENCH_NO_CPU") != "1":
            out["cpu_baseline"] = timed_section("cpu_baseline", lambda: cpu_baseline_c4(cfg, res) if config_name == "c4" else cpu_baseline(cfg))
        if config_name == "c3" and os.environ.get("GUMBI_BENCH_NO_C2") != "1":
            # the size class Gumbi users live in (BASELINE.json configs[1], N = 10k): a converging fit + prediction, with the
            # phases of one evaluation -- its factorisation is ONE launch of the persistent tile kernel (csrc/chol_tiles.hpp)
            est = 20.0
            if budget.allows(est):
                try:
                    out["c2_single_gpu"] = timed_section("c2_single_gpu", lambda: c2_side_section(local_rank, clock))
                except Exception as err:
                    out["c2_single_gpu"] = {"error": f"{type(err).__name__}: {err}"[:300]}
            else:
                out["c2_single_gpu"] = budget.skipped(est)
        if config_name in ("c2", "c3") and os.environ.get("GUMBI_BENCH_NO_C4") != "1":
            # BASELINE.json configs[3]: the two-output coregionalised GP through the Kronecker path, stacked engine beside it
            est = 35.0
            if budget.allows(est):
                try:
                    out["c4_single_gpu"] = timed_section("c4_single_gpu", lambda: c4_side_section(local_rank, clock))
                except Exception as err:
                    out["c4_single_gpu"] = {"error": f"{type(err).__name__}: {err}"[:300]}
            else:
                out["c4_single_gpu"] = budget.skipped(est)
        if os.environ.get("GUMBI_BENCH_NO_DIST") != "1" and config_name != "c4":
            # (no warm-up step: the C3 run before it has just run the very same kernels for minutes -- VERDICT r05 item 3)
            est = 45.0
            if budget.allows(est):
                out["c5_single_gpu"], _ = timed_section("c5_single_gpu", lambda: run_with_deadline(lambda: c5_on_one_gpu(1, 0), 600.0))
                out["strong_scaling_base_gflops"] = out["c5_single_gpu"].get("value")
                out["c5_fit_predict"] = c5_fit_predict_block(out["c5_single_gpu"])
            else:
                out["c5_single_gpu"] = budget.skipped(est)
                out["strong_scaling_base_gflops"] = None
        if os.environ.get("GUMBI_BENCH_NO_E2E") != "1" and config_name != "c4":
            # the user-level call sequence, declaration and host transfers inside the clock (ahead of the host's sections)
            est = 1.1 * out["fit_predict_seconds"] + 8.0
            if budget.allows(est):
                try:
                    out["end_to_end"] = timed_section("end_to_end", lambda: end_to_end_fit(cfg, local_rank, args.map_evals))
                except Exception as err:
                    out["end_to_end"] = {"error": f"{type(err).__name__}: {err}"[:300]}
            else:
                out["end_to_end"] = budget.skipped(est)
        if "cpu_baseline" in out and os.environ.get("GUMBI_BENCH_NO_CPU_CONFIG_SIZE") != "1" and config_name != "c4":
            # the host at the configs' own sizes (minutes of host time), beside fit_predict_seconds
            gpu_c2 = out.get("fit_quality") if config_name == "c2" else (out.get("c2_single_gpu") or {}).get("fit_quality")
            try:
                out["cpu_baseline"].update(timed_section("cpu_config_size", lambda: cpu_config_size_sections(budget, gpu_c2)))
            except Exception as err:  # noqa: BLE001
                out["cpu_baseline"]["config_size_error"] = f"{type(err).__name__}: {err}"[:300]
            gpu_c2_s = out.get("fit_predict_seconds") if config_name == "c2" else (out.get("c2_single_gpu") or {}).get("fit_predict_seconds")
            out["cpu_baseline"]["gpu_c2_fit_predict_seconds"] = gpu_c2_s
            if config_name == "c3":
                out["cpu_baseline"]["gpu_c3_cholesky_seconds"] = round(out["phases"].get("factorize_ms", 0.0) / 1e3, 4) or None
        if os.environ.get("GUMBI_BENCH_NO_DEFAULT_START") != "1" and config_name != "c4":
            est = 10.0 + 12.0 * out["seconds_per_map_evaluation"]
            if budget.allows(est):
                try:
                    out["default_start"] = timed_section("default_start", lambda: default_start_fit(cfg, local_rank))
                except Exception as err:
                    out["default_start"] = {"error": f"{type(err).__name__}: {err}"[:300]}
            else:
                out["default_start"] = budget.skipped(est)
    # several GPUs: the same step on ONE GPU of this node, by rank 0 (the other ranks wait at the final barrier) -- the curve's
    # one-GPU point AND the check of this line: the N-rank run must have computed what one GPU computes
    if world > 1 and out is not None and "error" not in out and healthy and os.environ.get("GUMBI_BENCH_NO_BASE") != "1":
        base, _ = run_with_deadline(lambda: c5_on_one_gpu(1, 1), 500.0)
        out["strong_scaling_base"] = base
        out["strong_scaling_base_gflops"] = base.get("value")
        if base.get("value"):
            out["speedup_over_one_gpu"] = round(out["value"] / base["value"], 3)
        if rank == 0:
            out.update(match_against_one_gpu(res.get("last"), base_last, args.map_evals))
            if out.get("results_match_one_gpu") is False:
                out["error"] = {"error": "the N-rank run and the one-GPU run of the same step disagree: " + out["results_match_note"]}
    elif world == 1 and out is not None and "error" not in out and config_name == "c5":
        out["strong_scaling_base_gflops"] = out["value"]  # this IS the one-GPU point
    if out is not None and "error" not in out:
        # ONE key under which every line of `--gpus 1 / 2 / 4 / 8` carries its point of the SAME curve (C5, the one-GP workload):
        # the default one-GPU line's headline is C3 (a different problem), its C5 point is the side figure
        c5_step = ("find_MAP(maxeval=%d) + grid prediction" % args.map_evals if (config_name == "c5" and args.map_evals > 0) else
                   "one MAP objective + gradient evaluation + fit at fixed theta + grid prediction")
        gf = out["value"] if config_name == "c5" else out.get("strong_scaling_base_gflops")
        out["scale_point"] = {"config": "c5", "n_gpus": world, "gflops": gf, "workload": CONFIGS["c5"]["label"], "step": c5_step,
                              "note": None if gf is not None else "the C5 side figure was skipped (time budget / GUMBI_BENCH_NO_DIST)"}
    if sections:
        out["section_seconds"] = sections
    if out is not None:
        out["bench_wall_s"] = round(budget.used(), 1)
    if rank == 0 and out is not None:
        print(json.dumps(out, ensure_ascii=False), flush=True)
    failed = out is not None and "error" in out
    if dist is not None:
        def leave():
            torch.cuda.set_device(local_rank)
            dist.barrier()
            dist.destroy_process_group()

        if healthy:
            _, healthy = run_with_deadline(leave, 700.0)
        if not healthy:
            sys.stderr.write(f"[bench rank {rank}] leaving without the final barrier\n")
            sys.stderr.flush()
            os._exit(1)  # a run whose ranks hung or died is not a healthy run, whatever was printed
    if failed or not healthy:
        sys.exit(1)


if __name__ == "__main__":
    main()
