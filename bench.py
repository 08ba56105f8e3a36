#!/usr/bin/env python
"""Headline benchmark: GP fit + predict on synthetic N x d fp64 tables (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c1] [--map-evals E]

A *step* is one pass of the hot path the reference reaches through ``gp.fit()`` +
``gp.predict_grid()``, with the inputs already resident in HBM:

    find_MAP  -- L-BFGS-B on -(log-lik + log-priors + log-Jacobians); every objective / gradient
                 evaluation = covariance build -> Cholesky -> v = L^-1 y, log-det ->
                 L^-1, Sigma^-1, fused trace reductions                  (all in libgumbi_hip.so)
    refactor at the MAP, predict mean / variance on the M-point grid     (libgumbi_hip.so)

The default workload is BASELINE.json configs[1]: N = 10k, d = 4, RBF-ARD, fp64, M = 10^4 grid
(100 x 100 over dims 0,1; other dims pinned at 0).  ``value`` = algorithmic GFLOP/s of the whole
step; ``ms_per_step`` is the fit + predict wall time.  One JSON line is printed by rank 0.

With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU over RCCL) the timed
workload shards by independent GPs: every rank runs the same MAP fit + prediction on its own GPU
(the cross-validation / per-category refit pattern, SURVEY.md section 8e last row) -- weak
scaling, no data-path collective; the barrier / max-over-ranks timing contract is kept.
In the same invocation, OUTSIDE the timed steps, the block-cyclic multi-GPU Cholesky
(gumbi_amd/distributed.py: diagonal-block broadcast + panel all-gather on RCCL) factors ONE large
covariance matrix (N = 40,960, d = 8, RBF-ARD, fixed hyper-parameters) across all ranks and
predicts a 10^4 grid sharded over the ranks; its wall time and rate are reported under
"distributed" in the same JSON line (at --gpus 1 the same problem runs on the single-GPU engine,
which gives the strong-scaling baseline).  GUMBI_BENCH_NO_DIST=1 skips that section.
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
}


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


def build_gp(cfg, device):
    """Front-end objects exactly as a Gumbi user builds them (DataSet -> GP -> build_model)."""
    import pandas as pd

    import gumbi_amd as gmb

    X, y, _ = synthetic_table(cfg["N"], cfg["d"])
    cols = [f"x{k}" for k in range(cfg["d"])]
    df = pd.DataFrame(X, columns=cols)
    df["y"] = y
    ds = gmb.DataSet(df, outputs=["y"])
    gp = gmb.GP(ds, outputs=["y"], device=device)
    gp.specify_model(continuous_dims=cols)
    gp.build_model(continuous_kernel=cfg["kernel"])
    return gp


def step_flops(N, M, n_eval):
    """Algorithmic flops of one step: per evaluation N^3/3 (Cholesky) +
    2N^3/3 (inverse + Sigma^-1); final Cholesky N^3/3; predict N^2 M + 4 N M."""
    n3 = float(N) ** 3
    return n_eval * n3 + n3 / 3.0 + float(N) ** 2 * M + 4.0 * N * M


def pmc_traffic(config):
    """HBM bytes per GEMM launch from the committed rocprofv3 PMC summary of this same bench command
    (profiles/*_pmc_bench_<config>_summary.csv, made by tools/gpu_pmc_bench.sh: separate --pmc
    passes for FETCH_SIZE and WRITE_SIZE; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950, WRITE_SIZE as counted).  None when no summary is committed."""
    import csv
    import glob

    files = sorted(glob.glob(os.path.join(str(ROOT), "profiles", f"*_pmc_bench_{config}_summary.csv")))
    if not files:
        return None
    launches, gbytes = 0, 0.0
    with open(files[-1]) as fh:
        for row in csv.DictReader(fh):
            if "gemm_f64" in row["Kernel"]:
                launches += int(row["Launches"])
                gbytes += float(row["FetchGB(x2 corrected)"]) + float(row["WriteGB(raw)"])
    if launches == 0:
        return None
    return {"bytes_per_launch": gbytes * 1e9 / launches, "source": os.path.relpath(files[-1], str(ROOT)), "launches": launches}


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


DIST_N, DIST_D = int(os.environ.get("GUMBI_BENCH_DIST_N", "40960")), 8  # the env override is for the tests


def distributed_section(world, local_rank, dist):
    """Fixed-theta fit (K-build + Cholesky + v + NLML), a sharded grid prediction and one MAP
    objective+gradient evaluation of ONE N = 40,960 GP over all ranks; max-over-ranks wall time."""
    import torch

    from gumbi_amd import engine as E

    N, d = DIST_N, DIST_D
    X, y, ls = synthetic_table(N, d)
    Xs = synthetic_grid(d, 100)
    theta = np.concatenate([ls, [1.0, 0.2]])
    spec = E.KernelSpec(D=d, idx_cont=list(range(d)), kind="ExpQuad")
    if world == 1:
        eng = E.Engine(local_rank)
        predict = lambda: eng.predict(Xs)  # noqa: E731
    else:
        from gumbi_amd.distributed import DistributedEngine

        eng = DistributedEngine(local_rank)
        predict = lambda: eng.predict(Xs)  # noqa: E731
    eng.set_data(X, y)
    eng.set_kernel(spec)
    eng.set_theta(theta)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    eng.factorize()  # warm-up (allocations, code objects, RCCL channels)
    eng.nlml(grad=True)  # ... including the gradient's N^2 workspace
    sync()
    t0 = time.perf_counter()
    eng.factorize()
    nlml = eng.nlml()
    sync()
    t1 = time.perf_counter()
    mu, var = predict()
    sync()
    t2 = time.perf_counter()
    # one MAP objective + gradient evaluation of the same GP (what find_MAP repeats): factorisation,
    # replicated L^-1, sharded Sigma^-1 + trace reductions, all-reduce of the accumulators
    eng.factorize()
    val, grad = eng.nlml(grad=True)
    sync()
    t3 = time.perf_counter()
    on_dev = dist is None or dist.get_backend() == "nccl"
    times = torch.tensor([t1 - t0, t2 - t1, t3 - t2], dtype=torch.float64,
                         device=torch.device("cuda", local_rank) if on_dev else "cpu")
    if dist is not None:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    fit_s, pred_s, eval_s = (float(v) for v in times.cpu())
    eng.close()
    flops_fit = float(N) ** 3 / 3.0
    flops_pred = float(N) ** 2 * len(Xs)
    return {
        "workload": f"ONE GP, N={N}, d={d}, RBF-ARD fp64, fixed theta; block-cyclic rows over {world} GPU(s)",
        "fit_fixed_theta_s": round(fit_s, 4),
        "predict_s": round(pred_s, 4),
        "fit_tflops": round(flops_fit / fit_s / 1e12, 2),
        "predict_tflops": round(flops_pred / pred_s / 1e12, 2),
        "map_eval_s": round(eval_s, 4),
        "map_eval_tflops": round(float(N) ** 3 / eval_s / 1e12, 2),
        "grad_finite": bool(np.all(np.isfinite(grad))),
        "nlml": float(nlml),
        "results_finite": bool(np.all(np.isfinite(mu)) and np.all(var > 0)),
    }


def run_with_deadline(fn, seconds):
    """fn() on a helper thread; (result, finished).  A collective that never completes (a peer
    died, a link is down) must not take the headline line with it: the caller prints what it has
    and leaves with os._exit."""
    import threading

    box = {}

    def target():
        try:
            box["value"] = fn()
        except Exception as err:  # never lose the headline line to the side measurement
            box["value"] = {"error": f"{type(err).__name__}: {err}"[:300]}

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
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--map-evals", type=int, default=0,
                    help="cap on L-BFGS objective evaluations per fit (0 = run to convergence, cap 200)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    cfg = CONFIGS[args.config]

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
    gp = build_gp(cfg, device=local_rank)
    eng = gp.engine
    Xs = synthetic_grid(cfg["d"], cfg["res"])
    M = len(Xs)
    dev = torch.device("cuda", local_rank)
    xs_dev = torch.tensor(Xs, dtype=torch.float64, device=dev)  # X* resident in HBM
    mean_dev = torch.empty(M, dtype=torch.float64, device=dev)
    var_dev = torch.empty(M, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    maxeval = args.map_evals if args.map_evals > 0 else 200

    def one_step():
        gp.find_MAP(maxeval=maxeval)
        eng.predict_device(xs_dev.data_ptr(), M, cfg["d"], mean_dev.data_ptr(), var_dev.data_ptr(), True)
        return gp.n_eval

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    n_evals = [one_step() for _ in range(args.steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    # Roofline pass: ONE more step of the same workload with a HIP event pair around every GEMM
    # launch (on the stream it is launched on).  Kept out of the timed region above: ~900 event
    # pairs per MAP evaluation cost 3.5 ms of host time per evaluation (+13 % on the step).
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
        "factorize_ms": wall_ms(eng.factorize),          # K-build + Cholesky + L^-1 y + log-det
        "factorize_plus_gradient_ms": wall_ms(fact_then_grad),  # one MAP objective evaluation
    }
    eng.factorize()
    phases["predict_ms"] = wall_ms(lambda: eng.predict_device(xs_dev.data_ptr(), M, cfg["d"], mean_dev.data_ptr(),
                                                                var_dev.data_ptr(), True))
    phases["ls_limits_ms"] = wall_ms(lambda: gp._prepare_lengthscales(gp.model.X, ARD=True))
    phases["profiled_last_evaluation_ms"] = {k: round(tm[k], 3) for k in
                                             ("kbuild_ms", "chol_ms", "chol_leaf_ms", "chol_trsm_ms", "chol_gemm_ms",
                                              "grad_ms", "grad_gemm_ms", "predict_ms", "predict_gemm_ms")}

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    mean_host = mean_dev.cpu().numpy()
    var_host = var_dev.cpu().numpy()
    finite = bool(np.all(np.isfinite(mean_host)) and np.all(var_host > 0))
    # the headline engine's streams go away before the distributed section creates its own: more
    # than four live HIP streams per process slow every kernel down on this stack
    eng.close()
    gp.engine = None

    dist_info, healthy = None, True
    if os.environ.get("GUMBI_BENCH_NO_DIST") != "1" and args.config == "c2":
        def section():
            torch.cuda.set_device(local_rank)  # the current device is per-thread state
            return distributed_section(world, local_rank, dist)

        dist_info, healthy = run_with_deadline(section, float(os.environ.get("GUMBI_BENCH_DIST_DEADLINE", "300")))

    flops_total = sum(step_flops(cfg["N"], M, n) for n in n_evals) * world

    if rank == 0:
        gemm_tf = tm["total_gemm_flops"] / max(tm["total_gemm_ms"], 1e-9) / 1e9
        kb_gbs = tm["total_kbuild_bytes"] / max(tm["total_kbuild_ms"], 1e-9) / 1e6
        out = {
            "metric": "fit+predict achieved GFLOP/s (fp64 exact GP: MAP fit + grid prediction)",
            "value": round(flops_total / elapsed / 1e9, 2),
            "unit": "GFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": cfg["label"],
                "N": cfg["N"], "d": cfg["d"], "kernel": cfg["kernel"], "M": M,
                "map_evals_per_step": n_evals,
                "parallelism": "1 GPU" if world == 1 else f"{world} independent GPs, one per GPU (no data-path collective)",
            },
            "fit_predict_seconds": round(elapsed / args.steps, 4),
            "roofline": {
                "bound": "mfma",
                "kernel": "gemm_f64_kernel (v_mfma_f64_16x16x4_f64 SYRK/GEMM: Cholesky trailing update, "
                          "triangular solves, inverse)",
                "achieved": round(gemm_tf, 3),
                "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(gemm_tf / FP64_MFMA_PEAK_TFLOPS, 4),
                "traffic": None,
                "measured_over": "one extra step of the same workload after the timed region (per-launch HIP events)",
                "launches": int(tm["total_gemm_launches"]),
                "avg_launch_ms": round(tm["total_gemm_ms"] / max(tm["total_gemm_launches"], 1), 5),
                "flops_per_launch": round(tm["total_gemm_flops"] / max(tm["total_gemm_launches"], 1), 1),
                # launches of the look-ahead schedule's two streams overlap: the same flops over the WALL time
                # with at least one GEMM launch in flight (union of the launch intervals)
                "achieved_over_wall_time": round(tm["total_gemm_flops"] / max(tm.get("total_gemm_wall_ms", 0.0), 1e-9) / 1e9, 3),
            },
            "kbuild": {
                "bound": "hbm", "achieved": round(kb_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(kb_gbs / HBM_PEAK_GBS, 4), "launches": int(tm["total_kbuild_launches"]),
            },
            "results_finite": finite,
            "phases": phases,
        }
        if tm.get("masked_gemm_flops", 0.0) > 0.0:
            # trailing updates of the masked look-ahead schedule run on masked_cus of the chip's compute
            # units BY DESIGN (the rest serves the concurrent panel chain): their share and rate are
            # reported separately; `achieved` / `frac` above stay the plain all-launch figures
            ncu = 256
            mtf = tm["masked_gemm_flops"] / max(tm["masked_gemm_ms"], 1e-9) / 1e9
            utf = (tm["total_gemm_flops"] - tm["masked_gemm_flops"]) / max(tm["total_gemm_ms"] - tm["masked_gemm_ms"], 1e-9) / 1e9
            out["roofline"]["cu_masked_launches"] = {
                "compute_units": int(tm["masked_cus"]), "of": ncu,
                "flops_share": round(tm["masked_gemm_flops"] / tm["total_gemm_flops"], 4),
                "achieved": round(mtf, 3),
                "frac_of_their_share_of_peak": round(mtf / (FP64_MFMA_PEAK_TFLOPS * tm["masked_cus"] / ncu), 4),
                "unmasked_achieved": round(utf, 3),
            }
        pt = pmc_traffic(args.config)
        if pt is not None:
            out["roofline"]["traffic"] = round(pt["bytes_per_launch"], 1)
            out["roofline"]["traffic_unit"] = "HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE)"
            out["roofline"]["traffic_source"] = pt["source"]
        try:
            tf, cyc = engine.mfma_f64_peak(local_rank)
            out["roofline"]["mfma_only_microbench_tflops"] = round(tf, 2)  # sustained ceiling under DVFS
        except Exception:
            pass
        if dist_info is not None:
            out["distributed"] = dist_info
        if not args.no_cpu_baseline and world == 1 and os.environ.get("GUMBI_BENCH_NO_CPU") != "1":
            out["cpu_baseline"] = cpu_baseline(cfg)
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
            os._exit(0)


if __name__ == "__main__":
    main()
