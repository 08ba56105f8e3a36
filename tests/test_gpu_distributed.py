"""Multi-rank path on real HIP kernels: the native driver (gumbi_amd/csrc/dist_driver.hpp, ``gmb_dist_*``)
with two / three processes sharing the one GPU of the test box and exchanging panels through ``gloo``
(RCCL refuses two ranks on one device), and with the library's own RCCL communicator on one rank; the 8-GPU
RCCL run is the driver's SCALE job.  Checks factor, v, NLML, gradient and predictions against the oracle."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _collect(out, n, timeout):
    """n results from the workers' queue; a worker that died reports its traceback instead of a result."""
    results = []
    for _ in range(n):
        r = out.get(timeout=timeout)
        if isinstance(r, tuple) and r and r[0] == "error":
            pytest.fail("worker failed:\n" + r[1])
        results.append(r)
    return results


def _worker(rank, world, port, N, d, M, out, model="matern", panel=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch  # noqa: F401
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gumbi_amd.distributed import DistributedEngine
        from gumbi_amd.engine import KernelSpec
        from oracle import gp_oracle as O

        if model == "additive":
            # additive two-output model with a linear and a categorical dim (reference pymc/GP.py:732-754):
            # the K-build of the owned block rows and the sharded gradient run term by term
            from test_oracle import additive_problem

            spec, theta, X, y = additive_problem(n=N // 2)
            Xs = X[::3].copy()
            Xs[:, 0] += 0.21
            kspec = KernelSpec(**{k: spec[k] for k in ("D", "idx_cont", "kind", "ard", "idx_lin", "coreg", "out_col",
                                                        "n_out", "hetero_noise", "jitter", "additive")})
        elif model == "composite":
            # linear x coregion x two outputs x heteroskedastic noise: every accumulator class of the sharded
            # gradient (lengthscales, linear, coregion tables, noise table) is hit
            from pathlib import Path

            gold = np.load(Path(__file__).resolve().parent / "golden" / "gp_goldens.npz", allow_pickle=True)
            spec = O.make_spec(4, [0, 1], idx_lin=[1], coreg=[(2, 3)], out_col=3, n_out=2, hetero_noise=True)
            Xg, yg, theta = (gold[f"composite_N140/{k}"] for k in ("X", "y", "theta"))
            rng = np.random.default_rng(3)
            reps = max(2, N // len(yg))
            X = np.concatenate([Xg] * reps)
            X[:, spec["idx_cont"]] += 0.05 * rng.standard_normal((len(X), len(spec["idx_cont"])))
            y = np.concatenate([yg] * reps) + 0.1 * rng.standard_normal(len(X))
            Xs = X[::5].copy()
            Xs[:, spec["idx_cont"][0]] += 0.13
            kspec = KernelSpec(**{k: spec[k] for k in ("D", "idx_cont", "kind", "ard", "idx_lin", "coreg", "out_col",
                                                        "n_out", "hetero_noise", "jitter")})
        else:
            X, y, ls = O.synthetic_table(N, d, seed=5)
            spec = O.make_spec(d, range(d), kind="Matern52")
            theta = O.pack_theta(spec, ls, 1.1, 0.3)
            Xs = np.random.default_rng(1).standard_normal((M, d))
            kspec = KernelSpec(D=d, idx_cont=list(range(d)), kind="Matern52")
        eng = DistributedEngine(0, panel_blocks=panel)
        assert eng.comm.kind == "torch-gloo"
        eng.set_data(X, y)
        eng.set_kernel(kspec)
        eng.set_theta(theta)
        eng.factorize()
        L_ref, v_ref = O.factorize(spec, theta, X, y, dist_mode="direct")
        L = np.tril(eng.copy_factor())
        err_L = np.max(np.abs(L - L_ref)) / np.max(np.abs(L_ref))
        err_v = np.max(np.abs(eng.copy_v() - v_ref)) / np.max(np.abs(v_ref))
        err_nl = abs(eng.nlml() - O.nlml(spec, theta, X, y, dist_mode="direct"))
        mu, var = eng.predict(Xs)
        mu_r, var_r = O.predict(spec, theta, X, y, Xs, dist_mode="direct")
        err_mu = np.max(np.abs(mu - mu_r)) / np.max(np.abs(mu_r))
        err_var = np.max(np.abs(var - var_r))
        # distributed gradient: every rank must return the same (value, gradient) == the oracle's
        eng.factorize()
        val, g = eng.nlml(grad=True)
        val_r, g_r = O.nlml_and_grad(spec, theta, X, y, dist_mode="direct")
        err_g = max(abs(val - val_r) / abs(val_r), np.max(np.abs(g - g_r)) / max(1.0, np.max(np.abs(g_r))))
        # alpha = Sigma^-1 y assembled from the ranks' rows of U
        from scipy.linalg import solve_triangular

        err_g = max(err_g, float(np.max(np.abs(eng.copy_alpha() - solve_triangular(L_ref, v_ref, lower=True, trans="T")))
                               / np.max(np.abs(v_ref))) * 1e-2)
        out.put((rank, err_L, err_v, err_nl, err_mu, err_var, err_g, g.tobytes()))
        eng.close()
    except BaseException:
        import traceback

        out.put(("error", traceback.format_exc()[-3000:]))
        raise
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,N,model,panel", [(2, 700, "matern", 0), (2, 512, "matern", 1), (3, 1000, "matern", 2),
                                                 (2, 600, "additive", 0), (3, 420, "composite", 1),
                                                 (2, 2500, "matern", 3), (3, 5000, "matern", 0),
                                                 (3, 100, "matern", 0), (2, 128, "matern", 1), (4, 130, "matern", 2)])
def test_two_ranks_one_gpu_match_oracle(gpu, world, N, model, panel):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, 3, 333, out, model, panel)) for r in range(world)]
    for p in procs:
        p.start()
    results = _collect(out, world, 300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err_L, err_v, err_nl, err_mu, err_var, err_g, gbytes in results:
        assert err_L < 1e-10 and err_v < 1e-10 and err_nl < 1e-8
        assert err_mu < 1e-8 and err_var < 1e-9
        assert err_g < 1e-8
    assert len({r[-1] for r in results}) == 1  # bit-identical gradient on every rank (optimisers stay in lock step)


def _notpd_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch  # noqa: F401
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gumbi_amd.distributed import DistributedEngine
        from gumbi_amd.engine import Engine, KernelSpec
        from oracle import gp_oracle as O

        N, d = 900, 3
        X, y, ls = O.synthetic_table(N, d, seed=2)
        bad = 517  # in the fifth block row: owned by another rank than block row 0
        X[bad, 1] = np.nan
        theta = np.concatenate([ls, [1.0, 0.2]])
        got = []
        for eng in (DistributedEngine(0, panel_blocks=2), Engine(0)):
            eng.set_data(X, y)
            eng.set_kernel(KernelSpec(D=d, idx_cont=list(range(d))))
            eng.set_theta(theta)
            try:
                eng.factorize()
                got.append(("no error", -1))
            except np.linalg.LinAlgError as err:
                got.append(("LinAlgError", eng.notpd_index(), str(err)[:80]))
            # the engine stays usable: a clean table factorises afterwards
            X2 = X.copy()
            X2[bad, 1] = 0.25
            eng.set_data(X2, y)
            eng.set_theta(theta)
            eng.factorize()
            got.append(round(float(eng.nlml()), 6))
            eng.close()
        out.put((rank, got))
    except BaseException:
        import traceback

        out.put(("error", traceback.format_exc()[-3000:]))
        raise
    finally:
        dist.destroy_process_group()


def test_not_positive_definite_is_reported_identically_on_every_rank(gpu):
    """A NaN input makes the covariance non-factorisable at its row: every rank of the multi-GPU driver must
    return the SAME failure (GMB_ENOTPD -> LinAlgError, same global row index as the single engine -- the
    diagonal squares are factored redundantly, so no rank can run ahead into a collective the others skip),
    and the engines must remain usable afterwards."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_notpd_worker, args=(r, 3, port, out)) for r in range(3)]
    for p in procs:
        p.start()
    results = _collect(out, 3, 300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got in results:
        dist_err, dist_nl, single_err, single_nl = got
        assert dist_err[0] == "LinAlgError" and single_err[0] == "LinAlgError", got
        assert dist_err[1] == single_err[1] == 517, got
        assert dist_nl == single_nl
    assert len({str(g) for _, g in results}) == 1


def _large_worker(rank, world, port, N, d, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch  # noqa: F401
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gumbi_amd.distributed import DistributedEngine
        from gumbi_amd.engine import Engine, KernelSpec
        from oracle import gp_oracle as O

        X, y, ls = O.synthetic_table(N, d, seed=7)
        theta = np.concatenate([ls, [1.0, 0.2]])
        kspec = KernelSpec(D=d, idx_cont=list(range(d)), kind="ExpQuad")
        Xs = O.synthetic_grid(d, 24)
        res = {}
        for name, eng in (("dist", DistributedEngine(0)), ("single", Engine(0))):
            eng.set_data(X, y)
            eng.set_kernel(kspec)
            eng.set_theta(theta)
            eng.factorize()
            rows = [0, 127, 128, 5000, N // 2 + 1, N - 1]
            Lr = np.stack([eng.copy_factor(r, 1, 0, N)[0] for r in rows])
            for a, r in enumerate(rows):
                Lr[a, r + 1:] = 0.0  # above the diagonal the buffer holds scratch (schedule-dependent)
            v = eng.copy_v()
            mu, var = eng.predict(Xs)
            val, g = eng.nlml(grad=True)
            res[name] = (Lr, v, mu, var, val, g)
            eng.close()
            del eng
        a, b = res["dist"], res["single"]
        errs = [float(np.max(np.abs(x - z)) / max(np.max(np.abs(z)), 1e-300)) for x, z in zip(a[:4], b[:4])]
        errs.append(abs(a[4] - b[4]) / abs(b[4]))
        errs.append(float(np.max(np.abs(a[5] - b[5])) / max(1.0, np.max(np.abs(b[5])))))
        ref = O.nlml(O.make_spec(d, range(d)), theta, X, y) if rank == 0 else a[4]
        out.put((rank, errs, abs(a[4] - ref) / abs(ref), a[5].tobytes()))
    except BaseException:
        import traceback

        out.put(("error", traceback.format_exc()[-3000:]))
        raise
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_at_n_20k_match_the_single_engine(gpu):
    """The block-cyclic driver well beyond a handful of blocks: N = 20,480 (160 block columns, 20 panels of
    the default width, 5 chunks of the row-partitioned inverse) over two ranks sharing the GPU: sampled rows
    of the factor, v, predictions, NLML and gradient equal the single-GPU engine's (same kernels, different
    order of the trailing updates), the NLML equals the oracle's, both ranks hold identical bits."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_large_worker, args=(r, 2, port, 20_480, 4, out)) for r in range(2)]
    for p in procs:
        p.start()
    results = _collect(out, 2, 900)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, errs, err_oracle, _ in results:
        assert max(errs[:2]) < 1e-9 and errs[2] < 1e-8 and errs[3] < 1e-8, errs   # factor rows, v, mean, variance
        assert errs[4] < 1e-11 and errs[5] < 1e-7, errs                            # NLML, gradient
        assert err_oracle < 1e-10
    assert results[0][-1] == results[1][-1]


def _fit_worker(rank, world, port, N, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch  # noqa: F401
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pandas as pd

        import gumbi_amd as gmb
        from oracle import gp_oracle as O

        X, y, _ = O.synthetic_table(N, 2, seed=9)
        df = pd.DataFrame(X, columns=["a", "b"])
        df["y"] = y
        grid = np.random.default_rng(2).uniform(-1.5, 1.5, (50, 2))

        def fit(distributed):
            gp = gmb.GP(gmb.DataSet(df, outputs=["y"]), outputs=["y"], distributed=distributed)
            gp.specify_model(continuous_dims=["a", "b"])
            gp.build_model()
            gp.find_MAP()
            pts = gp.parray(a=grid[:, 0], b=grid[:, 1])
            pred = gp.predict_points(pts)
            res = (gp._theta_fitted.copy(), np.asarray(pred.μ).copy(), np.asarray(pred.σ2).copy(), gp.n_eval)
            gp.engine.close()
            return res

        th_d, mu_d, var_d, n_d = fit(True)
        th_s, mu_s, var_s, n_s = fit(None)  # the same fit on this rank's GPU alone
        out.put((rank, float(np.max(np.abs(th_d - th_s) / np.maximum(np.abs(th_s), 1e-3))),
                 float(np.max(np.abs(mu_d - mu_s))), float(np.max(np.abs(var_d - var_s))), n_d, n_s, th_d.tobytes()))
    except BaseException:
        import traceback

        out.put(("error", traceback.format_exc()[-3000:]))
        raise
    finally:
        dist.destroy_process_group()


def test_distributed_map_fit_matches_single_gpu_fit(gpu):
    """GP(..., distributed=True).find_MAP() over two ranks: same optimum and predictions as the
    single-GPU fit, identical parameters on both ranks."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_fit_worker, args=(r, world, port, 400, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = _collect(out, world, 300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err_th, err_mu, err_var, n_d, n_s, _ in results:
        assert err_th < 1e-5 and err_mu < 1e-6 and err_var < 1e-6
    assert results[0][-1] == results[1][-1]


def _cv_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch  # noqa: F401
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pandas as pd

        import gumbi_amd as gmb
        from oracle import gp_oracle as O

        X, y, _ = O.synthetic_table(300, 2, seed=4)
        df = pd.DataFrame(X, columns=["a", "b"])
        df["y"] = y
        gp = gmb.GP(gmb.DataSet(df, outputs=["y"]), outputs=["y"])
        gp.specify_model(continuous_dims=["a", "b"])
        gp.build_model()
        res = gp.cross_validate_replicas([11, 12, 13], pct_train=0.7)
        out.put((rank, [float(r["test"]["NLPDs"].mean()) for r in res], [len(r["train"]["data"].wide) for r in res]))
    except BaseException:
        import traceback

        out.put(("error", traceback.format_exc()[-3000:]))
        raise
    finally:
        dist.destroy_process_group()


def test_cross_validation_replicas_over_ranks(gpu):
    """Three independent splits dealt over two ranks: both ranks end up with all three results, equal
    to what one process computes alone."""
    import pandas as pd
    import torch.multiprocessing as mp

    import gumbi_amd as gmb
    from oracle import gp_oracle as O

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cv_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    results = _collect(out, 2, 300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0][1:] == results[1][1:]
    X, y, _ = O.synthetic_table(300, 2, seed=4)
    df = pd.DataFrame(X, columns=["a", "b"])
    df["y"] = y
    gp = gmb.GP(gmb.DataSet(df, outputs=["y"]), outputs=["y"])
    gp.specify_model(continuous_dims=["a", "b"])
    gp.build_model()
    alone = gp.cross_validate_replicas([11, 12, 13], pct_train=0.7)
    # (MAP optima agree to the optimiser's tolerance, not bitwise: the trace reductions use atomics)
    assert np.allclose([float(r["test"]["NLPDs"].mean()) for r in alone], results[0][1], rtol=1e-4)
    assert [len(r["train"]["data"].wide) for r in alone] == results[0][2] == [210, 210, 210]


def _rccl_worker(out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from gumbi_amd.distributed import DistributedEngine
        from gumbi_amd.engine import KernelSpec
        from oracle import gp_oracle as O

        N, d = 900, 3
        X, y, ls = O.synthetic_table(N, d, seed=8)
        spec = O.make_spec(d, range(d))
        theta = O.pack_theta(spec, ls, 1.0, 0.25)
        eng = DistributedEngine(0, panel_blocks=2)
        kind = eng.comm.kind
        eng.set_data(X, y)
        eng.set_kernel(KernelSpec(D=d, idx_cont=list(range(d))))
        eng.set_theta(theta)
        eng.factorize()
        val, g = eng.nlml(grad=True)
        val_r, g_r = O.nlml_and_grad(spec, theta, X, y, dist_mode="direct")
        eng.factorize()
        Xs = np.random.default_rng(3).standard_normal((200, d))
        mu, var = eng.predict(Xs)
        mu_r, var_r = O.predict(spec, theta, X, y, Xs, dist_mode="direct")
        out.put((dist.get_backend() + "/" + kind, abs(val - val_r) / abs(val_r), float(np.max(np.abs(g - g_r)) / max(1.0, np.max(np.abs(g_r)))),
                 float(np.max(np.abs(mu - mu_r)) / np.max(np.abs(mu_r))), float(np.max(np.abs(var - var_r)))))
        eng.close()
    except BaseException:
        import traceback

        out.put(("error", traceback.format_exc()[-3000:]))
        raise
    finally:
        dist.destroy_process_group()


def test_driver_on_the_rccl_backend_with_one_rank(gpu):
    """The same driver on its production transport: the library's OWN RCCL communicator (unique id from
    ncclGetUniqueId, ncclCommInitRank, ncclAllGather called from the C++ panel loop on the engine's streams).
    One rank is all a one-GPU box allows (RCCL refuses two ranks on one device); the collectives are trivial
    but every call the 8-GPU run makes is made."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(out,))
    p.start()
    (backend, e_val, e_g, e_mu, e_var), = _collect(out, 1, 300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert backend == "nccl/rccl"
    assert e_val < 1e-10 and e_g < 1e-8 and e_mu < 1e-8 and e_var < 1e-9


def test_bench_contract_with_two_ranks_on_one_gpu(gpu):
    """bench.py as the driver launches it for N > 1 (``python -m torch.distributed.run --nproc-per-node N
    bench.py --gpus N ...``), with both ranks on the test box's one GPU over gloo: rank 0 prints exactly ONE
    JSON line with the contract's keys, n_gpus = 2, and the workload is ONE GP partitioned over the ranks."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, GUMBI_BENCH_SINGLE_DEVICE="1", GUMBI_BENCH_BACKEND="gloo", GUMBI_BENCH_DIST_N="3000",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(root / "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert "error" not in d, d
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong" and d["dtype"] == "f64"
    assert d["value"] > 0 and d["results_finite"] and "cpu_baseline" not in d
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert "ONE GP" in d["config"]["workload"] and "over 2 GPUs" in d["config"]["parallelism"]
    assert "torch-gloo" in d["config"]["parallelism"] and d["config"]["N"] == 3000
    ph = d["phases"]
    assert ph["map_eval_s"] > 0 and ph["fit_fixed_theta_s"] > 0 and ph["predict_s"] > 0 and np.isfinite(ph["nlml"])
    # whole-job value = algorithmic flops of the steps / max-over-ranks wall time
    N, M = 3000, 10_000
    flops = 2 * (float(N) ** 3 + float(N) ** 3 / 3 + float(N) ** 2 * M + 4.0 * N * M)
    assert abs(d["value"] - flops / (d["ms_per_step"] * 2e-3) / 1e9) / d["value"] < 1e-3


def test_bench_contract_single_process(gpu):
    """``python bench.py --config c2`` (N = 1): one JSON line with the contract's keys plus ``roofline`` (the
    trailing update alone), ``cpu_baseline`` (here on a shortened sample), the per-phase timings, the
    end-to-end fit and the one-GP section (shrunk by the test hook)."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, GUMBI_BENCH_DIST_N="2304", GUMBI_BENCH_CPU_SECONDS="2")
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--config", "c2", "--steps", "1", "--warmup", "0",
                          "--map-evals", "4"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["unit"] == "GFLOP/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["config"]["workload"].startswith("synthetic N=10k d=4") and len(d["config"]["map_evals_per_step"]) == 1
    assert 1 <= d["config"]["map_evals_per_step"][0] <= 8  # scipy checks maxfun between line searches
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "trailing-update" in r["kernel"] and 15.0 < r["achieved"] < r["peak"] and r["launches"] > 50
    assert r["all_gemm_launches"]["launches"] > r["launches"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "N=" in c["sample"]
    assert d["phases"]["factorize_ms"] > 0 and d["phases"]["predict_ms"] > 0
    assert d["kbuild"]["unit"] == "GB/s" and d["kbuild"]["achieved"] > 100
    assert d["end_to_end"]["results_finite"] and d["end_to_end"]["total_s"] > 0
    side = d["c5_single_gpu"]
    assert "error" not in side, side
    assert side["results_finite"] and side["value"] > 0 and "N=2304" in side["workload"]


def test_bench_default_config_is_the_largest_single_gpu_one():
    """Without --config the one-GPU headline is C3 (N = 50k, d = 8, Matern-5/2); checked on the argument
    handling only (the full C3 run is the driver's bench job)."""
    import importlib.util
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("bench_cfg", root / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["bench_cfg"] = mod
    spec.loader.exec_module(mod)
    src = (root / "bench.py").read_text()
    assert 'args.config or ("c3" if world == 1 else "c5")' in src
    assert mod.CONFIGS["c3"]["label"].startswith("synthetic N=50k d=8 Matern-5/2")
