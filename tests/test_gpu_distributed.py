"""Multi-rank path on real HIP kernels: two processes share the one GPU of the test box and
exchange panels through ``gloo`` (RCCL refuses two ranks on one device); the 8-GPU RCCL run is the
driver's SCALE job.  Checks factor, v, NLML and predictions against the oracle."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, N, d, M, out, model="matern"):
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
        else:
            X, y, ls = O.synthetic_table(N, d, seed=5)
            spec = O.make_spec(d, range(d), kind="Matern52")
            theta = O.pack_theta(spec, ls, 1.1, 0.3)
            Xs = np.random.default_rng(1).standard_normal((M, d))
            kspec = KernelSpec(D=d, idx_cont=list(range(d)), kind="Matern52")
        eng = DistributedEngine(0)
        eng.set_data(X, y)
        eng.set_kernel(kspec)
        eng.set_theta(theta)
        eng.factorize()
        L_ref, v_ref = O.factorize(spec, theta, X, y, dist_mode="direct")
        L = np.tril(eng.eng.copy_factor())
        err_L = np.max(np.abs(L - L_ref)) / np.max(np.abs(L_ref))
        err_v = np.max(np.abs(eng.eng.copy_v() - v_ref)) / np.max(np.abs(v_ref))
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
        # the replicated-inverse variant (only Sigma^-1 sharded) must agree with the partitioned one
        eng.partition_inverse = False
        eng.factorize()
        val2, g2 = eng.nlml(grad=True)
        err_g = max(err_g, np.max(np.abs(g2 - g)) / max(1.0, np.max(np.abs(g))) * 1e2)  # 1e-10 relative
        out.put((rank, err_L, err_v, err_nl, err_mu, err_var, err_g, g.tobytes()))
        eng.close()
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,N,model", [(2, 700, "matern"), (2, 512, "matern"), (3, 1000, "matern"),
                                           (2, 600, "additive")])
def test_two_ranks_one_gpu_match_oracle(gpu, world, N, model):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, 3, 333, out, model)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err_L, err_v, err_nl, err_mu, err_var, err_g, gbytes in results:
        assert err_L < 1e-10 and err_v < 1e-10 and err_nl < 1e-8
        assert err_mu < 1e-8 and err_var < 1e-9
        assert err_g < 1e-8
    assert len({r[-1] for r in results}) == 1  # bit-identical gradient on every rank (optimisers stay in lock step)


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
    results = [out.get(timeout=300) for _ in range(world)]
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
    results = [out.get(timeout=300) for _ in range(2)]
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
        eng = DistributedEngine(0)
        eng.force_partition = True
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
        out.put((dist.get_backend(), abs(val - val_r) / abs(val_r), float(np.max(np.abs(g - g_r)) / max(1.0, np.max(np.abs(g_r)))),
                 float(np.max(np.abs(mu - mu_r)) / np.max(np.abs(mu_r))), float(np.max(np.abs(var - var_r)))))
        eng.close()
    finally:
        dist.destroy_process_group()


def test_driver_on_the_rccl_backend_with_one_rank(gpu):
    """The same driver on the nccl (= RCCL) backend: device-tensor broadcast / all-gather / all-reduce on
    the engine's side stream, no host staging.  One rank is all a one-GPU box allows (RCCL refuses two
    ranks on one device); the collectives are trivial but every call the 8-GPU run makes is made."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(out,))
    p.start()
    backend, e_val, e_g, e_mu, e_var = out.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert backend == "nccl"
    assert e_val < 1e-10 and e_g < 1e-8 and e_mu < 1e-8 and e_var < 1e-9


def test_bench_contract_with_two_ranks_on_one_gpu(gpu):
    """bench.py as the driver launches it for N > 1 (``python -m torch.distributed.run --nproc-per-node N
    bench.py --gpus N ...``), with both ranks on the test box's one GPU over gloo: rank 0 prints exactly ONE
    JSON line with the contract's keys, n_gpus = 2, the replica workload and the distributed section."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, GUMBI_BENCH_SINGLE_DEVICE="1", GUMBI_BENCH_BACKEND="gloo", GUMBI_BENCH_DIST_N="2304",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(root / "bench.py"), "--gpus", "2", "--steps", "1",
           "--warmup", "0", "--map-evals", "4"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["scaling"] == "weak" and d["dtype"] == "f64"
    assert d["value"] > 0 and d["results_finite"] and "cpu_baseline" not in d
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    dist = d["distributed"]
    assert "error" not in dist, dist
    assert dist["results_finite"] and dist["grad_finite"] and "2 GPU(s)" in dist["workload"]


def test_bench_contract_single_process(gpu):
    """``python bench.py`` (N = 1): one JSON line with the contract's keys plus ``roofline``, ``cpu_baseline``
    (here on a shortened sample) and the per-phase timings."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, GUMBI_BENCH_DIST_N="2304", GUMBI_BENCH_CPU_SECONDS="2")
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--steps", "1", "--warmup", "0", "--map-evals", "4"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["unit"] == "GFLOP/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["config"]["workload"].startswith("synthetic N=10k d=4") and len(d["config"]["map_evals_per_step"]) == 1
    assert 1 <= d["config"]["map_evals_per_step"][0] <= 8  # scipy checks maxfun between line searches
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 20.0 < r["achieved"] < r["peak"] and r["launches"] > 100
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "N=" in c["sample"]
    assert d["phases"]["factorize_ms"] > 0 and d["phases"]["predict_ms"] > 0
    assert d["distributed"]["results_finite"] and "1 GPU(s)" in d["distributed"]["workload"]
