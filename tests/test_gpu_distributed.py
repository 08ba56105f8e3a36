"""Multi-rank path on real HIP kernels: the native driver (gumbi_amd/csrc/dist_driver.hpp, ``gmb_dist_*``)
with two / three / four processes sharing the one GPU of the test box and exchanging panels through ``gloo``
(RCCL refuses two ranks on one device), and with the library's own RCCL communicator on one rank; the 8-GPU
RCCL run is the driver's SCALE job.  Checks factor, v, NLML, gradient and predictions against the oracle.

ONE pool of four worker processes serves every multi-rank test of this module (a ``gloo`` world of four with
sub-groups of one to four ranks): a process that imports torch costs 5 - 20 s on a cold box, and a spawn per
parametrisation (41 processes in round 2) was two thirds of the suite's wall time there."""
import os
import queue
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

POOL_WORLD = 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _pool_main(rank, port, tasks, out):
    """A pool worker: joins the 4-rank gloo world once, then serves (function name, ranks, args) tasks.  A task
    runs on ranks < `ranks` with the sub-group of exactly those ranks; its function puts ONE result on `out`."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from datetime import timedelta

    import torch  # noqa: F401
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=POOL_WORLD, timeout=timedelta(seconds=240))
    groups = {w: dist.new_group(list(range(w)), timeout=timedelta(seconds=240)) for w in range(1, POOL_WORLD)}
    groups[POOL_WORLD] = dist.group.WORLD
    out.put(("ready", rank))
    try:
        while True:
            task = tasks.get()
            if task is None:
                break
            name, ranks, args = task
            try:
                globals()[name](rank, ranks, groups[ranks], out, *args)
            except BaseException:
                import traceback

                out.put(("error", f"rank {rank}: " + traceback.format_exc()[-3000:]))
    finally:
        dist.destroy_process_group()


class Pool:
    def __init__(self):
        import torch.multiprocessing as mp

        ctx = mp.get_context("spawn")
        self.tasks = [ctx.Queue() for _ in range(POOL_WORLD)]
        self.out = ctx.Queue()
        port = _free_port()
        self.procs = [ctx.Process(target=_pool_main, args=(r, port, self.tasks[r], self.out), daemon=True)
                      for r in range(POOL_WORLD)]
        for p in self.procs:
            p.start()
        for _ in range(POOL_WORLD):
            assert self.out.get(timeout=600)[0] == "ready"
        self.broken = False

    def run(self, fn, ranks, *args, timeout=300):
        """fn(rank, ranks, group, out, *args) on ranks 0 .. ranks-1; their results, or the test fails with the
        first worker's traceback (and the pool is rebuilt for the next test: its ranks may be out of step)."""
        for r in range(ranks):
            self.tasks[r].put((fn.__name__, ranks, args))
        results = []
        for _ in range(ranks):
            try:
                r = self.out.get(timeout=timeout)
            except queue.Empty:
                self.broken = True
                pytest.fail(f"{fn.__name__}: no result from a worker within {timeout} s")
            if isinstance(r, tuple) and r and r[0] == "error":
                self.broken = True
                pytest.fail("worker failed:\n" + r[1])
            results.append(r)
        return results

    def close(self):
        for q in self.tasks:
            q.put(None)
        for p in self.procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()


_POOL = [None]


@pytest.fixture
def pool(gpu):
    if _POOL[0] is not None and _POOL[0].broken:
        for p in _POOL[0].procs:
            p.kill()
        _POOL[0] = None
    if _POOL[0] is None:
        _POOL[0] = Pool()
    return _POOL[0]


@pytest.fixture(scope="module", autouse=True)
def _close_pool():
    yield
    if _POOL[0] is not None:
        _POOL[0].close()
        _POOL[0] = None


def _collect(out, n, timeout):
    """n results from a worker queue; a worker that died reports its traceback instead of a result."""
    results = []
    for _ in range(n):
        r = out.get(timeout=timeout)
        if isinstance(r, tuple) and r and r[0] == "error":
            pytest.fail("worker failed:\n" + r[1])
        results.append(r)
    return results


# ---------------------------------------------------------------------------------------------------------
def _oracle_task(rank, world, group, out, N, d, M, model, panel):
    from gumbi_amd.distributed import DistributedEngine
    from gumbi_amd.engine import KernelSpec, dist_plan
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
    eng = DistributedEngine(0, group, panel_blocks=panel)
    assert eng.comm.kind == "torch-gloo" and eng.comm.world == world
    eng.set_data(X, y)
    eng.set_kernel(kspec)
    eng.set_theta(theta)
    eng.factorize()
    tm = eng.timings()
    # communication probes of the factorisation: one all-gather per SQUARE / PANEL / TAIL step of this rank's plan, the
    # bytes it received, and an exposed part that cannot exceed the total
    plan = dist_plan(len(y), rank, world, panel)
    n_coll = sum(1 for st in plan if st["op"] in ("SQUARE", "PANEL", "TAIL"))
    recv = 8.0 * (world - 1) * sum(st["elems"] for st in plan if st["op"] in ("SQUARE", "PANEL", "TAIL"))
    probes_ok = (tm["dist_world"] == world and tm["dist_chol_collectives"] == n_coll and tm["dist_chol_comm_bytes"] == recv
                 and 0.0 < tm["dist_chol_comm_ms"] and 0.0 <= tm["dist_chol_comm_exposed_ms"] <= tm["dist_chol_comm_ms"] * (1 + 1e-9)
                 and tm["dist_chol_main_wait_ms"] >= 0.0 and tm["dist_chol_bulk_wait_ms"] >= 0.0)
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
    tm = eng.timings()
    probes_ok = probes_ok and (tm["dist_grad_collectives"] >= 3 and tm["dist_grad_comm_bytes"] > 0 and
                               0.0 <= tm["dist_grad_comm_exposed_ms"] <= tm["dist_grad_comm_ms"] * (1 + 1e-9))
    val_r, g_r = O.nlml_and_grad(spec, theta, X, y, dist_mode="direct")
    err_g = max(abs(val - val_r) / abs(val_r), np.max(np.abs(g - g_r)) / max(1.0, np.max(np.abs(g_r))))
    # alpha = Sigma^-1 y assembled from the ranks' rows of U
    from scipy.linalg import solve_triangular

    err_g = max(err_g, float(np.max(np.abs(eng.copy_alpha() - solve_triangular(L_ref, v_ref, lower=True, trans="T")))
                           / np.max(np.abs(v_ref))) * 1e-2)
    eng.close()
    out.put((rank, err_L, err_v, err_nl, err_mu, err_var, err_g, bool(probes_ok), g.tobytes()))


@pytest.mark.parametrize("world,N,model,panel", [(2, 700, "matern", 0), (2, 512, "matern", 1), (3, 1000, "matern", 2),
                                                 (2, 600, "additive", 0), (3, 420, "composite", 1),
                                                 (2, 2500, "matern", 3), (3, 5000, "matern", 0),
                                                 (3, 100, "matern", 0), (2, 128, "matern", 1), (4, 130, "matern", 2)])
def test_two_ranks_one_gpu_match_oracle(pool, world, N, model, panel):
    results = pool.run(_oracle_task, world, N, 3, 333, model, panel)
    for rank, err_L, err_v, err_nl, err_mu, err_var, err_g, probes_ok, gbytes in results:
        assert err_L < 1e-10 and err_v < 1e-10 and err_nl < 1e-8
        assert err_mu < 1e-8 and err_var < 1e-9
        assert err_g < 1e-8
        assert probes_ok
    assert len({r[-1] for r in results}) == 1  # bit-identical gradient on every rank (optimisers stay in lock step)


def _capacity_task(rank, world, group, out, N, d, M, panel, kind):
    """The capacity mode of the native driver (csrc/dist_capacity.hpp): no rank holds the factor -- NLML, v, alpha, gradient and
    predictions against the oracle; the rank's resident bytes against a single engine's doing the same three calls."""
    from scipy.linalg import solve_triangular

    from gumbi_amd.distributed import DistributedEngine
    from gumbi_amd.engine import Engine, GumbiHipError, KernelSpec
    from oracle import gp_oracle as O

    X, y, ls = O.synthetic_table(N, d, seed=5)
    spec = O.make_spec(d, range(d), kind=kind)
    theta = O.pack_theta(spec, ls, 1.1, 0.3)
    Xs = np.random.default_rng(1).standard_normal((M, d))
    kspec = KernelSpec(D=d, idx_cont=list(range(d)), kind=kind)
    eng = DistributedEngine(0, group, panel_blocks=panel, capacity=True)
    eng.set_data(X, y)
    eng.set_kernel(kspec)
    eng.set_theta(theta)
    eng.factorize()
    v, nl = eng.copy_v(), eng.nlml()
    mu, var = eng.predict(Xs)
    val, g = eng.nlml(grad=True)  # (the factor survives: L is never overwritten in this mode)
    alpha = eng.copy_alpha()
    if N <= 6000:  # everything against the oracle
        L_ref, v_ref = O.factorize(spec, theta, X, y, dist_mode="direct")
        nl_r = O.nlml(spec, theta, X, y, dist_mode="direct")
        mu_r, var_r = O.predict(spec, theta, X, y, Xs, dist_mode="direct")
        val_r, g_r = O.nlml_and_grad(spec, theta, X, y, dist_mode="direct")
        alpha_r = solve_triangular(L_ref, v_ref, lower=True, trans="T")
    else:  # the oracle for the NLML (one Cholesky on the host), the single engine for the rest
        nl_r = val_r = O.nlml(spec, theta, X, y) if rank == 0 else nl
        ref = Engine(0)
        ref.set_data(X, y)
        ref.set_kernel(kspec)
        ref.set_theta(theta)
        ref.factorize()
        v_ref = ref.copy_v()
        mu_r, var_r = ref.predict(Xs)
        _, g_r = ref.nlml(grad=True)
        alpha_r = ref.copy_alpha()
        ref.close()
    err_v = np.max(np.abs(v - v_ref)) / np.max(np.abs(v_ref))
    err_nl = abs(nl - nl_r)
    err_mu = np.max(np.abs(mu - mu_r)) / np.max(np.abs(mu_r))
    err_var = np.max(np.abs(var - var_r))
    err_g = max(abs(val - val_r) / abs(val_r), np.max(np.abs(g - g_r)) / max(1.0, np.max(np.abs(g_r))))
    err_a = float(np.max(np.abs(alpha - alpha_r)) / np.max(np.abs(v_ref)))
    mu2, var2 = eng.predict(Xs)  # after the gradient, without a re-factorisation
    same = bool(mu2.tobytes() == mu.tobytes() and var2.tobytes() == var.tobytes())
    try:
        eng.copy_factor()
        no_factor = False
    except (GumbiHipError, ValueError):
        no_factor = True  # nobody holds the factor
    peak = eng.eng.resident_bytes(peak=True)
    eng.close()
    single = Engine(0)
    single.set_data(X, y)
    single.set_kernel(kspec)
    single.set_theta(theta)
    single.factorize()
    single.predict(Xs)
    single.nlml(grad=True)
    peak_single = single.resident_bytes(peak=True)
    single.close()
    out.put((rank, err_v, err_nl, err_mu, err_var, err_g, err_a, same, no_factor, peak, peak_single, g.tobytes()))


@pytest.mark.parametrize("world,N,panel,kind", [(2, 700, 0, "Matern52"), (3, 1000, 2, "ExpQuad"), (2, 512, 1, "Matern32"), (3, 130, 0, "ExpQuad"),
                                                (4, 2500, 3, "Matern52"), (2, 5000, 0, "ExpQuad")])
def test_capacity_mode_matches_the_oracle_without_any_rank_holding_the_factor(pool, world, N, panel, kind):
    """gmb_dist_set_mode(e, 1): every rank keeps its own block rows (packed) and two panel buffers; the squares are factored on
    gathered copies, the gradient's forward substitution and the prediction stream L through again from its owners, Sigma^-1
    is accumulated chunk by chunk.  Against the oracle at the tolerances of the replicated mode; ragged sizes, a separate y
    block row (N % 128 == 0), more ranks than block rows; the same gradient bits on every rank."""
    results = pool.run(_capacity_task, world, N, 3, 333, panel, kind)
    for rank, err_v, err_nl, err_mu, err_var, err_g, err_a, same, no_factor, peak, peak_single, _ in results:
        assert err_v < 1e-10 and err_nl < 1e-8 and err_mu < 1e-8 and err_var < 1e-9 and err_g < 1e-8 and err_a < 1e-8
        assert same and no_factor
    assert len({r[-1] for r in results}) == 1


@pytest.mark.parametrize("world", [2, 3])
def test_capacity_mode_at_n_20k_keeps_each_rank_well_below_a_single_engine(pool, world):
    """N = 20,480 (160 block columns) over two and three ranks sharing the GPU: results against the oracle's NLML and the
    tolerances of the replicated mode, and each rank's PEAK resident bytes (factorisation + prediction + gradient) at most 0.62
    (two ranks) / 0.5 (three) of what a single engine allocates for the same three calls."""
    results = pool.run(_capacity_task, world, 20_480, 4, 1500, 0, "ExpQuad", timeout=1500)
    for rank, err_v, err_nl, err_mu, err_var, err_g, err_a, same, no_factor, peak, peak_single, _ in results:
        assert err_v < 1e-9 and err_nl < 1e-6 and err_mu < 1e-8 and err_var < 1e-8 and err_g < 1e-7 and err_a < 1e-7
        assert same and no_factor
        # (two ranks: half the factor and half of U / Sigma^-1, + three chunk-wide panel buffers since round 5 -- the panel of L
        # that is arriving, the one in use, the chunk of U: 0.604 measured; three ranks: 0.45.  Round 6: the factorisation's TAILs travel
        # through the full-height staging pair the other passes need anyway, its chain through a small second pair)
        assert peak <= (0.62 if world == 2 else 0.5) * peak_single, (peak / 2**30, peak_single / 2**30)
    assert len({r[-1] for r in results}) == 1


def _capacity_guard_task(rank, world, group, out):
    """the single-engine entry points against a capacity-mode factorisation and the reverse mix."""
    from gumbi_amd.distributed import DistributedEngine
    from gumbi_amd.engine import Engine, KernelSpec
    from oracle import gp_oracle as O

    N, d = 1500, 3
    X, y, ls = O.synthetic_table(N, d, seed=11)
    theta = np.concatenate([ls, [1.0, 0.25]])
    kspec = KernelSpec(D=d, idx_cont=list(range(d)), kind="ExpQuad")
    Xs = np.random.default_rng(2).standard_normal((200, d))
    res = {}
    # (a) a full factor left behind by a single-engine factorisation goes when the mode is switched on
    e1 = Engine(0)
    e1.set_data(X, y)
    e1.set_kernel(kspec)
    e1.set_theta(theta)
    e1.factorize()
    before = e1.resident_bytes()
    e1.set_dist_mode(Engine.DIST_CAPACITY)
    res["released"] = before - e1.resident_bytes()
    res["full_buffer"] = 8 * (((N + 1 + 127) // 128) * 128) * (((N + 127) // 128) * 128)
    res["valid_after_switch"] = e1.factor_is_current()
    e1.close()
    eng = DistributedEngine(0, group, panel_blocks=2, capacity=True)
    eng.set_data(X, y)
    eng.set_kernel(kspec)
    eng.set_theta(theta)
    eng.factorize()
    mu, var = eng.predict(Xs)

    def refused(fn):
        try:
            fn()
            return "no error"
        except ValueError as err:
            return str(err)

    # (b) the calls that read the complete factor out of dA: refused, with a message that says why
    res["predict"] = refused(lambda: eng.eng.predict(Xs))
    res["grad"] = refused(lambda: eng.eng.nlml(grad=True))
    res["copy_factor"] = refused(lambda: eng.eng.copy_factor())
    res["value_only"] = eng.eng.nlml()  # (log-det and |v|^2 are replicated scalars: no factor needed)
    mu_b, var_b = eng.predict(Xs)       # ... and the capacity factorisation is still intact
    res["intact"] = bool(mu_b.tobytes() == mu.tobytes() and var_b.tobytes() == var.tobytes())
    # (c) the reverse: a single-engine factorisation, then the capacity passes (their rows would be an older theta's)
    eng.set_theta(theta * 1.05)
    eng.eng.factorize()
    res["cap_grad_after_single"] = refused(lambda: eng.nlml(grad=True))
    res["cap_predict_after_single"] = refused(lambda: eng.predict(Xs))
    mu_s, _ = eng.eng.predict(Xs)       # (the single engine's own prediction works: it made that factor)
    eng.set_theta(theta)
    eng.factorize()
    mu_c, var_c = eng.predict(Xs)
    res["back"] = bool(mu_c.tobytes() == mu.tobytes() and var_c.tobytes() == var.tobytes())
    res["single_differs"] = bool(np.max(np.abs(mu_s - mu)) > 1e-6)
    eng.close()
    out.put((rank, res))


def test_capacity_mode_and_single_engine_calls_do_not_mix_silently(pool):
    """``cap_factorize`` leaves no factor in dA: gmb_predict / gmb_nlml(grad) / gmb_copy_factor on such an engine return
    GMB_EINVAL instead of running on a null or stale buffer, gmb_dist_nlml / gmb_dist_predict in capacity mode refuse a
    factorisation they did not make, and switching the mode on releases an Nr x Np buffer that is already there."""
    for rank, r in pool.run(_capacity_guard_task, 2):
        assert r["released"] >= r["full_buffer"] and not r["valid_after_switch"], r
        for key in ("predict", "grad", "copy_factor"):
            assert "capacity-mode" in r[key], (key, r[key])
        assert np.isfinite(r["value_only"]) and r["intact"]
        for key in ("cap_grad_after_single", "cap_predict_after_single"):
            # (rank 0 reports its own refusal; a later rank may report rank 0's, which it learns of first at the agreement)
            assert "needs a factorisation made by gmb_dist_factorize in capacity mode" in r[key] or (rank > 0 and "failed with status" in r[key]), (key, r[key])
        assert r["back"] and r["single_differs"]


def _notpd_task(rank, world, group, out):
    from gumbi_amd.distributed import DistributedEngine
    from gumbi_amd.engine import Engine, KernelSpec
    from oracle import gp_oracle as O

    N, d = 900, 3
    X, y, ls = O.synthetic_table(N, d, seed=2)
    bad = 517  # in the fifth block row: owned by another rank than block row 0
    X[bad, 1] = np.nan
    theta = np.concatenate([ls, [1.0, 0.2]])
    got = []
    for eng in (DistributedEngine(0, group, panel_blocks=2), Engine(0)):
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


def test_not_positive_definite_is_reported_identically_on_every_rank(pool):
    """A NaN input makes the covariance non-factorisable at its row: every rank of the multi-GPU driver must
    return the SAME failure (GMB_ENOTPD -> LinAlgError, same global row index as the single engine -- the
    diagonal squares are factored redundantly, so no rank can run ahead into a collective the others skip),
    and the engines must remain usable afterwards."""
    results = pool.run(_notpd_task, 3)
    for rank, got in results:
        dist_err, dist_nl, single_err, single_nl = got
        assert dist_err[0] == "LinAlgError" and single_err[0] == "LinAlgError", got
        assert dist_err[1] == single_err[1] == 517, got
        assert dist_nl == single_nl
    assert len({str(g) for _, g in results}) == 1


def _local_failure_task(rank, world, group, out, mode):
    """One rank fails LOCALLY; no rank may be left blocked in an all-gather, and every rank must get an error."""
    from gumbi_amd import engine as E
    from gumbi_amd.distributed import DistributedEngine, TorchDistComm
    from oracle import gp_oracle as O

    N, d = 1500, 3
    X, y, ls = O.synthetic_table(N, d, seed=11)
    theta = np.concatenate([ls, [1.0, 0.2]])

    class FlakyComm(TorchDistComm):
        """performs every collective (the peers are never blocked) but reports a transport error for the k-th"""

        calls = 0

        def _all_gather(self, ctx, send, recv, count, stream):
            rc = super()._all_gather(ctx, send, recv, count, stream)
            FlakyComm.calls += 1
            return -7 if (self.rank == 1 and FlakyComm.calls == 6) else rc

    comm = FlakyComm(0, group) if mode == "mid-sequence" else None
    eng = DistributedEngine(0, group, comm=comm, panel_blocks=2)
    eng.set_data(X, y)
    eng.set_kernel(E.KernelSpec(D=d, idx_cont=list(range(d))))
    if not (mode == "set-up" and rank == 1):
        eng.set_theta(theta)  # rank 1 "forgets" its hyper-parameters: its gmb_dist_factorize fails before any collective
    try:
        eng.factorize()
        got = "no error"
    except (ValueError, E.GumbiHipError) as err:
        got = f"{type(err).__name__}: {err}"
    # after the failure every rank is still in step: the same engines factorise a well-posed problem together
    eng.set_theta(theta)
    if comm is not None:
        FlakyComm.calls = 1000
    eng.factorize()
    val = eng.nlml()
    ref = O.nlml(O.make_spec(d, range(d)), theta, X, y)
    eng.close()
    out.put((rank, got, abs(val - ref) / abs(ref)))


@pytest.mark.parametrize("mode", ["set-up", "mid-sequence"])
def test_a_rank_that_fails_locally_takes_every_rank_to_an_error_and_blocks_nobody(pool, mode):
    """ADVICE r02 (dist_driver.hpp): a failure local to one rank -- a call-order / allocation error before the
    first collective, or an error in the middle of the panel loop -- used to make that rank return between
    collectives while its peers blocked in the next all-gather for ever.  Now the ranks agree on a status before
    the first data collective and after the last, and a failing rank keeps issuing the plan's all-gathers."""
    results = pool.run(_local_failure_task, 2, mode, timeout=200)
    by_rank = {r[0]: r for r in results}
    assert "no error" not in (by_rank[0][1], by_rank[1][1]), results
    if mode == "set-up":
        assert "gmb_set_theta has not been called" in by_rank[1][1]
    else:
        assert "transport status -7" in by_rank[1][1]
    assert "rank 1 of 2 failed" in by_rank[0][1], by_rank[0][1]
    assert by_rank[0][2] < 1e-10 and by_rank[1][2] < 1e-10  # both recovered together


def _large_task(rank, world, group, out, N, d):
    from gumbi_amd.distributed import DistributedEngine
    from gumbi_amd.engine import Engine, KernelSpec
    from oracle import gp_oracle as O

    X, y, ls = O.synthetic_table(N, d, seed=7)
    theta = np.concatenate([ls, [1.0, 0.2]])
    kspec = KernelSpec(D=d, idx_cont=list(range(d)), kind="ExpQuad")
    Xs = O.synthetic_grid(d, 24)
    res = {}
    for name, eng in (("dist", DistributedEngine(0, group)), ("single", Engine(0))):
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


def test_two_ranks_one_gpu_at_n_20k_match_the_single_engine(pool):
    """The block-cyclic driver well beyond a handful of blocks: N = 20,480 (160 block columns, 20 panels of
    the default width, 5 chunks of the row-partitioned inverse) over two ranks sharing the GPU: sampled rows
    of the factor, v, predictions, NLML and gradient equal the single-GPU engine's (same kernels, different
    order of the trailing updates), the NLML equals the oracle's, both ranks hold identical bits."""
    results = pool.run(_large_task, 2, 20_480, 4, timeout=900)
    for rank, errs, err_oracle, _ in results:
        assert max(errs[:2]) < 1e-9 and errs[2] < 1e-8 and errs[3] < 1e-8, errs   # factor rows, v, mean, variance
        assert errs[4] < 1e-11 and errs[5] < 1e-7, errs                            # NLML, gradient
        assert err_oracle < 1e-10
    assert results[0][-1] == results[1][-1]


def _fit_task(rank, world, group, out, N, d, ls_lower):
    import pandas as pd

    import gumbi_amd as gmb
    from oracle import gp_oracle as O

    X, y, ls = O.synthetic_table(N, d, seed=9)
    cols = [f"x{k}" for k in range(d)]
    df = pd.DataFrame(X, columns=cols)
    df["y"] = y
    grid = np.random.default_rng(2).uniform(-1.5, 1.5, (400, d))
    # the generator's noise-free function at the grid, in the z-scored units of y
    rng = np.random.default_rng(9)
    Xr = rng.standard_normal((N, d))
    y_raw = np.sum(np.sin(Xr / ls), axis=1) / np.sqrt(d) + 0.2 * rng.standard_normal(N)
    f_grid = (np.sum(np.sin(grid / ls), axis=1) / np.sqrt(d) - y_raw.mean()) / y_raw.std(ddof=1)

    def fit(distributed):
        ds = gmb.DataSet(df, outputs=["y"])
        gp = gmb.GP(ds, outputs=["y"], distributed=distributed)
        trace_dir = os.environ.get("GUMBI_TEST_TRACE")
        if trace_dir and distributed is not None:  # debugging aid: what every rank was given and returned, per evaluation
            import hashlib

            fh = open(os.path.join(trace_dir, f"fit_N{N}_rank{rank}.log"), "w")
            orig = gp._objective

            def traced(u, pos):
                f, g = orig(u, pos)
                fh.write(f"{hashlib.sha1(np.asarray(u).tobytes()).hexdigest()[:10]} {float(f)!r} {hashlib.sha1(np.asarray(g).tobytes()).hexdigest()[:10]}\n")
                fh.flush()
                return f, g

            gp._objective = traced
        lsb = gmb.make_deltas_parray(stdzr=ds.stdzr, scale="standardized", **{c: [ls_lower, None] for c in cols}) if ls_lower else None
        gp.fit(continuous_dims=cols, ls_bounds=lsb)  # the user-level call: specify_model + build_model + find_MAP
        pred = gp.predict_points(gp.parray(**{c: grid[:, k] for k, c in enumerate(cols)}))
        mu_z = gp.predict(np.asarray(grid, float))[0]
        res = (gp._theta_fitted.copy(), np.asarray(pred.μ).copy(), np.asarray(pred.σ2).copy(), gp.n_eval,
               float(np.corrcoef(mu_z, f_grid)[0, 1]), float(np.asarray(gp.MAP["σ"])) / (0.2 / y_raw.std(ddof=1)))
        gp.engine.close()
        return res

    th_d, mu_d, var_d, n_d, corr_d, sig_d = fit(group)
    th_s, mu_s, var_s, n_s, corr_s, sig_s = fit(None)  # the same fit on this rank's GPU alone
    out.put((rank, float(np.max(np.abs(th_d - th_s) / np.maximum(np.abs(th_s), 1e-3))),
             float(np.max(np.abs(mu_d - mu_s))), float(np.max(np.abs(var_d - var_s))), n_d, n_s, corr_d, sig_d, th_d.tobytes()))


def test_distributed_map_fit_matches_single_gpu_fit(pool):
    """GP(..., distributed=group).fit() over two ranks: same optimum and predictions as the
    single-GPU fit, identical parameters on both ranks."""
    results = pool.run(_fit_task, 2, 400, 2, None)
    for rank, err_th, err_mu, err_var, n_d, n_s, corr, sig, _ in results:
        assert err_th < 1e-5 and err_mu < 1e-6 and err_var < 1e-6
    assert results[0][-1] == results[1][-1]


def test_distributed_map_fit_at_n_8k_finds_the_function(pool):
    """``GP(distributed=group).fit(ls_bounds=...)`` at N = 8192, d = 4 over two ranks (64 block columns, 8 panels,
    two chunks of the row-partitioned inverse, ~25 lock-step L-BFGS-B evaluations): the optimiser must arrive
    where the single-GPU fit arrives, both ranks with identical bits, and the fit must be a FIT -- posterior
    mean correlated > 0.95 with the generator's noise-free function, sigma-hat within 10 % of the generator's."""
    results = pool.run(_fit_task, 2, 8192, 4, 0.5, timeout=900)
    for rank, err_th, err_mu, err_var, n_d, n_s, corr, sig, _ in results:
        assert err_th < 2e-2 and err_mu < 1e-3 and err_var < 1e-3, (err_th, err_mu, err_var)
        assert 10 <= n_d <= 80 and corr > 0.95 and abs(sig - 1.0) < 0.1, (n_d, corr, sig)
    assert results[0][-1] == results[1][-1]


def _cv_task(rank, world, group, out):
    import pandas as pd

    import gumbi_amd as gmb
    from oracle import gp_oracle as O

    X, y, _ = O.synthetic_table(300, 2, seed=4)
    df = pd.DataFrame(X, columns=["a", "b"])
    df["y"] = y
    gp = gmb.GP(gmb.DataSet(df, outputs=["y"]), outputs=["y"])
    gp.specify_model(continuous_dims=["a", "b"])
    gp.build_model()
    res = gp.cross_validate_replicas([11, 12, 13], group=group, pct_train=0.7)
    out.put((rank, [float(r["test"]["NLPDs"].mean()) for r in res], [len(r["train"]["data"].wide) for r in res]))


def test_cross_validation_replicas_over_ranks(pool):
    """Three independent splits dealt over two ranks: both ranks end up with all three results, equal
    to what one process computes alone."""
    import pandas as pd

    import gumbi_amd as gmb
    from oracle import gp_oracle as O

    results = pool.run(_cv_task, 2)
    assert results[0][1:] == results[1][1:]
    X, y, _ = O.synthetic_table(300, 2, seed=4)
    df = pd.DataFrame(X, columns=["a", "b"])
    df["y"] = y
    gp = gmb.GP(gmb.DataSet(df, outputs=["y"]), outputs=["y"])
    gp.specify_model(continuous_dims=["a", "b"])
    gp.build_model()
    alone = gp.cross_validate_replicas([11, 12, 13], pct_train=0.7)
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
        assert eng.comm.ranks == 1  # ncclCommCount of the library's own communicator
        # the collectives of the communication stream (a panel column's TAIL, the gradient's chunks) get a communicator of their own,
        # split off the first: RCCL runs ONE communicator's collectives in issue order whatever their streams
        assert eng.comm.two_communicators is True
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
        tm = eng.timings()
        assert tm["dist_world"] == 1 and tm["dist_chol_collectives"] > 0 and tm["dist_lockstep_repairs"] == 0
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


@pytest.mark.parametrize("launcher,map_evals", [("torchrun", 0), ("self", 0), ("self", 3), ("self-capacity", 0)])
def test_bench_contract_with_two_ranks_on_one_gpu(gpu, launcher, map_evals):
    """bench.py as the driver launches it for N > 1 (``python -m torch.distributed.run --nproc-per-node N
    bench.py --gpus N ...``) and, ``launcher == "self"``, as plain ``python bench.py --gpus 2`` with NO launcher around it
    (bench.py then re-runs itself under torch.distributed.run), with both ranks on the test box's one
    GPU over gloo: rank 0 prints exactly ONE
    JSON line with the contract's keys, n_gpus = 2, the workload is ONE GP partitioned over the ranks (fixed
    hyper-parameters, or a distributed ``find_MAP(maxeval=k)`` with ``--map-evals k``), and the line says what the
    collectives cost (``comm``) and what the same step does on ONE GPU (``strong_scaling_base_gflops``)."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, GUMBI_BENCH_SINGLE_DEVICE="1", GUMBI_BENCH_BACKEND="gloo", GUMBI_BENCH_DIST_N="3000",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    head = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
            "127.0.0.1", "--master-port", str(_free_port())] if launcher == "torchrun" else [sys.executable]
    capacity = launcher.endswith("capacity")  # (--capacity: the same step with no rank holding the factor)
    cmd = (head + [str(root / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"] + (["--map-evals", str(map_evals)] if map_evals else [])
           + (["--capacity"] if capacity else []))
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert "error" not in d, d
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "comm", "transport", "rccl_ranks", "comm_ms_total",
                "comm_ms_exposed", "strong_scaling_base_gflops", "speedup_over_one_gpu"):
        assert key in d, key
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong" and d["dtype"] == "f64"
    assert d["launcher"].startswith("bench.py itself" if launcher.startswith("self") else "external")
    assert d["config"]["factor_storage"].startswith("capacity mode" if capacity else "replicated")
    assert d["value"] > 0 and d["results_finite"] and "cpu_baseline" not in d
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert "ONE GP" in d["config"]["workload"] and "over 2 GPUs" in d["config"]["parallelism"]
    assert "torch-gloo" in d["config"]["parallelism"] and d["config"]["N"] == 3000
    # communication of the last factorisation + gradient: counted, timed, and never more exposed than spent
    c = d["comm"]
    assert c["transport"] == d["transport"] == "torch-gloo" and c["world"] == 2 and c["rccl_ranks"] is None
    assert c["factorize"]["collectives"] > 0 and c["factorize"]["received_GB"] > 0 and c["gradient"]["collectives"] >= 3
    assert 0 <= c["comm_ms_exposed"] <= c["comm_ms_total"] * (1 + 1e-9) and c["lockstep_repairs"] == 0
    assert d["comm_ms_total"] == c["comm_ms_total"]
    # the same step on one GPU, by rank 0 after the timed region
    base = d["strong_scaling_base"]
    assert "error" not in base and base["results_finite"] and base["value"] == d["strong_scaling_base_gflops"] > 0
    assert abs(d["speedup_over_one_gpu"] - d["value"] / base["value"]) < 2e-3
    # the line validates itself against that one-GPU run and carries its point of the C5 curve under one name
    assert d["results_match_one_gpu"] is True, d["results_match_note"]
    assert 0 <= d["nlml_rel_diff_vs_one_gpu"] <= (1e-6 if map_evals else 1e-10)
    assert d["mean_max_rel_diff_vs_one_gpu"] <= (1e-5 if map_evals else 1e-8) and d["var_max_abs_diff_vs_one_gpu"] <= (1e-6 if map_evals else 1e-9)
    assert d["scale_point"]["config"] == "c5" and d["scale_point"]["n_gpus"] == 2 and d["scale_point"]["gflops"] == d["value"]
    ph = d["phases"]
    N, M = 3000, 10_000
    if map_evals:
        assert "find_MAP(maxeval=3)" in d["metric"]
        assert ph["find_map_s"] > 0 and ph["predict_s"] > 0 and all(1 <= n <= 8 for n in ph["map_evals_per_step"])
        assert ph["refactorizations_at_the_map_per_step"] == [1, 1]  # the distributed gradient consumes the factor
        flops = sum(n * float(N) ** 3 + float(N) ** 3 / 3 + float(N) ** 2 * M + 4.0 * N * M for n in ph["map_evals_per_step"])
    else:
        assert ph["map_eval_s"] > 0 and ph["fit_fixed_theta_s"] > 0 and ph["predict_s"] > 0 and np.isfinite(ph["nlml"])
        flops = 2 * (float(N) ** 3 + float(N) ** 3 / 3 + float(N) ** 2 * M + 4.0 * N * M)
    # whole-job value = algorithmic flops of the steps / max-over-ranks wall time
    assert abs(d["value"] - flops / (d["ms_per_step"] * 2e-3) / 1e9) / d["value"] < 1e-3


def test_bench_contract_single_process(gpu):
    """``python bench.py --config c2`` (N = 1): one JSON line with the contract's keys plus ``roofline`` (the
    trailing update alone), ``cpu_baseline`` (here on a shortened sample), the per-phase timings, the
    end-to-end fit and the one-GP section (shrunk by the test hook)."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    # the bench runs alone on a box: retire the worker pool first -- four more processes with HIP contexts on the one GPU
    # cost the register-only MFMA ceiling a quarter of its rate (77.8 -> 57 TF/s with the pool idle, back to 77.6 once it
    # is closed; GUMBI_TEST_CEILING=1 prints the figure after every test)
    if _POOL[0] is not None:
        _POOL[0].close()
        _POOL[0] = None
    root = Path(__file__).resolve().parent.parent
    # (the host's config-size sections -- minutes of CPU time: a whole C2 fit over the oracle, a 50k dpotrf -- are switched off here)
    env = dict(os.environ, GUMBI_BENCH_DIST_N="2304", GUMBI_BENCH_C4_N="1536", GUMBI_BENCH_CPU_SECONDS="2", GUMBI_BENCH_NO_CPU_CONFIG_SIZE="1")
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
    # C2's evaluations are ONE launch of the persistent evaluation kernel each (tile Cholesky + inverse + Sigma^-1), and that
    # launch is the roofline's kernel
    assert "eval_tiles_kernel" in r["kernel"] and 15.0 < r["achieved"] < r["peak"] and 1 <= r["launches"] <= 16
    assert 15.0 < r["achieved_on_unpadded_N3"] <= r["achieved"]
    assert r["all_gemm_launches"]["launches"] > r["launches"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "N=" in c["sample"]
    assert d["phases"]["factorize_ms"] > 0 and d["phases"]["predict_ms"] > 0
    assert d["kbuild"]["unit"] == "GB/s" and d["kbuild"]["achieved"] > 100
    assert d["end_to_end"]["results_finite"] and d["end_to_end"]["total_s"] > 0
    side = d["c5_single_gpu"]
    assert "error" not in side, side
    assert side["results_finite"] and side["value"] > 0 and "N=2304" in side["workload"]
    assert d["strong_scaling_base_gflops"] == side["value"]
    assert d["scale_point"] == dict(d["scale_point"], config="c5", n_gpus=1, gflops=side["value"]) and side["warmup"] == 0
    assert set(d["section_seconds"]) >= {"cpu_baseline", "c4_single_gpu", "c5_single_gpu", "end_to_end", "default_start"}
    # what the timed fit found (capped at 4 evaluations here, so no quality bar -- only that it is reported)
    q = d["fit_quality"]
    assert set(q) >= {"corr", "rmse", "sigma_hat", "sigma_true", "nlml_final", "n_eval", "converged"} and q["n_eval"] == d["config"]["map_evals_per_step"][0]
    assert "ls_bounds" in d["config"] and "lower = 0.5" in d["config"]["ls_bounds"]
    ce = r["mfma_only_ceiling"]
    assert ce["before_timed_region"]["tflops_mean"] > 50 and ce["after_timed_region"]["tflops_mean"] > 50
    assert ce["before_timed_region"]["seconds"] >= 1.0 and 1500 < ce["after_timed_region"]["shader_mhz"] < 2600
    assert r["achieved_le_ceiling"] is True
    assert d["default_start"]["fit_quality"]["n_eval"] >= 1 and d["bench_wall_s"] > 0
    assert "restatement" in d["default_start"]["note"].lower() and "1e-3" in d["parity_pin"] and "1e-8" in d["parity_pin"]
    assert d["roofline_frac_dominant_kernel"] == r["frac"] and abs(d["roofline_frac_whole_step"] - d["value"] / 1e3 / r["peak"]) < 1e-3
    # BASELINE configs[3] as a side figure: the two-output GP through the Kronecker path, the stacked engine beside it
    c4 = d["c4_single_gpu"]
    assert "error" not in c4 and "skipped" not in c4, c4
    assert c4["results_finite"] and c4["seconds_per_map_evaluation"] > 0 and "N=1536" in c4["workload"] and "CAPPED" in c4["step"]
    assert "eval_tiles_kernel" in c4["roofline"]["kernel"] and 0 < c4["roofline"]["frac"] < 1
    assert c4["phases"]["stacked_map_evaluation_ms"] > 0 and c4["phases"]["kronecker_map_evaluation_ms"] > 0


def test_bench_config_c4_is_the_kronecker_fit(gpu):
    """``python bench.py --config c4`` (BASELINE.json configs[3], shrunk by the test hook): the two-output coregionalised GP
    declared through DataSet -> GP(outputs=[y0, y1]) -> build_model, fitted through the Kronecker engine HipGP chooses, predicted
    on the grid of both outputs; roofline = the P N x N evaluation launches; cpu_baseline = the oracle's STACKED evaluation."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    if _POOL[0] is not None:
        _POOL[0].close()
        _POOL[0] = None
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, GUMBI_BENCH_C4_N="1536", GUMBI_BENCH_CPU_SECONDS="2")
    out = subprocess.run([sys.executable, str(root / "bench.py"), "--config", "c4", "--steps", "1", "--warmup", "1", "--map-evals", "12"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["dtype"] == "f64" and d["results_finite"]
    assert "Kronecker" in d["metric"] and d["config"]["outputs"] == 2 and d["config"]["N"] == 1536 and d["config"]["M"] == 20_000
    n_eval = d["config"]["map_evals_per_step"][0]
    assert 1 <= n_eval <= 20 and d["fit_quality"]["n_hyperparameters"] == 18
    flops = 2 * (n_eval * 1536.0**3 + d["config"]["refactorizations_at_the_map_per_step"][0] * 1536.0**3 / 3 + 1536.0**2 * 20_000 + 4.0 * 1536 * 20_000)
    assert abs(d["value"] - flops / (d["ms_per_step"] * 1e-3) / 1e9) / d["value"] < 1e-3
    r = d["roofline"]
    assert "eval_tiles_kernel" in r["kernel"] and r["launches"] >= 2 and 0 < r["frac"] < 1
    ph = d["phases"]
    assert ph["stacked_map_evaluation_ms"] > 0 and ph["kronecker_map_evaluation_ms"] > 0 and ph["stacked_over_kronecker"]["map_evaluation"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and "stacked" in c["sample"] and c["value"] > 0
    assert "c5_single_gpu" not in d and "1e-3" in d["parity_pin"]


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


def test_bench_gpus_2_over_rccl_on_a_one_gpu_box_fails_loudly(gpu):
    """``python bench.py --gpus 2`` on the production transport (nccl = RCCL, one rank per GPU) needs two devices: on a box
    with fewer it exits non-zero with an error line naming the shortfall, whether bench.py launches the ranks itself or a
    launcher already did -- it never prints a one-GPU line for a multi-GPU request."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    if gpu >= 2:
        pytest.skip("this box has two devices: the request can be served")
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "GUMBI_BENCH_SINGLE_DEVICE", "GUMBI_BENCH_BACKEND")}
    for head in ([sys.executable],
                 [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                  "--master-port", str(_free_port())]):
        out = subprocess.run(head + [str(root / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=root, env=env,
                             capture_output=True, text=True, timeout=300)
        assert out.returncode != 0, out.stdout[-500:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, (out.stdout[-1000:], out.stderr[-1000:])
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["value"] is None and "need" in d["error"]["error"] and "this node shows 1" in d["error"]["error"]
