"""Host-side pieces of bench.py that need no GPU: the flop count of a step, the synthetic tables
(SURVEY.md section 8d recipe, shared with the oracle) and the deadline guard around the side
measurement (a stuck collective must not take the headline JSON line with it)."""
import importlib.util
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["bench_module"] = mod
    spec.loader.exec_module(mod)
    return mod


def test_deadline_guard_returns_result_error_or_timeout():
    bench = load_bench()
    val, ok = bench.run_with_deadline(lambda: {"x": 1}, 5.0)
    assert ok and val == {"x": 1}

    def boom():
        raise RuntimeError("peer died")

    val, ok = bench.run_with_deadline(boom, 5.0)
    assert ok and "RuntimeError: peer died" in val["error"]
    t0 = time.perf_counter()
    val, ok = bench.run_with_deadline(lambda: time.sleep(3.0), 0.2)
    assert not ok and "no result after" in val["error"] and time.perf_counter() - t0 < 2.0


def test_synthetic_tables_follow_the_survey_recipe():
    bench = load_bench()
    from oracle import gp_oracle as O

    X, y, ls = bench.synthetic_table(500, 4)
    Xo, yo, lso = O.synthetic_table(500, 4)
    assert np.array_equal(X, Xo) and np.array_equal(y, yo) and np.array_equal(ls, lso)
    assert abs(y.mean()) < 1e-12 and abs(y.std(ddof=1) - 1.0) < 1e-12
    assert np.allclose(ls, np.geomspace(0.7, 2.0, 4))
    G = bench.synthetic_grid(4, 100)
    assert G.shape == (10_000, 4) and np.all(G[:, 2:] == 0.0) and G[:, 0].min() == -2.4 and G[:, 1].max() == 2.4
    assert np.array_equal(G, O.synthetic_grid(4, 100))


def test_step_flops_counts_cholesky_inverse_and_prediction():
    bench = load_bench()
    N, M, n_eval = 10_000, 10_000, 70
    per_eval = N**3 / 3 + 2 * N**3 / 3                  # Cholesky + (L^-1 and Sigma^-1), SURVEY 8d
    final = N**3 / 3 + float(N) ** 2 * M                # re-factorisation + N^2 M solve of the grid prediction
    got = bench.step_flops(N, M, n_eval)
    assert abs(got - (n_eval * per_eval + final)) / got < 0.02
    # no re-factorisation at the MAP when the optimiser ended on an evaluated point: one Cholesky less
    assert abs(bench.step_flops(N, M, n_eval, 0) - (got - N**3 / 3)) / got < 1e-12


def test_roofline_block_is_the_trailing_update_alone():
    """``roofline`` = algorithmic flops of the Cholesky's trailing-update launches / their summed launch
    durations (frac = achieved / peak), with the all-launch figure beside it; ``kbuild`` in GB/s."""
    bench = load_bench()
    tm = {"total_chol_gemm_flops": 4.0e13, "total_chol_gemm_ms": 600.0, "total_chol_gemm_launches": 400,
          "total_chol_gemm_wall_ms": 580.0, "total_gemm_flops": 1.6e14, "total_gemm_ms": 2500.0, "total_gemm_launches": 4000,
          "total_gemm_wall_ms": 2400.0, "masked_gemm_flops": 0.0, "total_kbuild_bytes": 1.0e10, "total_kbuild_ms": 3.0,
          "total_kbuild_launches": 1}
    r = bench.roofline_block(tm, "no-such-config")
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 78.6
    assert abs(r["achieved"] - 4.0e13 / 0.6 / 1e12) < 1e-3 and abs(r["frac"] - r["achieved"] / 78.6) < 1e-4
    assert r["launches"] == 400 and abs(r["avg_launch_ms"] - 1.5) < 1e-9 and r["flops_per_launch"] == 1.0e11
    assert r["traffic"] is None  # no PMC summary committed for that name
    assert abs(r["all_gemm_launches"]["achieved"] - 64.0) < 1e-9
    k = bench.kbuild_block(tm)
    assert k["bound"] == "hbm" and k["unit"] == "GB/s" and abs(k["achieved"] - 1.0e10 / 3.0e-3 / 1e9) < 0.1


def test_configs_name_the_baseline_workloads():
    bench = load_bench()
    assert bench.CONFIGS["c3"]["N"] == 50_000 and bench.CONFIGS["c3"]["d"] == 8 and bench.CONFIGS["c3"]["kernel"] == "Matern52"
    assert bench.CONFIGS["c2"]["N"] == 10_000 and bench.CONFIGS["c2"]["d"] == 4
    assert bench.CONFIGS["c5"]["d"] == 8 and bench.CONFIGS["c5"]["kernel"] == "ExpQuad"
    assert bench.CONFIGS["c4"]["N"] == 20_000 and bench.CONFIGS["c4"]["d"] == 4 and bench.CONFIGS["c4"]["P"] == 2


def test_c4_table_is_the_survey_generator_and_its_flops_are_the_kronecker_forms():
    """bench.py's C4 table = the generator tests/test_gpu_configs.py::icm_problem restates (same draws, same stacking), so the
    bench times the problem the parity tests check; flops of a step = P systems of N^3 per evaluation."""
    sys.path.insert(0, str(ROOT / "tests"))
    from test_gpu_configs import icm_problem

    bench = load_bench()
    cfg = dict(bench.CONFIGS["c4"], N=300)
    Xc, Y, ls, Xg, Fg = bench.c4_table(cfg)
    X, y, _spec, _theta = icm_problem(300, 4)
    assert np.array_equal(X[:300, :4], Xc) and np.array_equal(X[300:, :4], Xc) and np.array_equal(X[:, 4], np.repeat([0.0, 1.0], 300))
    z = np.concatenate([(Y[p] - Y[p].mean()) / Y[p].std(ddof=1) for p in range(2)])
    assert np.allclose(z, y, rtol=0, atol=1e-14)
    assert Xg.shape == (10_000, 4) and Fg.shape == (2, 10_000) and np.all(np.isfinite(Fg))
    assert bench.c4_step_flops(1000, 2, 500, 3, 1) == 2 * (3 * 1e9 + 1e9 / 3 + 1e6 * 500 + 4.0 * 1000 * 500)
    r = bench.cpu_baseline_c4(dict(cfg, N=256, res=8), target_seconds=0.01)
    assert r["kind"] == "port" and r["value"] > 0 and "256 rows x 2 outputs" in r["sample"]


def test_truth_matches_the_table_and_fit_quality_sees_a_fit(monkeypatch):
    """``fit_quality`` -- what the bench line says about the fit it timed -- on the numpy oracle behind the same host
    code: the declared model (``ls_bounds`` through ``make_deltas_parray``) converges to the generator's function and
    noise level on a small table; the PyMC-default declaration on the same table is reported for what it is."""
    sys.path.insert(0, str(ROOT / "tests"))
    from gumbi_amd.regression import hip_gp
    from oracle_engine import OracleEngine

    bench = load_bench()
    monkeypatch.setattr(hip_gp, "Engine", OracleEngine)
    cfg = dict(bench.CONFIGS["c2"], N=400, d=2, res=20)
    # the truth is the generator's noise-free function in the z-scored units of the table's y
    X, y, ls = bench.synthetic_table(cfg["N"], cfg["d"])
    f_grid, sigma_z = bench.synthetic_truth(cfg)
    assert f_grid.shape == (400,) and 0.2 < sigma_z < 0.45
    resid = y - (np.sum(np.sin(X / ls), axis=1) / np.sqrt(cfg["d"]) - 0.0) / (0.2 / sigma_z)
    assert abs(np.std(resid - resid.mean(), ddof=1) - sigma_z) < 0.03  # what is left of y after f is the declared noise
    gp = bench.build_gp(cfg, device=0)  # ls_bounds lower = LS_LOWER_Z
    assert np.all(gp._initial_theta()[:2] > 1.0)
    gp.find_MAP()
    mu, _ = gp.predict(bench.synthetic_grid(cfg["d"], cfg["res"]))
    q = bench.fit_quality(gp, mu, cfg)
    assert q["converged"] and q["corr"] > 0.95 and q["sigma_rel_err"] < 0.15 and q["rmse"] < 0.5 * q["rmse_of_predicting_zero"], q
    assert q["n_eval"] == gp.n_eval and len(q["ls"]) == 2
    # a mean that ignores the data is reported as such
    q0 = bench.fit_quality(gp, np.zeros_like(mu), cfg)
    assert q0["corr"] == 0.0 and abs(q0["rmse"] - q0["rmse_of_predicting_zero"]) < 1e-12


def test_time_budget_skips_side_sections_but_says_so(monkeypatch):
    bench = load_bench()
    monkeypatch.setenv("GUMBI_BENCH_BUDGET_S", "100")
    b = bench.Budget()
    monkeypatch.setattr(bench, "T_PROCESS_START", time.perf_counter() - 70.0)
    assert b.allows(20.0) and not b.allows(40.0)
    msg = b.skipped(40.0)["skipped"]
    assert "time budget" in msg and "100" in msg and "GUMBI_BENCH_BUDGET_S" in msg


def test_pmc_traffic_is_marked_stale_when_the_kernels_changed(tmp_path, monkeypatch):
    """``roofline.traffic`` comes from a committed counter summary; the summary carries the hash of the kernel sources it
    was taken on (tools/gpu_pmc_bench.sh) and the bench says when that is not the tree's."""
    bench = load_bench()
    prof = tmp_path / "profiles"
    prof.mkdir()
    (tmp_path / "gumbi_amd" / "csrc").mkdir(parents=True)
    (tmp_path / "gumbi_amd" / "csrc" / "k.hpp").write_text("kernel v1")
    (prof / "r99_pmc_bench_cx_summary.csv").write_text(
        "Kernel,Launches,TotalMs(pass1),MfmaUtil%,MFMA_F64_TFLOPs,FetchGB(x2 corrected),FetchGB/s,WriteGB(raw),WriteGB/s\n"
        '"void gmb::gemm_f64_kernel<2, 2, 4, 4, 2, false>(gmb::GemmArgs)",10,100.0,90.0,70.0,50.0,500.0,10.0,100.0\n')
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    now = bench.kernel_sources_sha16()
    (prof / "r99_pmc_bench_cx_summary.meta.json").write_text('{"kernel_sources_sha16": "%s"}' % now)
    pt = bench.pmc_traffic("cx", "gemm_f64_kernel<2, 2, 4, 4")
    assert pt["launches"] == 10 and abs(pt["bytes_per_launch"] - 6.0e9) < 1 and pt["stale"] is False
    (tmp_path / "gumbi_amd" / "csrc" / "k.hpp").write_text("kernel v2")
    assert bench.pmc_traffic("cx", "gemm_f64_kernel<2, 2, 4, 4")["stale"] is True


def test_roofline_block_quotes_the_tile_kernel_when_the_factorisation_is_one_launch():
    """C2-class sizes: no trailing-update launches exist, the persistent tile kernel is the roofline's kernel."""
    bench = load_bench()
    tm = {"total_chol_gemm_flops": 0.0, "total_chol_gemm_ms": 0.0, "total_chol_gemm_launches": 0, "total_chol_gemm_wall_ms": 0.0,
          "total_gemm_flops": 6.0e12, "total_gemm_ms": 100.0, "total_gemm_launches": 300, "total_gemm_wall_ms": 90.0,
          "total_chol_panel_gemm_flops": 0.0, "total_chol_panel_gemm_ms": 0.0, "masked_gemm_flops": 0.0,
          "total_chol_tile_ms": 20.0, "total_chol_tile_flops": 1.0e12, "total_chol_tile_launches": 3}
    r = bench.roofline_block(tm, "no-such-config")
    assert "chol_tiles_kernel" in r["kernel"] and r["launches"] == 3 and abs(r["achieved"] - 50.0) < 1e-9
    assert abs(r["frac"] - 50.0 / bench.FP64_MFMA_PEAK_TFLOPS) < 1e-4 and r["traffic"] is None and "in_panel_products" not in r
    # with the committed counter summary of the C2 bench: HBM-side bytes per factorisation launch of that kernel, as counted
    r = bench.roofline_block(tm, "c2")
    pt = bench.pmc_traffic("c2", "chol_tiles_kernel<8>")
    assert pt is not None and r["traffic"] == round(pt["bytes_per_launch"], 1) and 4.0e8 < r["traffic"] < 1.0e11
    assert r["traffic_source"].startswith("profiles/") and r["traffic_is_stale"] == pt["stale"]


def test_roofline_block_quotes_the_evaluation_launch_when_the_gradient_rides_with_the_factorisation():
    """C2-class sizes under find_MAP: one persistent launch per evaluation (csrc/eval_tiles.hpp) is the roofline's kernel; the
    same time also priced on the unpadded N^3."""
    bench = load_bench()
    tm = {"total_chol_gemm_flops": 0.0, "total_chol_gemm_ms": 0.0, "total_chol_gemm_launches": 0, "total_chol_gemm_wall_ms": 0.0,
          "total_gemm_flops": 6.0e12, "total_gemm_ms": 100.0, "total_gemm_launches": 300, "total_gemm_wall_ms": 90.0,
          "total_chol_panel_gemm_flops": 0.0, "total_chol_panel_gemm_ms": 0.0, "masked_gemm_flops": 0.0,
          "total_chol_tile_ms": 7.0, "total_chol_tile_flops": 3.4e11, "total_chol_tile_launches": 1,
          "total_eval_tile_ms": 64.0, "total_eval_tile_flops": 4.0 * 1.034e12, "total_eval_tile_launches": 4}
    r = bench.roofline_block(tm, "c2")
    assert "eval_tiles_kernel" in r["kernel"] and r["launches"] == 4 and abs(r["achieved"] - 4.0 * 1.034e12 / 64.0 / 1e9) < 1e-3
    assert abs(r["achieved_on_unpadded_N3"] - 4.0e12 / 64.0 / 1e9) < 1e-3 and r["achieved_on_unpadded_N3"] < r["achieved"]
    assert abs(r["frac"] - r["achieved"] / bench.FP64_MFMA_PEAK_TFLOPS) < 1e-4 and "in_panel_products" not in r


def test_tile_trace_summary_reads_the_chain_off_the_stamps():
    """summarize_tile_trace: chain step = distance between the publications of consecutive diagonal tiles; k-block time from
    the bulk tiles (I >= J + 2) of the columns with at least 8 k-blocks."""
    bench = load_bench()
    nct = 12
    tiles, stamps = [], []
    for j in range(nct):
        for i in range(j, nct):
            tiles.append((i, j))
            t0 = 50e-6 * j
            ksum = 16e-6 * j
            stamps.append((t0, t0 + ksum, t0 + ksum + 1e-6, 50e-6 * (j + 1) + (0 if i == j else 20e-6)))
    out = bench.summarize_tile_trace(np.array(tiles), np.array(stamps))
    assert out["block_columns"] == nct and out["tasks"] == len(tiles)
    assert abs(out["chain_step_us"]["median"] - 50.0) < 1e-6 and abs(out["contraction_us_per_k_block"]["median"] - 16.0) < 1e-6



def test_schedule_model_orders_are_topological_and_column_major_is_not_beaten():
    """tools/ct_schedule_sim.py: the model behind the decision to keep the column-major ticket order -- every
    candidate order it compares is a topological order of the tile graph (simulate asserts it), chunked orders hold every
    k-block of every tile exactly once, and none of them beats column-major by more than a few per cent at a C2-class size."""
    sys.path.insert(0, str(ROOT / "tools"))
    import ct_schedule_sim as S

    nct = 30
    base, occ = S.simulate(nct, S.column_major(nct))
    assert nct * 40.0 < base < 2.0 * nct * 40.0 and 0.0 < occ <= 1.0  # chain-bound at this size: ~40 us per block column
    for order in (S.slanted(nct, 0.2), S.slanted(nct, 100.0), S.chunked(nct, 8, True), S.chunked(nct, 8, False)):
        seen = {}
        for (i, j, k0, k1, final) in order:
            assert seen.get((i, j), 0) == k0  # chunks of a tile come in k order and leave no gap
            seen[(i, j)] = k1
        assert all(seen[(i, j)] == j for j in range(nct) for i in range(j, nct)) and len(seen) == nct * (nct + 1) // 2
        t, _ = S.simulate(nct, order)
        assert t > 0.97 * base


def _run_bench(argv, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    import os
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_flag_without_enough_devices_is_an_error_line_not_a_one_gpu_run():
    """``python bench.py --gpus N`` started without a launcher must produce an N-rank run or FAIL.  With
    fewer than N devices on the node (here: none, or one on the GPU box) it prints the contract's line with value null and
    ``error``, n_gpus = what was asked, and exits non-zero -- it never degrades to the one-GPU C3 line."""
    import json

    from gumbi_amd import engine

    want = max(engine.device_count(), 1) + 1
    out = _run_bench(["--gpus", str(want), "--steps", "1", "--warmup", "0"])
    assert out.returncode == 2, (out.returncode, out.stderr[-500:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == want and d["value"] is None and d["scaling"] == "strong"
    assert f"needs {want} MI355X devices" in d["error"]["error"] and "this node shows" in d["error"]["error"]
    assert "ONE GP" in d["config"]["workload"]


def test_gpus_flag_must_agree_with_the_launchers_world_size():
    """A launcher that started another number of ranks than ``--gpus`` names is refused before any device is touched."""
    import json

    out = _run_bench(["--gpus", "4", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert out.returncode == 2
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["value"] is None and "WORLD_SIZE=1" in d["error"]["error"] and "--gpus 4" in d["error"]["error"]


def test_multi_gpu_line_validates_itself_against_the_one_gpu_run():
    """match_against_one_gpu: NLML / grid mean / variance of the N-rank step against the one-GPU step; a
    mismatch is reported as one, with the figures."""
    bench = load_bench()
    rng = np.random.default_rng(0)
    one = {"nlml": 56701.96, "mean": rng.standard_normal(100), "var": rng.random(100) + 0.1}
    same = {"nlml": one["nlml"] * (1 + 2e-14), "mean": one["mean"] + 1e-13, "var": one["var"] + 1e-14}
    r = bench.match_against_one_gpu(same, one, 0)
    assert r["results_match_one_gpu"] is True and r["nlml_rel_diff_vs_one_gpu"] < 1e-13 and r["mean_max_rel_diff_vs_one_gpu"] < 1e-12
    off = dict(same, nlml=one["nlml"] * (1 + 1e-8))
    r = bench.match_against_one_gpu(off, one, 0)
    assert r["results_match_one_gpu"] is False and "NLML rel" in r["results_match_note"]
    assert bench.match_against_one_gpu(off, one, 5)["results_match_one_gpu"] is True  # two optimiser trajectories: looser
    r = bench.match_against_one_gpu(dict(same, mean=one["mean"] + 1e-3), one, 0)
    assert r["results_match_one_gpu"] is False
    assert bench.match_against_one_gpu(None, one, 0)["results_match_one_gpu"] is None
    assert bench.match_against_one_gpu(dict(same, nlml=float("nan")), one, 0)["results_match_one_gpu"] is False


def test_c5_fit_predict_block_labels_its_estimate(tmp_path, monkeypatch):
    """The default line's figure for fit+predict at the north star's size: an ESTIMATE from this run's evaluation time and
    the recorded fit's evaluation count, beside the builder-run record itself."""
    import json

    bench = load_bench()
    (tmp_path / "profiles").mkdir()
    (tmp_path / "profiles" / "r98_bench_c5_map.json").write_text(json.dumps(
        {"fit_predict_seconds": 430.0, "phases": {"fit_quality_after_k_evals": {"n_eval": 30, "converged": True, "corr": 0.999}}}))
    monkeypatch.setattr(bench, "ROOT", tmp_path)
    b = bench.c5_fit_predict_block({"phases": {"map_eval_s": 14.0, "predict_s": 1.4}})
    assert b["measured_record"]["fit_predict_seconds"] == 430.0 and b["measured_record"]["n_eval"] == 30
    assert abs(b["c5_fit_predict_seconds_estimate"] - (30 * 14.0 + 1.4)) < 0.06 and "ESTIMATE" in b["estimate_note"]
    (tmp_path / "profiles" / "r98_bench_c5_map.json").unlink()
    assert bench.c5_fit_predict_block({"phases": {"map_eval_s": 14.0}}) == {}


def test_c4_host_baseline_counts_the_gpu_lines_flops():
    """host and GPU on ONE flop basis (the Kronecker form's), the stacked system's executed flops beside it."""
    bench = load_bench()
    cfg = dict(bench.CONFIGS["c4"], N=256, res=8)
    r = bench.cpu_baseline_c4(cfg, target_seconds=0.01)
    n = int(r["sample"].split("at N=")[1].split(" ")[0])
    # (value and seconds are rounded to two decimals in the report: at this toy size that alone is a few per cent)
    assert abs(r["value"] - bench.c4_step_flops(n, 2, 2 * 64, 1, 0) / r["seconds"] / 1e9) <= 0.011 + 0.05 * r["value"]
    assert r["executed_stacked_system_gflops"] > 2.0 * r["value"] and "SAME algorithmic flops" in r["note"]


def test_host_c2_fit_is_capped_and_extrapolated_when_the_budget_is_short(monkeypatch):
    """under the driver's command the host's whole C2 fit (~3 minutes) does not fit the time budget any more;
    a fit capped at C2_HOST_CAPPED_EVALS evaluations is measured instead and the whole fit is reported as a labelled extrapolation."""
    import json
    import subprocess
    import types

    bench = load_bench()
    monkeypatch.setattr(bench, "cpu_baseline", lambda cfg, target_seconds=20.0: {
        "seconds": 9.8, "evaluation_seconds": 5.9, "predict_seconds": 3.9, "sample": "fake"})
    calls = []

    def fake_run(cmd, **kw):
        calls.append(kw["env"]["GUMBI_BENCH_CPU_FIT_MAXEVAL"])
        n = 6 if kw["env"]["GUMBI_BENCH_CPU_FIT_MAXEVAL"] != "200" else 29
        return types.SimpleNamespace(returncode=0, stderr="", stdout=json.dumps(
            {"fit_predict_seconds": 6.0 * n + 4.0, "fit_seconds": 6.0 * n, "predict_seconds": 4.0, "n_eval": n, "nlml_final": 1.0}) + "\n")

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(bench, "cpu_dpotrf_seconds", lambda N: {"c3_cholesky_seconds": 44.0, "c3_cholesky": {"N": N}})

    class B:
        def __init__(self, left):
            self.left = left

        def allows(self, est):
            return est <= self.left

        def skipped(self, est):
            return {"skipped": f"needs ~{est:.0f} s"}

    # plenty of time: the whole fit is measured
    out = bench.cpu_config_size_sections(B(1000.0), {"n_eval": 29})
    assert calls == ["200"] and out["c2_seconds"] == 6.0 * 29 + 4.0 and "c2_seconds_extrapolated" not in out and out["c3_cholesky_seconds"] == 44.0
    # ~100 s left: the capped fit, the extrapolation says what it is; no dpotrf of 75 s... the budget object here allows it (est 75 <= 100)
    calls.clear()
    out = bench.cpu_config_size_sections(B(100.0), {"n_eval": 29})
    assert calls == [str(bench.C2_HOST_CAPPED_EVALS)] and out["c2_seconds"] is None and "skipped" in out["c2_fit"]
    assert abs(out["c2_seconds_extrapolated"] - (6.0 * 29 + 4.0)) < 1e-9 and "EXTRAPOLATION" in out["c2_seconds_extrapolated_note"]
    assert out["c2_fit_capped"]["n_eval"] == 6 and out["c2_seconds_estimate"] == round(29 * 5.9 + 3.9, 1)
    # 30 s left: one evaluation only, everything else skipped and saying so
    calls.clear()
    out = bench.cpu_config_size_sections(B(30.0), {"n_eval": 29})
    assert calls == [] and out["c2_seconds"] is None and "skipped" in out["c2_fit"] and out["c3_cholesky_seconds"] is None
    assert out["c2_one_evaluation_seconds"] == 5.9
    # nothing left
    out = bench.cpu_config_size_sections(B(5.0), None)
    assert out["c2_seconds"] is None and "skipped" in out["c2_fit"] and "c2_one_evaluation_seconds" not in out
