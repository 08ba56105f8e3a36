"""How many L-BFGS-B evaluations does a CONVERGING MAP fit of the bench tables need, and what does it find?
PyMC-default prior / start (white-noise plateau) against `ls_bounds` lower limits
(the reference's own knob, pymc/GP.py:630-650) -- wall time, n_eval, MAP, correlation with the generator's f.

    python tools/gpu_fit_probe.py c3 0.5 [more lower bounds ...]      (lower = 0 -> PyMC defaults)
"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
import gumbi_amd as gmb  # noqa: E402


def truth(cfg):
    N, d = cfg["N"], cfg["d"]
    rng = np.random.default_rng(2021)
    X = rng.standard_normal((N, d))
    ls = np.geomspace(0.7, 2.0, d) if d > 1 else np.array([1.0])
    f = np.sum(np.sin(X / ls), axis=1) / np.sqrt(d)
    y = f + 0.2 * rng.standard_normal(N)
    Xs = bench.synthetic_grid(d, cfg["res"])
    fs = (np.sum(np.sin(Xs / ls), axis=1) / np.sqrt(d) - y.mean()) / y.std(ddof=1)
    return Xs, fs, 0.2 / y.std(ddof=1)


def main():
    cfg = bench.CONFIGS[sys.argv[1]]
    lowers = [float(v) for v in sys.argv[2:]] or [0.0, 0.5]
    Xs, fs, sig = truth(cfg)
    ds, cols = bench.make_dataset(cfg)
    for lo in lowers:
        gp = gmb.GP(ds, outputs=["y"])
        gp.specify_model(continuous_dims=cols)
        lsb = gmb.make_deltas_parray(stdzr=ds.stdzr, scale="standardized", **{c: [lo, None] for c in cols}) if lo > 0 else None
        gp.build_model(continuous_kernel=cfg["kernel"], ls_bounds=lsb)
        t0 = time.perf_counter()
        gp.find_MAP()
        t1 = time.perf_counter()
        mu, var = gp.predict(Xs)
        t2 = time.perf_counter()
        print(f"{sys.argv[1]} lower={lo}: start ls={np.round(gp._initial_theta()[:cfg['d']], 3)} n_eval={gp.n_eval} rejected={gp._rejected} "
              f"fit={t1 - t0:.2f}s predict={t2 - t1:.2f}s success={gp.opt_result.success} msg={gp.opt_result.message}\n"
              f"   ls={np.round(gp.MAP['ls_total'], 2)} eta={float(gp.MAP['η_total']):.3f} sigma={float(gp.MAP['σ']):.4f} (truth {sig:.4f}) "
              f"nlml={gp.nlml_trace[-1]:.2f} corr={np.corrcoef(mu, fs)[0, 1]:.4f} rmse={np.sqrt(np.mean((mu - fs) ** 2)):.4f} "
              f"trace={np.round(gp.nlml_trace, 1).tolist()}", flush=True)
        gp.engine.close()


if __name__ == "__main__":
    main()
