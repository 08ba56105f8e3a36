"""Per-evaluation wall time of a front-end fit against the engine's own phase times (where the host's share goes):
   python tools/gpu_fit_overhead.py 2000 4   (N, d)"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, pandas as pd
import gumbi_amd as gmb
from oracle import gp_oracle as O
N, d = int(sys.argv[1]), int(sys.argv[2])
X, y, ls = O.synthetic_table(N, d, seed=2)
cols = [f"x{i}" for i in range(d)]
df = pd.DataFrame(X, columns=cols); df["y"] = y
ds = gmb.DataSet(df, outputs=["y"])
for rep in range(3):
    gp = gmb.GP(ds, outputs=["y"])
    t0 = time.perf_counter(); gp.fit(continuous_dims=cols, continuous_kernel="ExpQuad"); t1 = time.perf_counter()
    eng = gp.engine
    th = gp._theta_fitted
    best = [1e9, 1e9]
    for _ in range(5):
        a = time.perf_counter(); eng.set_theta(th); eng.factorize(); b = time.perf_counter(); eng.nlml(grad=True); c = time.perf_counter()
        best = [min(best[0], b - a), min(best[1], c - b)]
    tm = eng.timings()
    print(f"N={N} d={d}: fit {1e3*(t1-t0):.1f} ms, {gp.n_eval} evaluations = {1e3*(t1-t0)/gp.n_eval:.2f} ms each | engine calls alone: set_theta+factorize {1e3*best[0]:.2f} ms (kbuild {tm['kbuild_ms']:.2f} + chol {tm['chol_ms']:.2f} on the GPU), nlml+grad {1e3*best[1]:.2f} ms (GPU {tm['grad_ms']:.2f})")
    gp.engine.close()
