"""Wall time of engine.predict for small grids (the default prepare_grid is 100 points): host latency or GPU time?
   python tools/gpu_predict_small.py"""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O
for N, d in ((400, 1), (2000, 3), (10000, 4)):
    X, y, ls = O.synthetic_table(N, d)
    e = engine.Engine(0); e.set_data(X, y); e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d))))
    e.set_theta(np.concatenate([ls, [1.0, 0.2]])); e.factorize()
    for M in (100, 2500, 10000):
        Xs = np.random.default_rng(1).standard_normal((M, d))
        e.predict(Xs)
        best = 1e9
        for _ in range(10):
            t0 = time.perf_counter(); e.predict(Xs); best = min(best, time.perf_counter() - t0)
        print(f"N={N} M={M}: predict {1e3*best:.3f} ms wall, {e.timings()['predict_ms']:.3f} ms on the GPU ({N*N*M/best/1e12:.2f} TF/s)")
    e.close()
