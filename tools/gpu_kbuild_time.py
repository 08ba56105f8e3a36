import sys, time; sys.path.insert(0, '.')
import numpy as np
import bench
from gumbi_amd import engine
for N, d, kind in ((10000, 4, "ExpQuad"), (30000, 8, "Matern52")):
    X, y, ls = bench.synthetic_table(N, d)
    e = engine.Engine(0); e.set_data(X, y); e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d)), kind=kind)); e.set_theta(np.concatenate([ls, [1.0, 0.2]]))
    e.factorize(); ks = []
    for _ in range(5):
        e.factorize(); ks.append(e.timings()["kbuild_ms"])
    e.factorize(); t0 = time.perf_counter(); v, g = e.nlml(grad=True); tg = time.perf_counter() - t0
    print(f"N={N} {kind}: kbuild {min(ks):.3f} ms ({8*N*(N+1)/2/min(ks)/1e6:.0f} GB/s)  nlml {v:.9f}")
