"""Fused evaluation launch: wall time per gmb_evaluate against the lag of the inverse behind the factorisation in the ticket order.
ET_SIZES, ET_LAGS."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O

sizes = [int(v) for v in os.environ.get('ET_SIZES', '1536,2560,4096,5200,8192,10000,16384,20480').split(',')]
lags = [int(v) for v in os.environ.get('ET_LAGS', '1,2,4,8,12,16,24,32').split(',')]
d = 4
for N in sizes:
    X, y, ls = O.synthetic_table(N, d)
    spec = O.make_spec(d, range(d), kind='ExpQuad')
    theta = O.pack_theta(spec, ls, 1.0, 0.2)
    e = engine.Engine(0)
    e.set_data(X, y)
    e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d))))
    out = []
    for scheme, lag in [(0, -1)] + [(2, l) for l in lags]:
        e.set_grad_scheme(scheme, lag)
        e.evaluate(theta)
        best = 1e9
        for _ in range(6):
            t0 = time.perf_counter()
            e.evaluate(theta)
            best = min(best, (time.perf_counter() - t0) * 1e3)
        out.append(f"{'tree' if scheme == 0 else 'lag ' + str(lag)}: {best:.3f}")
    print(f"N={N} ({(N + 127) // 128} block columns), ms per evaluation: " + " | ".join(out), flush=True)
    e.close()
