import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O
N, d = int(sys.argv[1]), 2
X, y, ls = O.synthetic_table(N, d)
e = engine.Engine(0)
e.set_data(X, y)
e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d))))
e.set_theta(np.concatenate([ls, [1.0, 0.2]]))
print("scheme 0", flush=True)
e.set_chol_scheme(0); e.factorize(); L0 = e.copy_factor(); print("ok", e.nlml(), flush=True)
print("scheme 3", flush=True)
e.set_chol_scheme(3)
e.chol_task_trace(1)
t0 = time.time()
try:
    e.factorize()
    print("ok", e.nlml(), time.time() - t0, flush=True)
    L3 = e.copy_factor()
    print("dL", np.max(np.abs(np.tril(L3) - np.tril(L0))), flush=True)
except Exception as ex:
    print("failed after", time.time() - t0, ex, flush=True)
tr = e.chol_task_trace(0)
if tr is not None:
    tiles, st = tr
    for (i, j), s in zip(tiles[:40], st[:40]):
        print(i, j, np.round(s * 1e6, 1), flush=True)
