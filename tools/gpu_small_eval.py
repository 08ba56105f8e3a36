"""Small matrices: wall time per gmb_evaluate on the stream schedules (default below 6 block columns) against the persistent
evaluation launch forced for every size.  SE_SIZES."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O

sizes = [int(v) for v in os.environ.get('SE_SIZES', '100,200,392,500,640,768,1000,1500,2048').split(',')]
d = int(os.environ.get('SE_D', '1'))
for N in sizes:
    X, y, ls = O.synthetic_table(N, d)
    spec = O.make_spec(d, range(d), kind='ExpQuad')
    theta = O.pack_theta(spec, ls, 1.0, 0.2)
    e = engine.Engine(0)
    e.set_data(X, y)
    e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d))))
    out = []
    ref = None
    for name, cs, gs in (("default", -1, -1), ("streams", 0, 0), ("tiles+tree", 3, 0), ("fused", 3, 2)):
        e.set_chol_scheme(cs)
        e.set_grad_scheme(gs)
        val, g = e.evaluate(theta)
        if ref is None:
            ref = g
        best = 1e9
        for _ in range(20):
            t0 = time.perf_counter()
            e.evaluate(theta)
            best = min(best, (time.perf_counter() - t0) * 1e3)
        out.append(f"{name}: {best:.3f} ms (dg {np.max(np.abs(g - ref)) / max(1.0, np.max(np.abs(ref))):.0e})")
    print(f"N={N}: " + " | ".join(out), flush=True)
    e.close()
