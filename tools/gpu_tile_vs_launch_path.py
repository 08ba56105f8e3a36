"""Where should the persistent evaluation launch hand over to the launch schedules?  ms per gmb_evaluate on one box for the fused tile launch (the default up
to 224 block columns) against the large matrices' path (recursion with tile-kernel panels + launch-tree gradient), around the hand-over.
    python tools/gpu_tile_vs_launch_path.py        (TL_SIZES)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O

d = 4
for N in [int(v) for v in os.environ.get("TL_SIZES", "16384,20000,24576,28672").split(",")]:
    X, y, ls = O.synthetic_table(N, d)
    theta = np.concatenate([ls, [1.0, 0.2]])
    out = []
    for name, chol, grad in (("fused tile launch", engine.Engine.CHOL_BY_SIZE, engine.Engine.GRAD_BY_SIZE), ("launch path", 4, engine.Engine.GRAD_LAUNCH_TREE)):
        e = engine.Engine(0); e.set_data(X, y); e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d))))
        e.set_chol_scheme(chol); e.set_grad_scheme(grad)
        val, g = e.evaluate(theta); best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); e.evaluate(theta); best = min(best, (time.perf_counter() - t0) * 1e3)
        out.append((name, best, val, float(np.linalg.norm(g))))
        e.close()
    print("N", N, " | ".join("%s %.2f ms (%.1f TF/s on N^3)" % (n, t, float(N) ** 3 / t / 1e9) for n, t, _, _ in out),
          "| nlml rel diff %.1e, |g| rel diff %.1e" % (abs(out[0][2] - out[1][2]) / abs(out[0][2]), abs(out[0][3] - out[1][3]) / out[0][3]), flush=True)
