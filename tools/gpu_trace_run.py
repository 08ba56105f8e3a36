import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
d = 4
X, y, ls = O.synthetic_table(N, d); Xs = O.synthetic_grid(d)
eng = engine.Engine(0); eng.set_data(X, y); eng.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d))))
eng.set_theta(np.concatenate([ls, [1.0, 0.2]]))
eng.factorize(); eng.predict(Xs); eng.factorize(); eng.nlml(grad=True)   # warm
eng.set_profiling(True)
eng.factorize(); eng.predict(Xs); eng.factorize(); v, g = eng.nlml(grad=True)
print("timings", {k: round(v, 3) for k, v in eng.timings().items() if k.endswith("_ms")})
