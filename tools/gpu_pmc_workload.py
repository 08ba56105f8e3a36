"""Small fixed workload for rocprofv3 PMC passes: K-build + Cholesky + gradient + predict at N."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 4
kind = sys.argv[3] if len(sys.argv) > 3 else "ExpQuad"
X, y, ls = O.synthetic_table(N, d); Xs = O.synthetic_grid(d)
eng = engine.Engine(0); eng.set_data(X, y); eng.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d)), kind=kind))
eng.set_theta(np.concatenate([ls, [1.0, 0.2]]))
for _ in range(2):
    eng.factorize(); eng.predict(Xs); eng.factorize(); eng.nlml(grad=True)
print("done", N, d, kind)
