"""Two MAP evaluations at (GR_N, GR_D, GR_KIND) for a rocprofv3 --kernel-trace --stats run: the trace reductions
(grad_tile_kernel) and the covariance build (cov_tile_kernel) per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O

N, d, kind = int(os.environ.get('GR_N', '50000')), int(os.environ.get('GR_D', '8')), os.environ.get('GR_KIND', 'Matern52')
X, y, ls = O.synthetic_table(N, d)
spec = O.make_spec(d, range(d), kind=kind)
theta = O.pack_theta(spec, ls * 3.0, 1.0, 0.3)
e = engine.Engine(0)
e.set_data(X, y)
e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d)), kind=kind))
for _ in range(int(os.environ.get('GR_REPS', '2'))):
    val, g = e.evaluate(theta)
print(N, d, kind, val, g)
e.close()
