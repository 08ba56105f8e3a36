"""A few fused evaluations at one size for a counter pass (tools/gpu_pmc_script.sh): EP_N (10000), EP_D (4), EP_KIND (ExpQuad),
EP_REPS (4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O

N, d = int(os.environ.get('EP_N', '10000')), int(os.environ.get('EP_D', '4'))
X, y, ls = O.synthetic_table(N, d)
theta = np.concatenate([ls, [1.0, 0.2]])
e = engine.Engine(0)
e.set_data(X, y)
e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d)), kind=os.environ.get('EP_KIND', 'ExpQuad')))
for _ in range(int(os.environ.get('EP_REPS', '4'))):
    val, g = e.evaluate(theta)
print(N, val)
e.close()
