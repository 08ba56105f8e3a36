"""Probe (tuning build): fused evaluation time against the leading dimension of the factor buffer (GMB_LD_PAD doubles beyond Nr).
One process per setting.  LP_N."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O
N, d = int(os.environ.get("LP_N", "10000")), 4
X, y, ls = O.synthetic_table(N, d)
theta = np.concatenate([ls, [1.0, 0.2]])
e = engine.Engine(0); e.set_data(X, y); e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d))))
val, g = e.evaluate(theta); best = 1e9
for _ in range(10):
    t0 = time.perf_counter(); e.evaluate(theta); best = min(best, (time.perf_counter() - t0) * 1e3)
print("pad", os.environ.get("GMB_LD_PAD", "0"), "N", N, "ms %%.3f" %% best, "nlml %%.9f" %% val, "g0 %%.9e" %% g[0])
''' % root
for pad in os.environ.get("LP_PADS", "0,16,32,48,64,96,144,272").split(","):
    env = dict(os.environ, GMB_LD_PAD=pad, GUMBI_HIP_LIB=os.path.join(root, "gumbi_amd", "lib", "libgumbi_hip_tuning.so"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
