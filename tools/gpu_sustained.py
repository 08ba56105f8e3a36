"""Sustained rate of the N x N factorisation under environment settings (argv: VAR=val[,VAR2=val2] ...): REPS
back-to-back factorisations, mean of the last two thirds -- the chip settles to its power-limited clock after
~10 s (a 3-step bench reads 70 TF/s where 25 steps read 67).   SUSTAIN_N=50000 SUSTAIN_REPS=60"""
import os, subprocess, sys
CODE = r'''
import os, sys, time; sys.path.insert(0, '.')
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O
N = int(os.environ.get("SUSTAIN_N", "50000")); d = 8; reps = int(os.environ.get("SUSTAIN_REPS", "60"))
X, y, ls = O.synthetic_table(N, d)
e = engine.Engine(0); e.set_data(X, y); e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d)), kind="Matern52")); e.set_theta(np.concatenate([ls, [1.0, 0.2]]))
e.factorize(); ts = []
for _ in range(reps):
    t0 = time.perf_counter(); e.factorize(); ts.append(time.perf_counter() - t0)
tail = ts[reps // 3:]
print("N=%d: first 3 %.1f ms, mean of the last %d %.1f ms (%.1f TF/s), min %.1f max %.1f" % (N, 1e3 * np.mean(ts[:3]), len(tail), 1e3 * np.mean(tail), N**3 / 3 / np.mean(tail) / 1e12, 1e3 * min(ts), 1e3 * max(ts)))
'''
for setting in sys.argv[1:] or ["X=0"]:
    env = dict(os.environ)
    for kv in setting.split(","):
        if "=" in kv:
            k, v = kv.split("=", 1); env[k] = v
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(setting); print("  " + (out.stdout.strip().splitlines()[-1] if out.returncode == 0 and out.stdout.strip() else out.stderr[-400:]))
