"""A/B environment-variable settings (argv: VAR=val[,VAR2=val2] ...) on factorize / predict / gradient;
AB_SIZES=10000x4,30000x8 picks the (N, d) cases."""
import os, subprocess, sys
CODE = r'''
import sys, time; sys.path.insert(0, '.')
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O
import os
for N, d in [tuple(int(v) for v in t.split('x')) for t in os.environ.get('AB_SIZES', '10000x4,30000x8').split(',')]:
    X, y, ls = O.synthetic_table(N, d); Xs = O.synthetic_grid(d)
    e = engine.Engine(0); e.set_data(X, y); e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d)))); e.set_theta(np.concatenate([ls, [1.0, 0.2]]))
    e.factorize(); e.predict(Xs); e.factorize(); e.nlml(grad=True)
    best = [1e9, 1e9, 1e9]
    for _ in range(3):
        t0 = time.perf_counter(); e.factorize(); t1 = time.perf_counter(); e.predict(Xs); t2 = time.perf_counter()
        e.factorize(); t3 = time.perf_counter(); e.nlml(grad=True); t4 = time.perf_counter()
        best = [min(best[0], t1 - t0), min(best[1], t2 - t1), min(best[2], t4 - t3)]
    e.factorize(); nl = e.nlml()
    print('N=%d: factorize %.2f ms (%.1f TF/s)  predict %.2f ms (%.1f TF/s)  grad %.2f ms (%.1f TF/s)  nlml %.12g' % (N, best[0]*1e3, N**3/3/best[0]/1e12, best[1]*1e3, N*N*len(Xs)/best[1]/1e12, best[2]*1e3, 2*N**3/3/best[2]/1e12, nl))
    e.close()
'''
for setting in sys.argv[1:]:
    env = dict(os.environ)
    for kv in setting.split(","):
        if "=" in kv:
            k, v = kv.split("=", 1)
            env[k] = v
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(setting)
    print("  " + "\n  ".join(l for l in out.stdout.strip().splitlines() if l.startswith("N=")) if out.returncode == 0 else out.stderr[-400:])
