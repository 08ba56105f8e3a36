"""A/B environment settings on the gradient (inverse + Sigma^-1) of one big problem (argv: N setting ...)."""
import os, subprocess, sys
CODE = r'''
import sys, time; sys.path.insert(0, '.')
import numpy as np
import bench
from gumbi_amd import engine
N = int(sys.argv[1]); d = 8
X, y, ls = bench.synthetic_table(N, d)
e = engine.Engine(0); e.set_data(X, y); e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d)))); e.set_theta(np.concatenate([ls, [1.0, 0.2]]))
e.factorize(); e.nlml(grad=True)
best = 1e9
for _ in range(2):
    e.factorize(); t0 = time.perf_counter(); e.nlml(grad=True); best = min(best, time.perf_counter() - t0)
e.set_profiling(True); e.factorize(); e.nlml(grad=True); tm = e.timings()
print('N=%d: grad %.1f ms (%.1f TF/s); grad GEMM launches: %.1f ms at %.1f TF/s' % (N, best*1e3, 2*N**3/3/best/1e12, tm['grad_gemm_ms'], tm['grad_gemm_flops']/tm['grad_gemm_ms']/1e9))
'''
N = sys.argv[1]
for setting in sys.argv[2:]:
    env = dict(os.environ)
    for kv in setting.split(","):
        if "=" in kv:
            k, v = kv.split("=", 1); env[k] = v
    out = subprocess.run([sys.executable, "-c", CODE, N], env=env, capture_output=True, text=True)
    print(setting, "|", out.stdout.strip().splitlines()[-1] if out.returncode == 0 else out.stderr[-300:])
