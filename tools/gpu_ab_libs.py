"""Interleaved A/B of two builds of libgumbi_hip.so on the fused evaluation: ms per gmb_evaluate and result bits, one process per
build and round.   python tools/gpu_ab_libs.py LIB_A LIB_B      (AB_SIZES, AB_ROUNDS)"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time, hashlib
sys.path.insert(0, %r)
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O
for N in [int(v) for v in os.environ.get("AB_SIZES", "2000,5200,10000,20000").split(",")]:
    d = 4
    X, y, ls = O.synthetic_table(N, d)
    theta = np.concatenate([ls, [1.0, 0.2]])
    e = engine.Engine(0); e.set_data(X, y); e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d))))
    val, g = e.evaluate(theta); best = 1e9
    for _ in range(8):
        t0 = time.perf_counter(); e.evaluate(theta); best = min(best, (time.perf_counter() - t0) * 1e3)
    print("lib", os.path.basename(os.environ.get("GUMBI_HIP_LIB", "in-tree")), "N", N, "ms %%.3f" %% best, "TF/s on N^3 %%.1f" %% (float(N)**3 / best / 1e9),
          "bits", hashlib.sha1(g.tobytes() + e.copy_alpha().tobytes()).hexdigest()[:12], flush=True)
    e.close()
''' % root
libs = sys.argv[1:3]
for _ in range(int(os.environ.get("AB_ROUNDS", "2"))):
    for lib in libs:
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GUMBI_HIP_LIB=os.path.abspath(lib)), capture_output=True, text=True)
        print("\n".join(l for l in r.stdout.splitlines() if l.startswith("lib")) or r.stderr[-800:], flush=True)
