"""Interleaved A/B of two builds of libgumbi_hip.so: ms per gmb_evaluate, gmb_factorize, gmb_predict (10^4 points) and result bits, one process per
build and round.   python tools/gpu_ab_libs.py LIB_A LIB_B [LIB_C ...]      (AB_SIZES, AB_ROUNDS)
Builds for an A/B of compile-time switches: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DCT_STRIP_LDS=0 ... gumbi_amd/csrc/engine.hip
-o gumbi_amd/lib/ab_x.so (tools/build_ab_libs.sh)."""
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
    alpha = e.copy_alpha()
    Xs = O.synthetic_grid(d, 100)
    bf = bp = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); e.factorize(); bf = min(bf, (time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter(); mu, var = e.predict(Xs); bp = min(bp, (time.perf_counter() - t0) * 1e3)
    print("lib", os.path.basename(os.environ.get("GUMBI_HIP_LIB", "in-tree")), "N", N, "evaluate ms %%.3f" %% best, "(%%.1f TF/s on N^3)" %% (float(N)**3 / best / 1e9),
          "factorize ms %%.3f" %% bf, "predict(10^4) ms %%.3f" %% bp,
          "bits", hashlib.sha1(g.tobytes() + alpha.tobytes() + e.copy_v().tobytes() + mu.tobytes() + var.tobytes()).hexdigest()[:12], flush=True)
    e.close()
''' % root
libs = sys.argv[1:]  # two or more builds
for _ in range(int(os.environ.get("AB_ROUNDS", "2"))):
    for lib in libs:
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GUMBI_HIP_LIB=os.path.abspath(lib)), capture_output=True, text=True)
        print("\n".join(l for l in r.stdout.splitlines() if l.startswith("lib")) or r.stderr[-800:], flush=True)
