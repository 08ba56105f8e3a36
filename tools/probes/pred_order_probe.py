"""Probe (tuning build): tile order of the GEMM-form prediction's triangular-operand product -- longest-first up to GMB_LPT_MAX_TILES tiles, L2-aware strips beyond."""
import os, sys, time, subprocess
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import os, sys, time
sys.path.insert(0, os.environ["GUMBI_ROOT"])
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O
for N in (10000, 20000):
    d = 4
    X, y, ls = O.synthetic_table(N, d)
    theta = np.concatenate([ls, [1.0, 0.2]])
    e = engine.Engine(0); e.set_data(X, y); e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d))))
    e.evaluate(theta)
    Xs = O.synthetic_grid(d, 100)
    e.predict(Xs); best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); mu, var = e.predict(Xs); best = min(best, (time.perf_counter() - t0) * 1e3)
    print("variant", os.environ.get("TAG"), "N", N, "predict ms %.3f" % best, "gemm form", e.timings()["predict_gemm_form"], flush=True)
    e.close()
'''
lib = root + "/gumbi_amd/lib/libgumbi_hip_tuning.so"
for _ in range(2):
    for tag, env in (("lpt<=16384", {}), ("lpt<=4096", {"GMB_LPT_MAX_TILES": "4096"}), ("never", {"GMB_LPT_ORDER": "0"})):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GUMBI_HIP_LIB=lib, GUMBI_ROOT=root, TAG=tag, **env), capture_output=True, text=True)
        print("\n".join(l for l in r.stdout.splitlines() if l.startswith("variant")) or r.stderr[-500:], flush=True)
