"""Fused evaluation launch with Sigma^-1 as 128 x 256 pair tasks (gmb_set_eval_pairs(1)) against the 128 x 128 form: one process per
setting; ms per gmb_evaluate, NLML and gradient bits.  EP_SIZES."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time, hashlib
sys.path.insert(0, %r)
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O
for N in [int(v) for v in os.environ.get("EP_SIZES", "3000,10000,20000").split(",")]:
    d = 4
    X, y, ls = O.synthetic_table(N, d)
    theta = np.concatenate([ls, [1.0, 0.2]])
    e = engine.Engine(0); e.set_data(X, y); e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d))))
    e.set_eval_pairs(int(os.environ.get("GMB_ET_PAIRS", "0")))
    val, g = e.evaluate(theta); best = 1e9
    for _ in range(8):
        t0 = time.perf_counter(); e.evaluate(theta); best = min(best, (time.perf_counter() - t0) * 1e3)
    a = e.copy_alpha()
    print("pairs", os.environ.get("GMB_ET_PAIRS", "0"), "N", N, "ms %%.3f" %% best, "TF/s on N^3 %%.1f" %% (float(N)**3 / best / 1e9), "nlml %%.12g" %% val,
          "bits", hashlib.sha1(g.tobytes() + a.tobytes()).hexdigest()[:12], flush=True)
    e.close()
''' % root
for pairs in ("0", "1", "0", "1"):
    env = dict(os.environ, GMB_ET_PAIRS=pairs)
    if os.environ.get("EP_TUNING") == "1":
        env["GUMBI_HIP_LIB"] = os.path.join(root, "gumbi_amd", "lib", "libgumbi_hip_tuning.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("\n".join(l for l in r.stdout.splitlines() if l.startswith("pairs")) or r.stderr[-800:], flush=True)
