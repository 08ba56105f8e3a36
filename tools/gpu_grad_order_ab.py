"""A/B on one box of the tile order of the gradient's triangular-operand products at C3's size (tuning build): GMB_LPT_ORDER=1 (longest contraction
first, the default) against 0 (XCD runs with L2-aware strips), and the launch size up to which the longest-first order is used (GMB_LPT_MAX_TILES; GO_N, GO_MAX_TILES, GO_ROUNDS).  One process per variant and round; prints ms per
gmb_evaluate, its gradient phase and the factorisation.     GUMBI_BUILD_TUNING=1 python -m gumbi_amd.build; python tools/gpu_grad_order_ab.py"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O
N, d = int(os.environ.get("GO_N", "50000")), 8
X, y, ls = O.synthetic_table(N, d)
theta = np.concatenate([ls, [1.0, 0.2]])
e = engine.Engine(0); e.set_data(X, y); e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d)), kind="Matern52"))
val, g = e.evaluate(theta); best = 1e9
for _ in range(2):
    t0 = time.perf_counter(); e.evaluate(theta); best = min(best, (time.perf_counter() - t0) * 1e3)
tm = e.timings()
print("variant", os.environ.get("GO_TAG"), "evaluate ms %%.1f" %% best, "chol %%.1f grad %%.1f" %% (tm["chol_ms"], tm["grad_ms"]), "nlml %%.6f |g| %%.6e" %% (val, float(np.linalg.norm(g))), flush=True)
e.close()
''' % root
lib = os.path.join(root, "gumbi_amd", "lib", "libgumbi_hip_tuning.so")
variants = [("lpt_always", {"GMB_LPT_MAX_TILES": "1000000000"}), ("lpt_never", {"GMB_LPT_ORDER": "0"})] + [
    ("lpt_up_to_%s_tiles" % m, {"GMB_LPT_MAX_TILES": m}) for m in os.environ.get("GO_MAX_TILES", "4096,16384,65536").split(",")]
if os.environ.get("GO_VARIANTS"):  # free-form: "name:KEY=VAL,KEY2=VAL2;name2:..." (any environment switch of the tuning build)
    variants = [("default", {})]
    for item in os.environ["GO_VARIANTS"].split(";"):
        name, _, kv = item.partition(":")
        variants.append((name, dict(p.split("=", 1) for p in kv.split(",") if p)))
for _ in range(int(os.environ.get("GO_ROUNDS", "2"))):
    for tag, env in variants:
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GUMBI_HIP_LIB=lib, GO_TAG=tag, **env), capture_output=True, text=True)
        print("\n".join(l for l in r.stdout.splitlines() if l.startswith("variant")) or r.stderr[-600:], flush=True)
