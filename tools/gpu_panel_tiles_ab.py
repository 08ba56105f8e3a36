"""A/B of the large matrices' Cholesky schedule on ONE box: plain recursion with launches only (scheme 0) against the recursion whose
bottom panels are single launches of the tile kernel (scheme 4; csrc/engine.hip: chol_tiles_panel), panel widths GMB_PANEL_TILES_W.
    GUMBI_BUILD_TUNING=1 python -m gumbi_amd.build; python tools/gpu_panel_tiles_ab.py      (PT_SIZES, PT_WIDTHS, PT_ROUNDS)
One process per (scheme, width) and round, interleaved; prints ms per gmb_factorize (best of 3) and the NLML."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
from gumbi_amd import engine
from oracle import gp_oracle as O
N, scheme = int(os.environ["PT_N"]), int(os.environ["PT_SCHEME"])
d = 8
X, y, ls = O.synthetic_table(N, d)
theta = np.concatenate([ls, [1.0, 0.2]])
e = engine.Engine(0); e.set_data(X, y); e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d)), kind="Matern52")); e.set_theta(theta)
e.set_chol_scheme(scheme)
e.factorize(); best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); e.factorize(); best = min(best, (time.perf_counter() - t0) * 1e3)
e.set_profiling(True); e.factorize(); tm = e.timings()
print("N", N, "scheme", scheme, "w", os.environ.get("GMB_PANEL_TILES_W", "-"), "factorize ms %%.2f" %% best, "(%%.2f TF/s)" %% (float(N)**3 / 3 / best / 1e9),
      "chol_ms %%.2f leaf %%.2f trsm %%.2f panel-tile launches %%d %%.2f ms; in-panel %%.2f ms" %% (tm["chol_ms"], tm["chol_leaf_ms"], tm["chol_trsm_ms"],
      tm["total_chol_panel_tile_launches"], tm["total_chol_panel_tile_ms"], tm["total_chol_panel_gemm_ms"]), "nlml %%.6f" %% e.nlml(), flush=True)
e.close()
''' % root
lib = os.path.join(root, "gumbi_amd", "lib", "libgumbi_hip_tuning.so")
variants = [(0, None)] + [(4, w) for w in os.environ.get("PT_WIDTHS", "4,8,16").split(",")]
for N in os.environ.get("PT_SIZES", "50000").split(","):
    for _ in range(int(os.environ.get("PT_ROUNDS", "2"))):
        for scheme, w in variants:
            env = dict(os.environ, GUMBI_HIP_LIB=lib, PT_N=N, PT_SCHEME=str(scheme))
            if w is not None:
                env["GMB_PANEL_TILES_W"] = str(w)
            r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
            print("\n".join(l for l in r.stdout.splitlines() if l.startswith("N ")) or r.stderr[-800:], flush=True)
