"""Per-launch rates of the Cholesky's bulk trailing updates (event kind 7) at N x d, under the default
look-ahead schedule and under the plain recursion (GMB_LOOKAHEAD=0): how much of the plain-GEMM rate do the
triangular, chain-accompanied launches keep?     python tools/gpu_bulk_trace.py [N] [d]"""
import os, subprocess, sys, tempfile
CODE = r'''
import sys; sys.path.insert(0, '.')
import numpy as np, time
from gumbi_amd import engine
from oracle import gp_oracle as O
N, d = int(sys.argv[1]), int(sys.argv[2])
X, y, ls = O.synthetic_table(N, d)
e = engine.Engine(0); e.set_data(X, y); e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d)))); e.set_theta(np.concatenate([ls, [1.0, 0.2]]))
e.factorize(); t0 = time.perf_counter(); e.factorize(); dt = time.perf_counter() - t0
e.set_profiling(True); e.factorize(); tm = e.timings()
print("WALL %.2f ms (%.1f TF/s); bulk updates: %d launches, %.1f ms summed, %.1f TF/s; over their wall time %.1f TF/s; in-panel products %.1f ms summed" % (
    dt * 1e3, N**3 / 3 / dt / 1e12, tm["total_chol_gemm_launches"], tm["total_chol_gemm_ms"], tm["total_chol_gemm_flops"] / tm["total_chol_gemm_ms"] / 1e9,
    tm["total_chol_gemm_flops"] / tm["total_chol_gemm_wall_ms"] / 1e9, tm["total_chol_panel_gemm_ms"]))
'''
N = sys.argv[1] if len(sys.argv) > 1 else "50000"
d = sys.argv[2] if len(sys.argv) > 2 else "8"
for label, env in (("default look-ahead", {}), ("plain recursion", {"GMB_LOOKAHEAD": "0"}), ("masked look-ahead", {"GMB_CHOL_SCHEME": "2"})):
    with tempfile.NamedTemporaryFile(suffix=".trace") as tf:
        out = subprocess.run([sys.executable, "-c", CODE, N, d], env=dict(os.environ, GMB_TRACE_FILE=tf.name, **env), capture_output=True, text=True)
        print(label, "|", [l for l in out.stdout.splitlines() if l.startswith("WALL")] or out.stderr[-300:])
        rows = [ln.split() for ln in open(tf.name) if ln.strip()]
        bulk = [(int(r[1]), int(r[2]), int(r[3]), float(r[5]), float(r[6])) for r in rows if r[0] == "7"]
        bulk = bulk[-len(bulk) // 3:] if len(bulk) > 60 else bulk   # the last (profiled) factorisation only
        for mt, nt, k, ms, gf in sorted(bulk, key=lambda b: -b[3])[:12]:
            print("   mt %4d nt %4d k %5d  %8.3f ms  %6.1f TF/s" % (mt, nt, k, ms, gf / ms))
