"""A/B builds of libgumbi_hip.so (paths as argv) on the MFMA GEMM itself (plain products through
gmb_blk_gemm_nt) and on factorize / gradient at N = 30000: one line per library."""
import os, subprocess, sys
CODE = r'''
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from gumbi_amd import engine
from oracle import gp_oracle as O
eng = engine.Engine(0); dev = torch.device("cuda:0"); out = []
for m, n, k in ((8192, 8192, 8192), (16384, 16384, 3072), (4096, 4096, 1024)):
    A = torch.randn(k, m, dtype=torch.float64, device=dev); B = torch.randn(k, n, dtype=torch.float64, device=dev)
    Cm = torch.zeros(m, n, dtype=torch.float64, device=dev); torch.cuda.synchronize(); best = 1e9
    for rep in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.blk_gemm_nt(Cm.data_ptr(), n, A.data_ptr(), m, B.data_ptr(), n, m, n, k, -1.0, 1.0)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    out.append("%dx%dx%d %.1f" % (m, n, k, 2.0 * m * n * k / best / 1e12))
    del A, B, Cm
eng.close()
N, d = 30000, 8
X, y, ls = O.synthetic_table(N, d)
e = engine.Engine(0); e.set_data(X, y); e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d)))); e.set_theta(np.concatenate([ls, [1.0, 0.2]]))
e.factorize(); e.nlml(grad=True); b = [1e9, 1e9]
for _ in range(3):
    t0 = time.perf_counter(); e.factorize(); t1 = time.perf_counter(); e.nlml(grad=True); t2 = time.perf_counter()
    b = [min(b[0], t1 - t0), min(b[1], t2 - t1)]
v = e.nlml()
print("GEMM TF/s: " + "  ".join(out) + "  | N=30k factorize %.1f ms (%.1f TF/s) grad %.1f ms (%.1f TF/s) nlml %.10g" % (b[0]*1e3, N**3/3/b[0]/1e12, b[1]*1e3, 2*N**3/3/b[1]/1e12, v))
'''
for lib in sys.argv[1:]:
    out = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, GUMBI_HIP_LIB=os.path.abspath(lib)), capture_output=True, text=True)
    last = [l for l in out.stdout.strip().splitlines() if l.startswith("GEMM")]
    print("%-44s %s" % (os.path.basename(lib), last[-1] if out.returncode == 0 and last else out.stderr[-400:]), flush=True)
