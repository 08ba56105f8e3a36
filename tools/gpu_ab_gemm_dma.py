"""A/B of the 128 x 128 GEMM's staging: register-staged double buffer (shipped) against the LDS-DMA ring (GMB_GEMM_DMA=1),
interleaved runs in separate processes on one box: plain products through gmb_blk_gemm_nt (with a correctness check of the
DMA variant against the shipped one) and the N = 50k factorisation whose bulk trailing updates are the headline kernel."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, time, os; sys.path.insert(0, %r)
import numpy as np, torch
from gumbi_amd import engine
from oracle import gp_oracle as O
eng = engine.Engine(0); dev = torch.device("cuda:0"); out = []
torch.manual_seed(1)
for m, n, k in ((8192, 8192, 8192), (16384, 16384, 3072), (24576, 24576, 1024), (4096, 4096, 1024)):
    A = torch.randn(k, m, dtype=torch.float64, device=dev); B = torch.randn(k, n, dtype=torch.float64, device=dev)
    Cm = torch.zeros(m, n, dtype=torch.float64, device=dev); torch.cuda.synchronize(); best = 1e9
    for rep in range(6):
        Cm.zero_()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.blk_gemm_nt(Cm.data_ptr(), n, A.data_ptr(), m, B.data_ptr(), n, m, n, k, -1.0, 1.0)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    chk = float(Cm[:64].double().abs().sum())  # (compared between the variants by the caller)
    ref = -(A[:, :64].T @ B)
    err = float((Cm[:64] - ref).abs().max() / ref.abs().max())
    out.append("%%dx%%dx%%d %%.1f (err %%.0e)" %% (m, n, k, 2.0 * m * n * k / best / 1e12, err))
    del A, B, Cm
eng.close()
N, d = int(os.environ.get("AB_N", "50000")), 8
X, y, ls = O.synthetic_table(N, d)
e = engine.Engine(0); e.set_data(X, y); e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d)), kind="Matern52")); e.set_theta(np.concatenate([3 * ls, [1.0, 0.3]]))
e.factorize(); b = 1e9
for _ in range(3):
    t0 = time.perf_counter(); e.factorize(); b = min(b, time.perf_counter() - t0)
v = e.nlml()
print("GEMM TF/s: " + "  ".join(out) + "  | N=%%d factorize %%.1f ms (%%.1f TF/s) nlml %%.12g" %% (N, b*1e3, N**3/3/b/1e12, v))
''' % ROOT
for rep in range(int(os.environ.get("AB_REPS", "2"))):
    for dma in ("0", "1"):
        out = subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, GMB_GEMM_DMA=dma), capture_output=True, text=True)
        last = [l for l in out.stdout.strip().splitlines() if l.startswith("GEMM")]
        print("GMB_GEMM_DMA=%s  %s" % (dma, last[-1] if out.returncode == 0 and last else out.stderr[-600:]), flush=True)
