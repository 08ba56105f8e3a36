"""Intrinsic rate of each GEMM tile shape on square-ish problems through gmb_blk_gemm_nt
(argv: sizes m n k triples separated by commas, e.g. 8192,8192,8192 4096,4096,1024)."""
import os, subprocess, sys
CODE = r'''
import sys, time; sys.path.insert(0, '.')
import torch
from gumbi_amd.engine import Engine
eng = Engine(0)
dev = torch.device("cuda:0")
for spec in sys.argv[1:]:
    m, n, k = (int(v) for v in spec.split(","))
    A = torch.randn(k, m, dtype=torch.float64, device=dev)   # column-major m x k  (k-major)
    B = torch.randn(k, n, dtype=torch.float64, device=dev)
    Cm = torch.zeros(m, n, dtype=torch.float64, device=dev)  # column-major n x m
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.blk_gemm_nt(Cm.data_ptr(), n, A.data_ptr(), m, B.data_ptr(), n, m, n, k, -1.0, 1.0)
        eng.synchronize() if hasattr(eng, "synchronize") else None
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("  m=%d n=%d k=%d: %.3f ms  %.1f TF/s  tiles128=%d" % (m, n, k, best*1e3, 2.0*m*n*k/best/1e12, (m//128)*(n//128)))
'''
for v in os.environ.get("SWEEP_VARIANTS", "0,1,2,3").split(","):
    out = subprocess.run([sys.executable, "-c", CODE] + sys.argv[1:], env=dict(os.environ, GMB_GEMM_VARIANT=v), capture_output=True, text=True)
    print("variant", v)
    print(out.stdout.rstrip() if out.returncode == 0 else out.stderr[-600:])
