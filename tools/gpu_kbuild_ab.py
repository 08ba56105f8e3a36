"""A/B builds of libgumbi_hip.so (paths as argv; none = the in-tree library) on the covariance build ALONE
(gmb_blk_covariance into a scratch buffer): min wall time of 5 launches and algorithmic GB/s, 8 N(N+1)/2 bytes.
    python tools/gpu_kbuild_ab.py [LIB ...]      AB_CASES=50000:8:Matern52,10000:4:ExpQuad overrides the cases
"""
import os, subprocess, sys
CODE = r'''
import os, sys, time; sys.path.insert(0, '.')
import numpy as np, torch
import bench
from gumbi_amd import engine
cases = os.environ.get("AB_CASES", "50000:8:Matern52,50000:8:ExpQuad,100000:8:ExpQuad,10000:4:ExpQuad,30000:2:Matern32,30000:16:Matern52")
for c in cases.split(","):
    N, d, kind = c.split(":"); N = int(N); d = int(d)
    X, y, ls = bench.synthetic_table(N, d)
    e = engine.Engine(0); e.set_data(X, y); e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d)), kind=kind)); e.set_theta(np.concatenate([ls, [1.0, 0.2]]))
    Nr, Np = (N + 128) // 128 * 128, (N + 127) // 128 * 128
    out = torch.empty((Np, Nr), dtype=torch.float64, device="cuda:0"); torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); e.blk_covariance(out.data_ptr(), Nr); ts.append(time.perf_counter() - t0)
    t = min(ts[1:])
    tf = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out.fill_(1.5); torch.cuda.synchronize(); tf.append(time.perf_counter() - t0)
    fill = out.numel() * 8 / min(tf[1:]) / 1e9
    for _ in range(2): e.blk_covariance(out.data_ptr(), Nr)
    chk = float(out[:N, :N].diagonal().sum().cpu()), float(out[N // 3, N // 2:N].sum().cpu())
    print("N=%d d=%d %s: %.3f ms  %.0f GB/s   (checks %.12g %.12g; torch fill of the whole buffer %.0f GB/s)" % (N, d, kind, t * 1e3, 8.0 * N * (N + 1) / 2 / t / 1e9, chk[0], chk[1], fill))
    e.close(); del out; torch.cuda.empty_cache()
'''
libs = sys.argv[1:] or [None]
for lib in libs:
    env = dict(os.environ)
    if lib:
        env["GUMBI_HIP_LIB"] = os.path.abspath(lib)
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(lib or "in-tree library")
    print("  " + "\n  ".join(l for l in out.stdout.strip().splitlines() if l.startswith("N=")) if out.returncode == 0 else out.stderr[-1500:])
