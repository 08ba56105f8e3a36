"""The covariance build alone (gmb_blk_covariance into a scratch buffer) a few times, for a counter pass
(tools/gpu_pmc_script.sh): KB_CASE = N:d:kind (default 50000:8:Matern52 = C3), KB_REPS (5)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from gumbi_amd import engine

N, d, kind = os.environ.get("KB_CASE", "50000:8:Matern52").split(":")
N, d = int(N), int(d)
X, y, ls = bench.synthetic_table(N, d)
e = engine.Engine(0)
e.set_data(X, y)
e.set_kernel(engine.KernelSpec(D=d, idx_cont=list(range(d)), kind=kind))
e.set_theta(np.concatenate([ls, [1.0, 0.2]]))
Nr, Np = (N + 128) // 128 * 128, (N + 127) // 128 * 128
out = torch.empty((Np, Nr), dtype=torch.float64, device="cuda:0")
torch.cuda.synchronize()
for _ in range(int(os.environ.get("KB_REPS", "5"))):
    e.blk_covariance(out.data_ptr(), Nr)
print(N, d, kind, float(out[N // 3, N // 2:N].sum().cpu()))
e.close()
