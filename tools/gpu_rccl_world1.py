"""RCCL smoke on a one-GPU box: world_size = 1 process group on the nccl backend, the multi-GPU
driver end to end (device-tensor collectives on the engine's side stream, no host staging)."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from gumbi_amd.distributed import DistributedEngine  # noqa: E402
from gumbi_amd.engine import Engine, KernelSpec  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402

N, d, M = 3000, 4, 500
X, y, ls = O.synthetic_table(N, d, seed=3)
theta = np.concatenate([ls, [1.0, 0.2]])
Xs = np.random.default_rng(0).standard_normal((M, d))
spec = KernelSpec(D=d, idx_cont=list(range(d)))
de = DistributedEngine(0)
de.set_data(X, y); de.set_kernel(spec); de.set_theta(theta)
de.factorize()
v1 = de.nlml()
mu1, var1 = de.predict(Xs)
de.factorize()
val1, g1 = de.nlml(grad=True)
de.force_partition = True   # row-partitioned inverse + all-gather of U through RCCL
de.factorize()
val3, g3 = de.nlml(grad=True)
assert np.max(np.abs(g3 - g1)) < 1e-9 * max(1.0, np.max(np.abs(g1))), np.max(np.abs(g3 - g1))
e = Engine(0)
e.set_data(X, y); e.set_kernel(spec); e.set_theta(theta); e.factorize()
v2 = e.nlml(); mu2, var2 = e.predict(Xs); e.factorize(); val2, g2 = e.nlml(grad=True)
print("backend", dist.get_backend(), " nlml diff", abs(v1 - v2), " mu diff", np.max(np.abs(mu1 - mu2)),
      " grad diff", np.max(np.abs(g1 - g2)) / max(1.0, np.max(np.abs(g2))))
assert abs(v1 - v2) < 1e-8 and np.max(np.abs(mu1 - mu2)) < 1e-10 and np.max(np.abs(g1 - g2)) < 1e-8 * max(1.0, np.max(np.abs(g2)))
de.close(); e.close()
dist.destroy_process_group()
print("rccl world-1 smoke ok")
