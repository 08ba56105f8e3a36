"""The native multi-GPU driver on a one-GPU box: world_size = 1 process group on the nccl backend, the
library's own RCCL communicator, timed against the single engine on the same problem (what the panel loop,
the packs / unpacks and the collectives cost when there is nothing to exchange), for a sweep of panel widths.

    python tools/gpu_rccl_world1.py [N] [d] [panel_blocks,...]
"""
import os
import sys
import time
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

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40960
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
widths = [int(w) for w in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 8, 16]
X, y, ls = O.synthetic_table(N, d, seed=3)
theta = np.concatenate([ls, [1.0, 0.2]])
Xs = O.synthetic_grid(d, 100)
spec = KernelSpec(D=d, idx_cont=list(range(d)))


def timed(eng, label):
    eng.set_data(X, y); eng.set_kernel(spec); eng.set_theta(theta)
    eng.factorize(); eng.nlml(grad=True); eng.factorize()  # warm-up incl. workspaces
    t0 = time.perf_counter(); eng.factorize(); t1 = time.perf_counter()
    nl = eng.nlml()
    mu, var = eng.predict(Xs); t2 = time.perf_counter()
    eng.factorize(); t3 = time.perf_counter()
    val, g = eng.nlml(grad=True); t4 = time.perf_counter()
    n3 = float(N) ** 3
    print(f"{label:28s} factorize {t1 - t0:7.3f} s ({n3 / 3 / (t1 - t0) / 1e12:5.1f} TF/s)  predict {t2 - t1:6.3f} s  "
          f"gradient {t4 - t3:7.3f} s ({2 * n3 / 3 / (t4 - t3) / 1e12:5.1f} TF/s)", flush=True)
    return nl, mu, var, val, g


e = Engine(0)
ref = timed(e, "single engine")
e.close()
for w in widths:
    de = DistributedEngine(0, panel_blocks=w)
    assert de.comm.kind == "rccl", de.comm.kind
    got = timed(de, f"dist driver world=1 panel={w}")
    de.close()
    assert abs(got[0] - ref[0]) < 1e-9 * abs(ref[0]) and np.max(np.abs(got[1] - ref[1])) < 1e-9
    assert np.max(np.abs(got[4] - ref[4])) < 1e-7 * max(1.0, np.max(np.abs(ref[4])))
dist.destroy_process_group()
print("rccl world-1 ok")
