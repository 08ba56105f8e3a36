"""Are the ranks of the multi-GPU driver in lock step?  Two ranks sharing the GPU (gloo) evaluate the same sequence of
hyper-parameters; per evaluation every rank reports hashes of (nlml, gradient, v, sampled factor rows).  Any difference
between the ranks breaks the lock step of L-BFGS-B (GP(distributed=True).find_MAP()).

    python tools/gpu_dist_lockstep.py [N d n_thetas]
"""
import hashlib
import os
import socket
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def h(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:10]


def worker(rank, world, port, N, d, n_th, out):
    try:
        _worker(rank, world, port, N, d, n_th, out)
    except BaseException:
        import traceback

        out.put((rank, traceback.format_exc()))
        out.close()
        out.join_thread()
        os._exit(1)


def _worker(rank, world, port, N, d, n_th, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch  # noqa: F401
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gumbi_amd.distributed import DistributedEngine
    from gumbi_amd.engine import KernelSpec
    from oracle import gp_oracle as O

    X, y, ls = O.synthetic_table(N, d, seed=9)
    eng = DistributedEngine(0)
    eng.set_data(X, y)
    eng.set_kernel(KernelSpec(D=d, idx_cont=list(range(d))))
    rng = np.random.default_rng(1)
    rows = [0, 130, N // 2, N - 1]
    res = []
    for it in range(n_th):
        theta = np.concatenate([ls * np.exp(0.3 * rng.standard_normal(d)), [1.0 * np.exp(0.2 * rng.standard_normal()), 0.25]])
        eng.set_theta(theta)
        eng.factorize()
        v = eng.copy_v()
        Lr = np.concatenate([eng.copy_factor(r, 1, 0, r + 1)[0] for r in rows])
        f0 = eng.nlml()
        f, g = eng.nlml(grad=True)
        res.append((h(np.float64(f0)), h(g), h(v), h(Lr), float(f0), g.copy()))
    out.put((rank, res))
    eng.close()
    dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp

    N, d, n_th = (int(v) for v in (sys.argv[1:4] + ["8192", "4", "16"][len(sys.argv) - 1:]))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, N, d, n_th, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, val = out.get(timeout=600)
        if isinstance(val, str):
            print(f"rank {r} failed:\n{val}")
            for p in procs:
                p.kill()
            sys.exit(1)
        got[r] = val
    for p in procs:
        p.join()
    bad = 0
    for it, (a, b) in enumerate(zip(got[0], got[1])):
        same = [x == z for x, z in zip(a[:4], b[:4])]
        flag = "" if all(same) else "   <-- RANKS DIFFER (nlml, grad, v, factor rows) = " + str(same)
        bad += not all(same)
        print(f"eval {it:2d}: nlml {a[4]:.10f} / {b[4]:.10f}  max|dg| {np.max(np.abs(a[5] - b[5])):.3e}{flag}")
    print(f"N={N} d={d}: {bad} of {len(got[0])} evaluations differ between the ranks")


if __name__ == "__main__":
    main()
