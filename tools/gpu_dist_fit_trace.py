"""GP(distributed=True).fit() over two ranks sharing the GPU (gloo): every rank logs, per objective evaluation, hashes
of the unconstrained parameters it was GIVEN and of the (value, gradient) it RETURNED -- where do the ranks part?

    python tools/gpu_dist_fit_trace.py [N d ls_lower]
"""
import hashlib
import os
import socket
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def h(a):
    return hashlib.sha1(np.ascontiguousarray(np.asarray(a, dtype=np.float64)).tobytes()).hexdigest()[:10]


def _worker(rank, world, port, N, d, lower, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import pandas as pd
    import torch  # noqa: F401
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gumbi_amd as gmb
    from oracle import gp_oracle as O

    X, y, ls = O.synthetic_table(N, d, seed=9)
    cols = [f"x{k}" for k in range(d)]
    df = pd.DataFrame(X, columns=cols)
    df["y"] = y
    ds = gmb.DataSet(df, outputs=["y"])
    gp = gmb.GP(ds, outputs=["y"], distributed=True)
    lsb = gmb.make_deltas_parray(stdzr=ds.stdzr, scale="standardized", **{c: [lower, None] for c in cols}) if lower > 0 else None
    gp.specify_model(continuous_dims=cols)
    gp.build_model(ls_bounds=lsb)
    log = [("prior", h(gp.model.ls_params["alpha"]), h(gp.model.ls_params["beta"]), h(gp._initial_theta()), h(gp.model.X), h(gp.model.y))]
    orig = gp._objective

    def traced(u, pos):
        f, g = orig(u, pos)
        log.append((h(u), h(f), h(g), float(f)))
        if len(log) > 60:
            raise RuntimeError("stop")
        return f, g

    gp._objective = traced
    try:
        gp.find_MAP(maxeval=40)
    except Exception as err:  # the ranks may part company: report what was seen
        log.append(("exception", repr(err)[:200]))
    out.put((rank, log))
    out.close()
    out.join_thread()  # the feeder thread must have flushed before the process goes
    os._exit(0)


def worker(*args):
    try:
        _worker(*args)
    except BaseException:
        import traceback

        args[-1].put((args[0], traceback.format_exc()))
        args[-1].close()
        args[-1].join_thread()
        os._exit(1)


def main():
    import torch.multiprocessing as mp

    a = sys.argv[1:]
    N, d, lower = int(a[0]) if a else 8192, int(a[1]) if len(a) > 1 else 4, float(a[2]) if len(a) > 2 else 0.5
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, N, d, lower, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        r, val = out.get(timeout=240)
        got[r] = val
    for p in procs:
        p.join(timeout=20)
        if p.is_alive():
            p.kill()
    for r in (0, 1):
        if isinstance(got[r], str):
            print(f"rank {r} failed:\n{got[r]}")
            return
    n = max(len(got[0]), len(got[1]))
    for i in range(n):
        a0 = got[0][i] if i < len(got[0]) else None
        a1 = got[1][i] if i < len(got[1]) else None
        print(i, "SAME" if a0 == a1 else "DIFF", a0, a1 if a0 != a1 else "")


if __name__ == "__main__":
    main()
