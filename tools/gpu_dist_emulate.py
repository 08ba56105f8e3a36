"""What does ONE rank of a G-GPU run spend on the multi-GPU factorisation, compute-side?  Runs the native driver
as rank r of world G on the one GPU of the box with a FAKE transport (the all-gather replicates the rank's own
segment into all G slots: same kernels, same launch pattern, same stream structure as a real run; the collectives
cost a device copy instead of xGMI time, and the numbers in the matrix are garbage, so the factorisation ends with
GMB_ENOTPD -- only the TIMING is meaningful).  Ideal = single-engine time / G.

    python tools/gpu_dist_emulate.py N G [panel_blocks,...] [rank]
"""
import ctypes as C
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gumbi_amd.engine import Engine, GmbComm, KernelSpec  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
widths = [int(w) for w in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
rank = int(sys.argv[4]) if len(sys.argv) > 4 else G - 1
d = 8
dev = torch.device("cuda", 0)


class _Raw:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


class FakeComm:
    def __init__(self, rank, world):
        self.bytes = 0
        self.calls = 0
        self._cb = GmbComm.ALL_GATHER(self._all_gather)
        self._struct = GmbComm(rank, world, None, self._cb)
        self.world = world
        self._streams = {}

    @property
    def handle(self):
        return C.pointer(self._struct)

    def _all_gather(self, _ctx, send, recv, count, stream):
        st = self._streams.get(stream)
        if st is None:
            st = self._streams[stream] = torch.cuda.ExternalStream(stream, device=dev)
        inp = torch.as_tensor(_Raw(send, count), device=dev)
        out = torch.as_tensor(_Raw(recv, count * self.world), device=dev).view(self.world, count)
        with torch.cuda.stream(st):
            out.copy_(inp.unsqueeze(0).expand(self.world, count))
        self.bytes += 8 * count * (self.world - 1)
        self.calls += 1
        return 0


X, y, ls = O.synthetic_table(N, d)
theta = np.concatenate([ls, [1.0, 0.2]])
spec = KernelSpec(D=d, idx_cont=list(range(d)))
eng = Engine(0)
eng.set_data(X, y)
eng.set_kernel(spec)
eng.set_theta(theta)
eng.factorize()
t0 = time.perf_counter()
eng.factorize()
single = time.perf_counter() - t0
print(f"N={N}: single engine {single * 1e3:.1f} ms ({N**3 / 3 / single / 1e12:.1f} TF/s); ideal per rank at G={G}: {single / G * 1e3:.1f} ms", flush=True)
for w in widths:
    comm = FakeComm(rank, G)
    times = []
    for rep in range(2):
        comm.bytes = comm.calls = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            eng.dist_factorize(comm, w)
        except np.linalg.LinAlgError:
            pass
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    t = min(times)
    print(f"  rank {rank} of {G}, panel_blocks={w}: {t * 1e3:.1f} ms = {single / G / t:.2f} of ideal "
          f"({N**3 / 3 / G / t / 1e12:.1f} TF/s per GPU); {comm.calls} all-gathers, {comm.bytes / 1e9:.1f} GB received "
          f"(= {comm.bytes / 1e9 / 300:.3f} s at 300 GB/s if not hidden)", flush=True)
    tm = eng.timings()  # communication probes of the last (unprofiled) run: what the streams waited for
    print(f"    probes: {int(tm['dist_chol_collectives'])} collectives, {tm['dist_chol_comm_bytes'] / 1e9:.1f} GB, in collectives "
          f"{tm['dist_chol_comm_ms']:.1f} ms (device copies here), of it exposed {tm['dist_chol_comm_exposed_ms']:.1f} ms; main stream "
          f"waited {tm['dist_chol_main_wait_ms']:.1f} ms for U2 at the JOINs, bulk stream waited {tm['dist_chol_bulk_wait_ms']:.1f} ms for the "
          f"panel chain at the FORKs (chain-bound time); Cholesky {tm['chol_ms']:.1f} ms", flush=True)
    eng.set_profiling(True)  # one more run with an event pair around every launch: where the time goes
    try:
        eng.dist_factorize(comm, w)
    except np.linalg.LinAlgError:
        pass
    tm = eng.timings()
    eng.set_profiling(False)
    print(f"    profiled: K-build {tm['kbuild_ms']:.1f} ms, Cholesky {tm['chol_ms']:.1f} ms of which the sum of U1 / U2 launches "
          f"{tm['total_chol_gemm_ms']:.1f} ms ({tm['total_chol_gemm_flops'] / max(tm['total_chol_gemm_ms'], 1e-9) / 1e9:.1f} TF/s, "
          f"{int(tm['total_chol_gemm_launches'])} launches), in-panel products {tm['total_chol_panel_gemm_ms']:.1f} ms, "
          f"leaves {tm['chol_leaf_ms']:.1f} ms, strips {tm['chol_trsm_ms']:.1f} ms", flush=True)
# gradient: every rank holds the complete factor after the factorisation, so a single-engine factorisation puts this
# GPU in exactly that state; the row-partitioned gradient then runs as rank `rank` of G (the other ranks' rows of U
# arrive as copies of this rank's -- garbage values, real timing)
eng.factorize()
t0 = time.perf_counter()
eng.nlml(grad=True)
single_g = time.perf_counter() - t0
comm = FakeComm(rank, G)
times = []
for rep in range(2):
    eng.factorize()
    comm.bytes = comm.calls = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.dist_nlml(comm, grad=True)
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
t = min(times)
tm = eng.timings()
print(f"    probes: {int(tm['dist_grad_collectives'])} collectives, {tm['dist_grad_comm_bytes'] / 1e9:.1f} GB, in collectives "
      f"{tm['dist_grad_comm_ms']:.1f} ms (device copies here), of it exposed {tm['dist_grad_comm_exposed_ms']:.1f} ms; gradient {tm['grad_ms']:.1f} ms")
print(f"  gradient: single engine {single_g * 1e3:.1f} ms; rank {rank} of {G}: {t * 1e3:.1f} ms = {single_g / G / t:.2f} of ideal; "
      f"{comm.calls} all-gathers, {comm.bytes / 1e9:.1f} GB received (= {comm.bytes / 1e9 / 300:.3f} s at 300 GB/s if not hidden)", flush=True)
eng.close()
