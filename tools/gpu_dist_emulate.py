"""What does ONE rank of a G-GPU run spend on the multi-GPU factorisation, compute-side?  Runs the native driver
as rank r of world G on the one GPU of the box with a FAKE transport (the all-gather replicates the rank's own
segment into all G slots: same kernels, same launch pattern, same stream structure as a real run; the collectives
cost a device copy instead of xGMI time, and the numbers in the matrix are garbage, so the factorisation ends with
GMB_ENOTPD -- only the TIMING is meaningful).  Ideal = single-engine time / G.

    python tools/gpu_dist_emulate.py N G [panel_blocks,...] [rank]

DE_LINK_GBS (default 0 = copies only): every all-gather also occupies its stream for (bytes received) / DE_LINK_GBS -- a spin of
that length behind the copy -- so that what the streams hide of a transport of that rate can be read off the probes.
DE_CAPACITY=1: the capacity mode (gmb_dist_set_mode(e, 1)): factorisation, then the gradient and a prediction behind it
(gmb_debug_assume_factored: the numbers are garbage, the launch and collective pattern is the real one).
"""
import ctypes as C
import os

os.environ.setdefault("GUMBI_HIP_DEBUG_DOORS", "1")  # gmb_debug_assume_factored is a tools-only door
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
            if LINK_GBS > 0:  # the time the transfer would take at that rate, spent on the collective's own stream
                torch.cuda._sleep(int(8 * count * (self.world - 1) / (LINK_GBS * 1e9) * CLOCK_HZ))
        self.bytes += 8 * count * (self.world - 1)
        self.calls += 1
        return 0


LINK_GBS = float(os.environ.get("DE_LINK_GBS", "0"))
CLOCK_HZ = 1e3 * torch.cuda.get_device_properties(0).clock_rate if hasattr(torch.cuda.get_device_properties(0), "clock_rate") else 2.4e9
if LINK_GBS > 0:  # calibrate torch.cuda._sleep's cycle against the wall clock
    torch.cuda._sleep(1000)  # (first use: compile / load)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    torch.cuda._sleep(int(0.05 * CLOCK_HZ))
    torch.cuda.synchronize()
    CLOCK_HZ *= 0.05 / (time.perf_counter() - t0)
X, y, ls = O.synthetic_table(N, d)
theta = np.concatenate([ls, [1.0, 0.2]])
if os.environ.get("DE_CAPACITY") == "1":
    spec = KernelSpec(D=d, idx_cont=list(range(d)))
    eng = Engine(0)
    eng.set_data(X, y)
    eng.set_kernel(spec)
    eng.set_theta(theta)
    eng.set_dist_mode(Engine.DIST_CAPACITY)
    comm = FakeComm(rank, G)
    Xs = O.synthetic_grid(d, 100)
    for rep in range(2):
        t0 = time.perf_counter()
        try:
            eng.dist_factorize(comm, widths[0])
        except np.linalg.LinAlgError:
            pass
        torch.cuda.synchronize()
        t_f = time.perf_counter() - t0
        eng.debug_assume_factored()
        comm.bytes = comm.calls = 0
        t0 = time.perf_counter()
        eng.dist_nlml(comm, grad=True)
        torch.cuda.synchronize()
        t_g = time.perf_counter() - t0
        tm = eng.timings()
        gb, calls = comm.bytes / 1e9, comm.calls
        t0 = time.perf_counter()
        eng.dist_predict(comm, Xs)
        torch.cuda.synchronize()
        t_p = time.perf_counter() - t0
    print(f"capacity mode, N={N}, rank {rank} of {G}, transport model {LINK_GBS:.0f} GB/s per rank received: factorisation {t_f * 1e3:.1f} ms, "
          f"gradient {t_g * 1e3:.1f} ms ({calls} all-gathers, {gb:.1f} GB received), prediction of {len(Xs)} points {t_p * 1e3:.1f} ms")
    print(f"    gradient probes: {int(tm['dist_grad_collectives'])} collectives, in collectives {tm['dist_grad_comm_ms']:.1f} ms, of it exposed "
          f"{tm['dist_grad_comm_exposed_ms']:.1f} ms = {tm['dist_grad_comm_exposed_ms'] / max(tm['dist_grad_comm_ms'], 1e-9):.2f}; gradient {tm['grad_ms']:.1f} ms "
          f"(resident {eng.resident_bytes() / 2**30:.1f} GiB, peak {eng.resident_bytes(peak=True) / 2**30:.1f} GiB)")
    eng.close()
    sys.exit(0)
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
