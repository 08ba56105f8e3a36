"""ONE GP over the GPUs of a node: Python face of the native multi-GPU driver
(``gumbi_amd/csrc/dist_driver.hpp``, C ABI ``gmb_dist_*`` in ``include/gumbi_hip.h``).

One process per GPU.  The panel loop, the look-ahead over HIP streams, the row-partitioned gradient and the
sharded prediction all run inside ``libgumbi_hip.so``; what this module adds is only the TRANSPORT the driver
is handed -- ``gmb_comm``, a single collective (all-gather of float64 device buffers, stream-ordered):

* :class:`RcclComm` -- production: a RCCL communicator of the library's own (``ncclAllGather`` over xGMI,
  called straight from the C++ loop, no Python between the collectives).  The 128-byte unique id is created on
  rank 0 and handed to the other ranks through the already-initialised ``torch.distributed`` group (any
  backend) -- that is all ``torch.distributed`` is used for on this path;
* :class:`TorchDistComm` -- the same contract implemented by a Python callback on ``torch.distributed``
  collectives: ``gloo`` in the tests (several ranks sharing one GPU, host-staged) and a fallback for ``nccl``
  when the library's own communicator cannot be created.

There is no reference counterpart (the reference is single-process; SURVEY.md section 8e defines the
partition); the work replaced is what ``pm.find_MAP`` / ``Marginal.predict`` do per evaluation
(gumbi/regression/pymc/GP.py:811, 845-847).
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

from .engine import Engine, GmbComm, GumbiHipError, load_library

__all__ = ["RcclComm", "TorchDistComm", "DistributedEngine", "make_comm", "owned_blocks"]

BLK = 128


def owned_blocks(rank: int, world: int, start: int, stop: int):
    """(first, count) of the block rows i in [start, stop) with i % world == rank."""
    first = start + ((rank - start) % world)
    if first >= stop:
        return first, 0
    return first, (stop - first + world - 1) // world


def _rccl_path() -> bytes | None:
    """The librccl the process already uses: torch's bundled copy when torch is installed (so that RCCL, torch
    and libgumbi_hip share one library instance), else whatever the loader finds (NULL)."""
    override = os.environ.get("GUMBI_RCCL_LIB")
    if override:
        return override.encode()
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.submodule_search_locations:
            cand = Path(list(spec.submodule_search_locations)[0]) / "lib" / "librccl.so"
            if cand.exists():
                return str(cand).encode()
    except (ImportError, ValueError):
        pass
    return None


class RcclComm:
    """``gmb_comm`` backed by a RCCL communicator created inside libgumbi_hip.so (``gmb_rccl_comm_create``).
    Collective constructor: every rank of ``group`` must call it."""

    def __init__(self, device_index: int, group=None):
        import torch.distributed as dist

        self._lib = load_library()
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        path = _rccl_path()
        uid = (C.c_char * 128)()
        err = None
        if self.rank == 0:
            rc = self._lib.gmb_rccl_unique_id(path, uid)
            if rc != 0:
                err = (self._lib.gmb_rccl_last_error() or b"").decode()
        box = [bytes(uid.raw) if err is None else None, err]
        if self.world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if box[0] is None:
            raise GumbiHipError(f"RCCL unique id could not be created on rank 0: {box[1]}")
        uid_buf = (C.c_char * 128).from_buffer_copy(box[0])
        self._comm = C.POINTER(GmbComm)()
        rc = self._lib.gmb_rccl_comm_create(C.byref(self._comm), path, uid_buf, self.rank, self.world, int(device_index))
        if rc != 0:
            raise GumbiHipError("gmb_rccl_comm_create failed: " + (self._lib.gmb_rccl_last_error() or b"").decode())
        self.kind = "rccl"

    @property
    def handle(self):
        return self._comm

    @property
    def ranks(self) -> int:
        """``ncclCommCount`` of the communicator: the rank count RCCL itself reports."""
        return int(self._lib.gmb_rccl_comm_ranks(self._comm))

    @property
    def two_communicators(self) -> bool:
        """True when the transport holds a second communicator (``ncclCommSplit``) for the collectives the driver issues off its main
        stream -- RCCL runs one communicator's collectives in issue order whatever their streams (``gmb_rccl_comm_split``)."""
        return int(self._lib.gmb_rccl_comm_split(self._comm)) == 1

    def close(self):
        if getattr(self, "_comm", None):
            self._lib.gmb_rccl_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _RawDeviceArray:
    """Zero-copy handle on HBM for ``torch.as_tensor`` (``__cuda_array_interface__``)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


class TorchDistComm:
    """``gmb_comm`` whose all-gather is a Python callback on ``torch.distributed`` (``gloo``: host-staged;
    ``nccl``: RCCL through torch, ordered on the driver's stream)."""

    def __init__(self, device_index: int, group=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.device = torch.device("cuda", device_index)
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.host_staged = dist.get_backend(group) == "gloo"
        self.kind = "torch-" + dist.get_backend(group)
        self.error = None
        self._cb = GmbComm.ALL_GATHER(self._all_gather)  # keep the trampoline alive
        self._struct = GmbComm(self.rank, self.world, None, self._cb)
        self._streams = {}

    @property
    def handle(self):
        return C.pointer(self._struct)

    def _all_gather(self, _ctx, send, recv, count, stream):
        torch = self.torch
        try:
            st = self._streams.get(stream)
            if st is None:
                st = self._streams[stream] = torch.cuda.ExternalStream(stream, device=self.device)
            inp = torch.as_tensor(_RawDeviceArray(send, count), device=self.device)
            out = torch.as_tensor(_RawDeviceArray(recv, count * self.world), device=self.device)
            with torch.cuda.stream(st):
                if self.host_staged:
                    st.synchronize()
                    h_in = inp.cpu()
                    h_out = torch.empty(count * self.world, dtype=torch.float64)
                    self.dist.all_gather_into_tensor(h_out, h_in, group=self.group)
                    out.copy_(h_out)
                    st.synchronize()
                else:
                    self.dist.all_gather_into_tensor(out, inp, group=self.group)
            return 0
        except Exception as err:  # nothing may propagate through the C frames
            self.error = err
            return -1

    def close(self):
        pass


def _all_ranks_ok(ok: bool, group, dev) -> bool:
    """MIN over the group's ranks of a local flag, through ``torch.distributed`` (not the communicator under test)."""
    import torch
    import torch.distributed as dist

    if dist.get_world_size(group) == 1:
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(int(flag.item()))


def _handshake(comm, device_index: int, group=None, deadline_s: float | None = None) -> bool:
    """One all-gather of four doubles per rank through ``comm`` (the ``gmb_comm`` the native driver will call), verified:
    slot r of the result must hold rank r's values.  Collective over ``group``; every rank returns the same answer.

    Everything that can fail LOCALLY (allocations, the handle) happens first and the ranks agree on it through
    ``torch.distributed`` -- a rank that cannot even start must not leave its peers inside the RCCL call.  The all-gather
    itself runs on a side stream and is awaited by polling an event against a deadline (``GUMBI_DIST_HANDSHAKE_S``, default
    60 s): a communicator that delivers nothing is reported as failed instead of blocking the process for ever."""
    import time

    import torch
    import torch.distributed as dist

    dev = torch.device("cuda", device_index)
    agree_dev = dev if dist.get_backend(group) == "nccl" else "cpu"
    deadline_s = float(os.environ.get("GUMBI_DIST_HANDSHAKE_S", "60")) if deadline_s is None else deadline_s
    n = 4
    h = send = recv = side = done = None
    try:
        h = comm.handle.contents
        send = torch.arange(n, dtype=torch.float64, device=dev) + 1000.0 * (h.rank + 1)
        recv = torch.full((h.world * n,), -1.0, dtype=torch.float64, device=dev)
        side = torch.cuda.Stream(device=dev)
        done = torch.cuda.Event()
        torch.cuda.synchronize(dev)
        ready = True
    except Exception:  # noqa: BLE001 -- whatever went wrong, the caller falls back to torch.distributed's collectives
        ready = False
    if not _all_ranks_ok(ready, group, agree_dev):
        return False
    delivered = False
    try:
        rc = h.all_gather(h.ctx, send.data_ptr(), recv.data_ptr(), n, side.cuda_stream)
        done.record(side)
        t_end = time.perf_counter() + deadline_s
        while not done.query() and time.perf_counter() < t_end:
            time.sleep(0.002)
        if done.query():
            want = (torch.arange(n, dtype=torch.float64)[None, :] + 1000.0 * (torch.arange(h.world, dtype=torch.float64)[:, None] + 1)).reshape(-1)
            delivered = rc == 0 and bool(torch.equal(recv.cpu(), want))
        else:
            import sys

            sys.stderr.write(f"[gumbi_amd] first all-gather through the library's RCCL communicator did not complete within {deadline_s:.0f} s "
                             f"on rank {h.rank}\n")
    except Exception:  # noqa: BLE001
        delivered = False
    if not delivered:
        comm._handshake_buffers = (send, recv, side, done)  # a collective that may still be in flight keeps its buffers
    return _all_ranks_ok(delivered, group, agree_dev)


def make_comm(device_index: int, group=None, prefer: str | None = None):
    """The transport for ``group``: the library's own RCCL communicator when the group runs on ``nccl`` (one
    rank per GPU), the ``torch.distributed`` callback otherwise (``gloo`` tests) or when RCCL cannot be set up
    (all ranks agree on the choice).  ``prefer`` / ``GUMBI_DIST_COMM`` = "rccl" | "torch" pins it."""
    import torch
    import torch.distributed as dist

    prefer = prefer or os.environ.get("GUMBI_DIST_COMM")
    backend = dist.get_backend(group)
    if prefer == "torch" or (prefer is None and backend != "nccl"):
        return TorchDistComm(device_index, group)
    comm, ok = None, 1
    try:
        comm = RcclComm(device_index, group)
    except Exception as err:
        if prefer == "rccl" and dist.get_world_size(group) == 1:
            raise
        ok, comm = 0, None
        import sys

        sys.stderr.write(f"[gumbi_amd] own RCCL communicator unavailable on rank {dist.get_rank(group)}: {err}; "
                         f"using torch.distributed collectives\n")
    dev = torch.device("cuda", device_index) if backend == "nccl" else "cpu"
    if dist.get_world_size(group) > 1:
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        ok = int(flag.item())
    if ok:
        # first contact: one small all-gather through the new communicator, checked on every rank, before the driver's panel
        # loop depends on it (a communicator that delivers the wrong ranks' data -- or none -- is replaced, not trusted);
        # the ranks agree on the outcome inside _handshake
        ok = 1 if _handshake(comm, device_index, group) else 0
        if not ok:
            import sys

            sys.stderr.write(f"[gumbi_amd] own RCCL communicator failed its first all-gather on some rank (this is rank "
                             f"{dist.get_rank(group)}); using torch.distributed collectives\n")
    if ok:
        return comm
    if comm is not None:
        comm.close()
    return TorchDistComm(device_index, group)


class DistributedEngine:
    """Same calls as :class:`gumbi_amd.engine.Engine` (``set_data / set_kernel / set_theta / factorize / nlml /
    predict``), executed by the native multi-GPU driver over the ranks of ``group``.  Every rank makes the same
    calls with the same arguments and receives the same results."""

    #: the evaluations do not run on the host's BLAS (HipGP.find_MAP keeps OpenBLAS to one thread inside the optimiser's loop)
    host_blas_free = True

    def __init__(self, device_index: int, group=None, comm=None, panel_blocks: int = 0, capacity: bool = False):
        import torch

        torch.cuda.set_device(device_index)
        self.device_index = device_index
        self.eng = Engine(device_index)
        self.comm = comm if comm is not None else make_comm(device_index, group)
        self.panel_blocks = panel_blocks
        self.spec = None
        # capacity mode: no rank holds the whole factor -- its own block rows + two panel buffers only, L streamed through
        # again for the gradient and the prediction (csrc/dist_capacity.hpp); for N beyond one GPU
        self.capacity = bool(capacity)
        if self.capacity:
            self.eng.set_dist_mode(Engine.DIST_CAPACITY)

    # -- model definition (replicated) ---------------------------------------------------------------
    def set_data(self, X, y):
        self.eng.set_data(X, y)

    def set_kernel(self, spec):
        self.eng.set_kernel(spec)
        self.spec = spec

    def set_theta(self, theta):
        self.eng.set_theta(theta)

    @property
    def N(self):
        return self.eng.N

    @property
    def D(self):
        return self.eng.D

    # -- hot path --------------------------------------------------------------------------------------
    def _raise_transport(self):
        err = getattr(self.comm, "error", None)
        if err is not None:
            self.comm.error = None
            raise GumbiHipError(f"collective failed on rank {self.comm.rank}: {err!r}") from err

    def factorize(self):
        try:
            self.eng.dist_factorize(self.comm, self.panel_blocks)
        except GumbiHipError:
            self._raise_transport()
            raise

    def nlml(self, grad: bool = False):
        try:
            return self.eng.dist_nlml(self.comm, grad)
        except GumbiHipError:
            self._raise_transport()
            raise

    def predict(self, Xs, with_noise=True):
        """Every rank passes the same ``Xs``; the points are sharded over the ranks and every rank receives
        the complete (mean, var)."""
        try:
            return self.eng.dist_predict(self.comm, np.asarray(Xs, dtype=np.float64), with_noise)
        except GumbiHipError:
            self._raise_transport()
            raise

    # -- introspection: the resident state is replicated, the single-engine accessors apply ------------
    def notpd_index(self):
        return self.eng.notpd_index()

    def factor_is_current(self):
        return self.eng.factor_is_current()  # False after a gradient: U = L^-T sits in the factor buffer

    def copy_factor(self, *args, **kwargs):
        return self.eng.copy_factor(*args, **kwargs)

    def copy_v(self):
        return self.eng.copy_v()

    def copy_alpha(self):
        return self.eng.copy_alpha()

    def timings(self):
        return self.eng.timings()

    def set_profiling(self, on):
        self.eng.set_profiling(on)

    def close(self):
        self.eng.close()
        if self.comm is not None:
            self.comm.close()
            self.comm = None
