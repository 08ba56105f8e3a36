"""Multi-GPU Cholesky: 1-D block-cyclic ROW partition of the covariance / factor over the GPUs of
one node, one process per GPU, collectives through ``torch.distributed`` (backend ``nccl`` = RCCL
over xGMI on ROCm; ``gloo`` in the CPU tests).

Partition.  Block = 128 rows.  Rank g of G owns block rows i = g (mod G):
it builds them (covariance tiles need no communication -- X is replicated), solves them against each
finished diagonal block and applies every trailing update to them.  What is exchanged:

* per block column k: the owner factors the diagonal block and **broadcasts** ``L_kk`` with the
  inverses of its eight 16 x 16 diagonal sub-blocks (128 + 16 KiB, what the strip solve
  ``gmb_blk_trsm`` consumes); every rank solves its rows of the panel; one **all-gather** delivers the finished
  panel column to all ranks (the north star's "panel broadcast" -- with one sender per block row the
  all-gather is what drives all xGMI links at once);
* at the end: an all-reduce of the log-determinant partials and of the failure flag.

Every rank keeps a FULL-size factor buffer (80 GB at N = 1e5 -- inside one MI355X's 288 GB), so
after the last panel each rank holds the complete factor and ``v = L^-1 y`` with no further
exchange: prediction shards the M test points across ranks with zero communication until the final
gather of (mean, var), and it runs the single-GPU ``gmb_predict`` unchanged.

The column recursion is the single-GPU one (``engine.hip: chol_cols``): ``chol(c0,c1) =
chol(c0,mid); update A[mid:, mid:c1] -= L[mid:, c0:mid] L[mid:c1, c0:mid]^T; chol(mid,c1)`` -- the
update is purely local (strided over the owned block rows, ``gmb_blk_gemm_strided``) because the
all-gathers already put ``L[:, c0:mid]`` on every rank, and most of its flops sit in GEMMs with a
long contraction.

The driver is written against two small interfaces so that the orchestration (ownership
arithmetic, message sizes, recursion order) is exercised by world_size-2 ``gloo`` tests on CPU:
``ops`` performs block operations on this rank's buffers (``HipBlockOps`` here -- HIP kernels
through the C ABI, nothing else ships; the tests inject a numpy stand-in) and ``comm`` wraps
``torch.distributed``.
"""

from __future__ import annotations

import numpy as np

__all__ = ["BlockCyclicCholesky", "TorchComm", "HipBlockOps", "DistributedEngine", "owned_blocks"]

BLK = 128
DINV = 8 * 16 * 16  # doubles in the sub-block inverses of one diagonal block


def owned_blocks(rank: int, world: int, start: int, stop: int):
    """(first, count) of the block rows i in [start, stop) with i % world == rank."""
    first = start + ((rank - start) % world)
    if first >= stop:
        return first, 0
    return first, (stop - first + world - 1) // world


class TorchComm:
    """The three collectives the factorisation needs, on torch tensors living wherever ``ops`` keeps
    its staging buffers (HBM for nccl, host for gloo)."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

        # gloo moves host memory only: device tensors are staged through the host (tests, and the
        # 2-process-on-one-GPU validation); nccl/RCCL works on HBM directly
        self.host_staged = dist.get_backend(group) == "gloo"

    def _run(self, fn, *tensors):
        if self.host_staged and any(t.is_cuda for t in tensors):
            host = [t.cpu() for t in tensors]
            fn(*host)
            for t, h in zip(tensors, host):
                t.copy_(h)
        else:
            fn(*tensors)

    def broadcast(self, tensor, src):
        self._run(lambda t: self.dist.broadcast(t, src=src, group=self.group), tensor)

    def all_gather(self, out, inp):
        self._run(lambda o, i: self.dist.all_gather_into_tensor(o, i, group=self.group), out, inp)

    def all_reduce(self, tensor, op="sum"):
        red = self.dist.ReduceOp.SUM if op == "sum" else self.dist.ReduceOp.MAX
        self._run(lambda t: self.dist.all_reduce(t, op=red, group=self.group), tensor)
        return tensor


class BlockCyclicCholesky:
    """Factorisation driver (see module docstring).  ``nct`` / ``nrt`` = number of 128-blocks along
    the columns / rows of the padded factor buffer (rows include the appended y row)."""

    def __init__(self, ops, comm, nct: int, nrt: int):
        self.ops, self.comm = ops, comm
        self.rank, self.world = comm.rank, comm.world
        self.nct, self.nrt = nct, nrt
        self.log = []  # (event, args) trace, used by the tests

    # -- top level ---------------------------------------------------------------------------------
    def factorize(self):
        ops = self.ops
        ops.begin()
        first, cnt = owned_blocks(self.rank, self.world, 0, self.nrt)
        for t in range(cnt):
            ops.build_block_row(first + t * self.world)
        self._cols(0, self.nct)
        logdet, info = ops.local_logdet_info()
        red = ops.scalar_tensor([logdet])
        self.comm.all_reduce(red, "sum")
        flag = ops.scalar_tensor([float(info)])
        self.comm.all_reduce(flag, "max")
        ops.finish(float(red[0]), int(flag[0]))

    def _cols(self, c0, c1):
        if c1 - c0 == 1:
            return self._leaf(c0)
        mid = c0 + (c1 - c0 + 1) // 2
        self._cols(c0, mid)
        first, cnt = owned_blocks(self.rank, self.world, mid, self.nrt)
        self.log.append(("update", c0, mid, c1, first, cnt))
        if cnt > 0:
            self.ops.update(c0, mid, c1, first, cnt, self.world)
        self._cols(mid, c1)

    def _leaf(self, k):
        ops, comm, G = self.ops, self.comm, self.world
        owner = k % G
        stage = ops.diag_stage()
        if self.rank == owner:
            ops.potrf(k)
            ops.diag_to_stage(k)
        comm.broadcast(stage, src=owner)
        if self.rank != owner:
            ops.stage_to_diag(k)
        maxcnt = (self.nrt - (k + 1) + G - 1) // G
        self.log.append(("leaf", k, owner, maxcnt))
        if maxcnt <= 0:
            return
        first, cnt = owned_blocks(self.rank, G, k + 1, self.nrt)
        send, recv = ops.panel_buffers(maxcnt, G)
        ops.pack_panel(k, first, cnt, G, send, maxcnt)
        if cnt > 0:
            ops.solve_packed(k, cnt, send, maxcnt)
        comm.all_gather(recv, send)
        for r in range(G):
            f_r, c_r = owned_blocks(r, G, k + 1, self.nrt)
            if c_r > 0:
                ops.unpack_panel(k, f_r, c_r, G, recv, r, maxcnt)


# ---------------------------------------------------------------------------------------------------
class _RawDeviceArray:
    """Zero-copy handle on engine-owned HBM for ``torch.as_tensor`` (``__cuda_array_interface__``)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


class HipBlockOps:
    """Block operations on one rank's resident factor through the C ABI (``gmb_blk_*``)."""

    def __init__(self, engine, device):
        import torch

        self.torch = torch
        self.eng = engine
        self.device = device
        b = engine.factor_buffers()
        self.A, self.ld, self.Nr, self.Np, self.dinv, self.scal, self.info = (
            b["A"], b["ld"], b["Nr"], b["Np"], b["dinv16"], b["scal"], b["info"])
        self.N = engine.N
        # broadcast unit of one block column: [L_kk (128 x 128) | its eight 16 x 16 sub-block inverses]
        self._stage = torch.zeros(BLK * BLK + DINV, dtype=torch.float64, device=device)
        self._send = self._recv = None
        self._cap = 0

    # pointer arithmetic (bytes)
    def _a(self, row, col):
        return self.A + 8 * (row + col * self.ld)

    def begin(self):
        self.eng.begin_external_factorization()

    def build_block_row(self, i):
        ncols = min(i + 1, self.Np // BLK) * BLK
        self.eng.blk_kbuild(self._a(i * BLK, 0), self.ld, i * BLK, BLK, 0, ncols)

    def potrf(self, k):
        nvalid = min(BLK, self.N - k * BLK)
        self.eng.blk_potrf(self._a(k * BLK, k * BLK), self.ld, nvalid, self.dinv + 8 * k * DINV, self.scal,
                           self.info)

    def diag_stage(self):
        return self._stage

    def diag_to_stage(self, k):
        self.eng.blk_pack(self._a(k * BLK, k * BLK), self.ld, 1, 1, self._stage.data_ptr(), BLK, True)
        inv = self.torch.as_tensor(_RawDeviceArray(self.dinv + 8 * k * DINV, DINV), device=self.device)
        self._stage[BLK * BLK:].copy_(inv)

    def stage_to_diag(self, k):
        self.eng.blk_pack(self._a(k * BLK, k * BLK), self.ld, 1, 1, self._stage.data_ptr(), BLK, False)
        inv = self.torch.as_tensor(_RawDeviceArray(self.dinv + 8 * k * DINV, DINV), device=self.device)
        inv.copy_(self._stage[BLK * BLK:])

    def panel_buffers(self, maxcnt, G):
        need = maxcnt * BLK * BLK
        if self._cap < need:
            self._cap = need
            self._send_full = self.torch.zeros(need, dtype=self.torch.float64, device=self.device)
            self._recv_full = self.torch.zeros(need * G, dtype=self.torch.float64, device=self.device)
        return self._send_full[:need], self._recv_full[: need * G]

    def pack_panel(self, k, first, cnt, G, send, maxcnt):
        if cnt > 0:
            self.eng.blk_pack(self._a(first * BLK, k * BLK), self.ld, G, cnt, send.data_ptr(), maxcnt * BLK, True)

    def solve_packed(self, k, cnt, send, maxcnt):
        nvalid = min(BLK, self.N - k * BLK)
        self.eng.blk_trsm(send.data_ptr(), maxcnt * BLK, cnt * BLK, self._a(k * BLK, k * BLK), self.ld,
                          self.dinv + 8 * k * DINV, nvalid)

    def unpack_panel(self, k, first, cnt, G, recv, r, maxcnt):
        src = recv.data_ptr() + 8 * r * maxcnt * BLK * BLK
        self.eng.blk_pack(self._a(first * BLK, k * BLK), self.ld, G, cnt, src, maxcnt * BLK, False)

    def update(self, c0, mid, c1, first, cnt, G):
        self.eng.blk_gemm_strided(self._a(first * BLK, mid * BLK), self.ld, self._a(mid * BLK, c0 * BLK), self.ld,
                                  self._a(first * BLK, c0 * BLK), self.ld, (c1 - mid) * BLK, cnt * BLK,
                                  (mid - c0) * BLK, -1.0, 1.0, 1, (first - mid) * BLK, G)

    def local_logdet_info(self):
        return self.eng.local_logdet_info()

    def scalar_tensor(self, values):
        return self.torch.tensor(values, dtype=self.torch.float64, device=self.device)

    def finish(self, logdet, info):
        self.eng.finish_external_factorization(logdet, info)


class DistributedEngine:
    """One GP spread over the GPUs of a node: same calls as :class:`gumbi_amd.engine.Engine`
    (``set_data / set_kernel / set_theta / factorize / nlml / predict``), factorisation by
    :class:`BlockCyclicCholesky`, prediction sharded over the test points."""

    def __init__(self, device_index: int, group=None):
        import torch

        from .engine import Engine

        self.torch = torch
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        self.comm = TorchComm(group)
        # one side stream shared by the engine's kernels, torch's copies and the collectives, so
        # everything is ordered on the device without host synchronisation
        self.stream = torch.cuda.Stream(self.device)
        self.eng = Engine(device_index, stream=self.stream.cuda_stream)
        self.ops = None
        # True: L^-T is computed by rows across the ranks (scales with the group size);
        # False: every rank inverts the factor itself and only Sigma^-1 is sharded
        self.partition_inverse = True
        self.force_partition = False  # tests: take the partitioned path even with a single rank

    def set_data(self, X, y):
        self.eng.set_data(X, y)
        self.ops = None

    def set_kernel(self, spec):
        self.eng.set_kernel(spec)

    def set_theta(self, theta):
        self.eng.set_theta(theta)

    def factorize(self):
        with self.torch.cuda.stream(self.stream):
            if self.ops is None:
                self.ops = HipBlockOps(self.eng, self.device)
            driver = BlockCyclicCholesky(self.ops, self.comm, self.ops.Np // BLK, self.ops.Nr // BLK)
            driver.factorize()
        return driver

    def nlml(self, grad: bool = False):
        """NLML of the distributed factorisation; with ``grad=True`` also its gradient: every rank inverts
        the (complete, local) factor, reduces the trace terms over ITS block rows of Sigma^-1
        (``gmb_nlml_shard``), the accumulators are all-reduced and every rank applies the same chain rule,
        so all ranks return bit-identical (value, gradient) and an optimiser stays in lock step."""
        if not grad:
            return self.eng.nlml()
        torch = self.torch
        with torch.cuda.stream(self.stream):
            if self.partition_inverse and (self.comm.world > 1 or self.force_partition):
                acc = self._grad_partitioned()
            else:
                acc = self.eng.nlml_shard(self.comm.rank, self.comm.world)
            t = torch.as_tensor(acc, device=self.device)
            self.comm.all_reduce(t, "sum")
            acc = t.cpu().numpy()
        return self.eng.nlml_from_acc(acc)

    def _grad_partitioned(self):
        """L^-T by rows: rank r solves for ITS block rows of U (``gmb_inv_rows``, N^3/G flops, no
        communication), one all-gather delivers all of U and alpha to every rank (8 N^2 bytes in total),
        then each rank forms its block rows of Sigma^-1 = U U^T and reduces over them."""
        torch = self.torch
        eng, G, r = self.eng, self.comm.world, self.comm.rank
        b = eng.factor_buffers()
        Np, ld = b["Np"], b["ld"]
        nt = Np // BLK
        maxown = (nt + G - 1) // G
        ldv = maxown * BLK
        V = torch.empty(Np * ldv, dtype=torch.float64, device=self.device)      # V[m + c*ldv]
        a_loc = torch.zeros(ldv, dtype=torch.float64, device=self.device)
        self.stream.synchronize()
        eng.inv_rows(r, G, V.data_ptr(), ldv, a_loc.data_ptr())
        recvV = torch.empty(G * Np * ldv, dtype=torch.float64, device=self.device)
        recva = torch.empty(G * ldv, dtype=torch.float64, device=self.device)
        self.comm.all_gather(recvV, V)
        self.comm.all_gather(recva, a_loc)
        # scatter the gathered block rows into the upper triangle of the factor buffer (U[i][k] at
        # i + k*ld) and into alpha; the factor is consumed by the gradient anyway
        A = torch.as_tensor(_RawDeviceArray(b["A"], Np * ld), device=self.device).view(Np, ld)
        alpha = torch.as_tensor(_RawDeviceArray(eng.grad_alpha_ptr(), Np), device=self.device)
        for q in range(G):
            Vq = recvV[q * Np * ldv:(q + 1) * Np * ldv].view(Np, ldv)
            for t in range((nt - q + G - 1) // G if q < nt else 0):
                g0 = (q + t * G) * BLK
                A[:, g0:g0 + BLK].copy_(Vq[:, t * BLK:(t + 1) * BLK])
                alpha[g0:g0 + BLK].copy_(recva[q * ldv + t * BLK:q * ldv + (t + 1) * BLK])
        self.stream.synchronize()
        return eng.nlml_shard_u(r, G)

    def predict(self, Xs, with_noise=True):
        """Every rank passes the same ``Xs``; rank r predicts ``np.array_split`` slice r and the
        slices are all-gathered, so every rank returns the full (mean, var)."""
        torch = self.torch
        Xs = np.ascontiguousarray(Xs, dtype=np.float64)
        with torch.cuda.stream(self.stream):
            return self._predict(Xs, with_noise)

    def _predict(self, Xs, with_noise):
        torch = self.torch
        M, G, r = len(Xs), self.comm.world, self.comm.rank
        bounds = np.linspace(0, M, G + 1).astype(int)
        lo, hi = bounds[r], bounds[r + 1]
        mean, var = self.eng.predict(Xs[lo:hi], with_noise=with_noise) if hi > lo else (np.empty(0), np.empty(0))
        width = int(np.max(np.diff(bounds)))
        send = torch.zeros(2 * width, dtype=torch.float64, device=self.device)
        send[: hi - lo] = torch.as_tensor(mean, device=self.device)
        send[width: width + hi - lo] = torch.as_tensor(var, device=self.device)
        recv = torch.empty(2 * width * G, dtype=torch.float64, device=self.device)
        self.comm.all_gather(recv, send)
        out = recv.cpu().numpy().reshape(G, 2, width)
        means = np.concatenate([out[g, 0, : bounds[g + 1] - bounds[g]] for g in range(G)])
        vars_ = np.concatenate([out[g, 1, : bounds[g + 1] - bounds[g]] for g in range(G)])
        return means, vars_

    def close(self):
        self.eng.close()
