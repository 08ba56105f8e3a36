"""The block-cyclic multi-GPU Cholesky driver (gumbi_amd/distributed.py) on CPU: world_size-2
``gloo`` processes, with the rank-local block operations supplied by a numpy stand-in built on the
oracle (test infrastructure only -- the product's ``HipBlockOps`` runs HIP kernels).  Checks the
ownership arithmetic, message sizes, recursion order and the assembled factor against LAPACK."""
import os
import socket

import numpy as np
import pytest

from gumbi_amd.distributed import BLK, BlockCyclicCholesky, TorchComm, owned_blocks
from oracle import gp_oracle as O


def test_owned_blocks_partition():
    for world in (1, 2, 3, 8):
        for start, stop in [(0, 10), (3, 11), (5, 5), (7, 8)]:
            seen = []
            for r in range(world):
                first, cnt = owned_blocks(r, world, start, stop)
                blocks = [first + t * world for t in range(cnt)]
                assert all(b % world == r and start <= b < stop for b in blocks)
                seen += blocks
            assert sorted(seen) == list(range(start, stop))


class NumpyBlockOps:
    """Same interface as HipBlockOps, on a full-size numpy buffer per rank."""

    def __init__(self, spec, theta, X, y):
        import torch

        self.torch = torch
        self.N = N = len(y)
        self.Np = -(-N // BLK) * BLK
        self.Nr = -(-(N + 1) // BLK) * BLK
        S = np.zeros((self.Nr, self.Np))
        S[:N, :N] = O.sigma_matrix(spec, theta, X, "direct")
        S[N, :N] = y
        for j in range(N, self.Np):
            S[j, j] = 1.0
        self._sigma = S
        self.A = np.full((self.Nr, self.Np), np.nan)  # poison: unbuilt rows must never be read
        self.inv = np.full((self.Np // BLK, BLK, BLK), np.nan)
        self._stage = torch.zeros(2 * BLK * BLK, dtype=torch.float64)
        self.logdet, self.info = 0.0, 0

    def begin(self):
        self.logdet, self.info = 0.0, 0

    def build_block_row(self, i):
        ncols = min(i + 1, self.Np // BLK) * BLK
        self.A[i * BLK:(i + 1) * BLK, :ncols] = np.tril(self._sigma)[i * BLK:(i + 1) * BLK, :ncols]

    def potrf(self, k):
        blk = self.A[k * BLK:(k + 1) * BLK, k * BLK:(k + 1) * BLK]
        nv = min(BLK, self.N - k * BLK)
        L11 = np.linalg.cholesky(np.tril(blk[:nv, :nv]) + np.tril(blk[:nv, :nv], -1).T)
        blk[:nv, :nv] = L11
        if nv < BLK:
            blk[nv:, :nv] = np.linalg.solve(L11, blk[nv:, :nv].T).T
        inv = np.eye(BLK)
        inv[:nv, :nv] = np.linalg.inv(L11)
        self.inv[k] = inv
        self.logdet += float(np.sum(np.log(np.diag(L11))))

    def diag_stage(self):
        return self._stage

    def diag_to_stage(self, k):
        blk = self.A[k * BLK:(k + 1) * BLK, k * BLK:(k + 1) * BLK]
        self._stage[:BLK * BLK] = self.torch.from_numpy(np.nan_to_num(blk).ravel().copy())
        self._stage[BLK * BLK:] = self.torch.from_numpy(self.inv[k].ravel().copy())

    def stage_to_diag(self, k):
        st = self._stage.numpy()
        self.A[k * BLK:(k + 1) * BLK, k * BLK:(k + 1) * BLK] = st[:BLK * BLK].reshape(BLK, BLK)
        self.inv[k] = st[BLK * BLK:].reshape(BLK, BLK)

    def panel_buffers(self, maxcnt, G):
        n = maxcnt * BLK * BLK
        return self.torch.zeros(n, dtype=self.torch.float64), self.torch.zeros(n * G, dtype=self.torch.float64)

    def pack_panel(self, k, first, cnt, G, send, maxcnt):
        buf = send.numpy().reshape(maxcnt, BLK, BLK)
        for t in range(cnt):
            i = first + t * G
            buf[t] = self.A[i * BLK:(i + 1) * BLK, k * BLK:(k + 1) * BLK]

    def solve_packed(self, k, cnt, send, maxcnt):
        buf = send.numpy().reshape(maxcnt, BLK, BLK)
        for t in range(cnt):
            buf[t] = buf[t] @ self.inv[k].T

    def unpack_panel(self, k, first, cnt, G, recv, r, maxcnt):
        buf = recv.numpy().reshape(G, maxcnt, BLK, BLK)[r]
        for t in range(cnt):
            i = first + t * G
            self.A[i * BLK:(i + 1) * BLK, k * BLK:(k + 1) * BLK] = buf[t]

    def update(self, c0, mid, c1, first, cnt, G):
        panel_cols = self.A[mid * BLK:c1 * BLK, c0 * BLK:mid * BLK]
        for t in range(cnt):
            i = first + t * G
            rows = slice(i * BLK, (i + 1) * BLK)
            hi = min(i + 1, c1)  # lower-triangular tiles only
            if hi <= mid:
                continue
            self.A[rows, mid * BLK:hi * BLK] -= self.A[rows, c0 * BLK:mid * BLK] @ panel_cols[:(hi - mid) * BLK].T

    def local_logdet_info(self):
        return self.logdet, self.info

    def scalar_tensor(self, values):
        return self.torch.tensor(values, dtype=self.torch.float64)

    def finish(self, logdet, info):
        self.total_logdet, self.total_info = logdet, info


def _worker(rank, world, port, N, d, out):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, y, ls = O.synthetic_table(N, d, seed=5)
        spec = O.make_spec(d, range(d), kind="Matern52")
        theta = O.pack_theta(spec, ls, 1.1, 0.3)
        ops = NumpyBlockOps(spec, theta, X, y)
        drv = BlockCyclicCholesky(ops, TorchComm(), ops.Np // BLK, ops.Nr // BLK)
        drv.factorize()
        L_ref, v_ref = O.factorize(spec, theta, X, y, dist_mode="direct")
        # off-diagonal blocks of every finished panel are on every rank; diagonal blocks too (broadcast)
        got = np.tril(ops.A[:N, :N])
        err_L = np.max(np.abs(got - L_ref)) / np.max(np.abs(L_ref))
        err_v = np.max(np.abs(ops.A[N, :N] - v_ref)) / np.max(np.abs(v_ref))
        err_ld = abs(ops.total_logdet - np.sum(np.log(np.diag(L_ref))))
        leaves = [e for e in drv.log if e[0] == "leaf"]
        out.put((rank, err_L, err_v, err_ld, ops.total_info, [e[1] for e in leaves], [e[2] for e in leaves]))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,N", [(2, 300), (2, 256), (2, 130), (3, 700)])
def test_block_cyclic_cholesky_gloo(world, N):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, 3, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    nct = -(-N // BLK)
    for rank, err_L, err_v, err_ld, info, leaf_ks, owners in results:
        assert err_L < 1e-12 and err_v < 1e-12 and err_ld < 1e-10 and info == 0
        assert leaf_ks == list(range(nct)) and owners == [k % world for k in range(nct)]
