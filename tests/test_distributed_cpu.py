"""The multi-GPU Cholesky SCHEDULE on CPU: the native driver (gumbi_amd/csrc/dist_driver.hpp) builds its
panel loop as a plan first (pure integer arithmetic, exported as ``gmb_dist_plan``) and then executes it with
HIP kernels.  Here world_size-2 / -3 ``gloo`` processes replay the SAME plan with numpy blocks built on the
oracle (test infrastructure only) and real all-gathers of the message sizes the plan states: ownership
arithmetic, message sizes, step order and the assembled factor are checked against LAPACK.  The HIP execution
of the plan is checked by tests/test_gpu_distributed.py."""
import os
import socket

import numpy as np
import pytest

from gumbi_amd import engine as E
from gumbi_amd.distributed import BLK, owned_blocks
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


@pytest.mark.parametrize("N,world,w", [(1000, 2, 2), (100_000, 8, 0), (100_000, 8, 8), (50_000, 4, 16), (130, 3, 1),
                                       (4096, 2, 8), (12_800, 8, 4), (100, 3, 0), (128, 2, 1), (1, 2, 0), (127, 8, 3)])
def test_plan_is_consistent_across_ranks(N, world, w):
    """Every rank issues the same sequence of collectives with the same sizes (anything else deadlocks RCCL);
    the rows named by SQUARE / PANEL / UPDATE steps partition their ranges; panels tile the columns."""
    plans = [E.dist_plan(N, r, world, w) for r in range(world)]
    nct, nrt = -(-N // BLK), -(-(N + 1) // BLK)
    coll = [[(s["op"], s["c0"], s["c1"], s["lo"], s["hi"], s["elems"], s["stream"]) for s in p if s["op"] in ("SQUARE", "PANEL", "TAIL")]
            for p in plans]
    assert all(c == coll[0] for c in coll)
    assert [len(p) for p in plans] == [len(plans[0])] * world
    squares = [c for c in coll[0] if c[0] == "SQUARE"]
    assert squares[0][1] == 0 and squares[-1][2] == nct
    assert all(a[2] == b[1] for a, b in zip(squares, squares[1:]))  # panels tile [0, nct)
    for i, step0 in enumerate(plans[0]):
        steps = [p[i] for p in plans]
        assert len({s["op"] for s in steps}) == 1
        op = step0["op"]
        if op in ("SQUARE", "PANEL", "TAIL", "SOLVE", "UPDATE", "KBUILD"):
            hi = nrt if op in ("UPDATE", "KBUILD") else step0["hi"]
            rows = sorted(s["first"] + t * world for s in steps for t in range(s["count"]))
            assert rows == list(range(step0["lo"], hi))
        if op in ("SQUARE", "PANEL", "TAIL"):
            assert step0["elems"] == step0["maxcount"] * BLK * (step0["c1"] - step0["c0"]) * BLK
            assert step0["maxcount"] == max(s["count"] for s in steps)
            assert step0["stream"] == (2 if op == "TAIL" else 0)  # the large piece of a panel column travels beside the chain
        if op == "SOLVE":
            assert step0["elems"] == 0 and step0["hi"] == nrt  # no communication; every row below the square
    # a panel column = SOLVE of all rows below the square, then its HEAD (the next square's rows: all the chain needs) and TAIL
    p0 = plans[0]
    for i, st in enumerate(p0):
        if st["op"] == "SOLVE":
            head = p0[i + 1]
            assert head["op"] == "PANEL" and (head["c0"], head["c1"], head["lo"]) == (st["c0"], st["c1"], st["lo"])
            width = max(q["c1"] - q["c0"] for q in p0 if q["op"] == "SQUARE")
            assert head["hi"] == min(st["lo"] + width, nrt)
            if head["hi"] < nrt:
                tail = p0[i + 2]
                assert tail["op"] == "TAIL" and (tail["lo"], tail["hi"]) == (head["hi"], nrt)
            else:
                assert i + 2 >= len(p0) or p0[i + 2]["op"] != "TAIL"
            # U1 of this panel reads rows [c1, c2) of the column from the other ranks: inside the HEAD
            u1 = [q for q in p0 if q["op"] == "UPDATE" and (q["c0"], q["c1"]) == (st["c0"], st["c1"]) and q["stream"] == 0]
            assert all(head["lo"] <= q["lo"] and q["hi"] <= head["hi"] for q in u1)
    # every trailing block (i, j), j <= i, receives the update of every earlier panel exactly once
    ups = [s for s in plans[0] if s["op"] == "UPDATE"]
    for c0, c1 in {(s["c0"], s["c1"]) for s in ups}:
        cols = sorted((s["lo"], s["hi"]) for s in ups if (s["c0"], s["c1"]) == (c0, c1))
        assert cols[0][0] == c1 and cols[-1][1] == nct and all(a[1] == b[0] for a, b in zip(cols, cols[1:]))


class NumpyRank:
    """One rank's full-size buffer and the numpy versions of the plan's steps."""

    def __init__(self, spec, theta, X, y, rank, world):
        self.rank, self.G = rank, world
        self.N = N = len(y)
        self.Np = -(-N // BLK) * BLK
        self.Nr = -(-(N + 1) // BLK) * BLK
        S = np.zeros((self.Nr, self.Np))
        S[:N, :N] = O.sigma_matrix(spec, theta, X, "direct")
        S[N, :N] = y
        for j in range(N, self.Np):
            S[j, j] = 1.0
        self._sigma = np.tril(S)
        self.A = np.full((self.Nr, self.Np), np.nan)  # poison: rows this rank never built / received stay NaN
        self.logdet = 0.0
        self.pending = []  # TAILs issued and not yet waited for

    def rows(self, s):
        return [s["first"] + t * self.G for t in range(s["count"])]

    def kbuild(self, s):
        for i in self.rows(s):
            ncols = min(i + 1, self.Np // BLK) * BLK
            self.A[i * BLK:(i + 1) * BLK, :ncols] = self._sigma[i * BLK:(i + 1) * BLK, :ncols]

    def _exchange(self, s, dist, torch):
        c0, c1, G = s["c0"] * BLK, s["c1"] * BLK, self.G
        W, mc = c1 - c0, s["maxcount"]
        send = np.zeros((mc, BLK, W))
        for t, i in enumerate(self.rows(s)):
            send[t] = np.nan_to_num(self.A[i * BLK:(i + 1) * BLK, c0:c1])
        assert send.size == s["elems"]
        t_in = torch.from_numpy(send.ravel().copy())
        t_out = torch.empty(send.size * G, dtype=torch.float64)
        dist.all_gather_into_tensor(t_out, t_in)
        recv = t_out.numpy().reshape(G, mc, BLK, W)

        def deliver():
            for q in range(G):
                first, cnt = owned_blocks(q, G, s["lo"], s["hi"])
                for t in range(cnt):
                    i = first + t * G
                    self.A[i * BLK:(i + 1) * BLK, c0:c1] = recv[q, t]

        # a TAIL travels on the communication stream: nothing of it may be assumed to have arrived before the next FORK (the
        # bulk stream's wait) -- the replay delivers it only then, so a main-stream step that read it would see the poison
        if s["op"] == "TAIL":
            self.pending.append(deliver)
        else:
            deliver()

    def fork(self):
        for deliver in self.pending:
            deliver()
        self.pending = []

    def square(self, s, dist, torch):
        self._exchange(s, dist, torch)
        c0, c1 = s["c0"] * BLK, s["c1"] * BLK
        nv = min(c1, self.N) - c0  # rows / columns >= N inside the square: y row and identity padding
        blk = self.A[c0:c1, c0:c1]
        L11 = np.linalg.cholesky(np.tril(blk[:nv, :nv]) + np.tril(blk[:nv, :nv], -1).T)
        blk[:nv, :nv] = L11
        if nv < c1 - c0:
            blk[nv:, :nv] = np.linalg.solve(L11, blk[nv:, :nv].T).T
        self.logdet += float(np.sum(np.log(np.diag(L11))))

    def solve(self, s):
        """SOLVE: this rank's rows below the square against it, in place (no communication)."""
        c0, c1 = s["c0"] * BLK, s["c1"] * BLK
        nv = min(c1, self.N) - c0
        L11 = np.tril(self.A[c0:c0 + nv, c0:c0 + nv])
        for i in self.rows(s):
            rows = slice(i * BLK, (i + 1) * BLK)
            self.A[rows, c0:c0 + nv] = np.linalg.solve(L11, self.A[rows, c0:c0 + nv].T).T

    def panel(self, s, dist, torch):
        """PANEL (the head of the panel column) and TAIL (the rest): everybody's solved rows [lo, hi)."""
        self._exchange(s, dist, torch)

    def update(self, s):
        c0, c1, lo, hi = s["c0"] * BLK, s["c1"] * BLK, s["lo"], s["hi"]
        panel_rows = self.A[lo * BLK:hi * BLK, c0:c1]
        assert not np.isnan(panel_rows).any()  # the all-gathers must have delivered the whole panel
        for i in self.rows(s):
            rows = slice(i * BLK, (i + 1) * BLK)
            top = min(i + 1, hi)  # lower-triangular tiles only
            if top <= lo:
                continue
            self.A[rows, lo * BLK:top * BLK] -= self.A[rows, c0:c1] @ panel_rows[:(top - lo) * BLK].T


class NumpyRankCapacity(NumpyRank):
    """The same plan in the driver's CAPACITY mode (csrc/dist_capacity.hpp): a rank stores ONLY its own block rows
    (``own[i]``: 128 x Np) and two column-panel buffers with all rows (``pan[p & 1]``); squares are factored on gathered copies,
    a panel buffer is poisoned before it is refilled -- an update that reads a panel after its buffer has been reused (a
    missing JOIN in the plan) would see NaN."""

    def __init__(self, spec, theta, X, y, rank, world):
        super().__init__(spec, theta, X, y, rank, world)
        self.A = None  # nobody holds the matrix
        self.own, self.pan, self.pan_cols, self.npanel = {}, [None, None], [None, None], -1
        self.v = np.full(self.Np, np.nan)

    def kbuild(self, s):
        for i in self.rows(s):
            self.own[i] = np.tril(self._sigma)[i * BLK:(i + 1) * BLK].copy()

    def _panel_of(self, c0):
        for k in (0, 1):
            if self.pan_cols[k] is not None and self.pan_cols[k][0] == c0:
                return self.pan[k]
        raise AssertionError("the panel's buffer has been reused")

    def _exchange(self, s, dist, torch):
        c0, c1, G = s["c0"] * BLK, s["c1"] * BLK, self.G
        W, mc = c1 - c0, s["maxcount"]
        send = np.zeros((mc, BLK, W))
        for t, i in enumerate(self.rows(s)):
            send[t] = self.own[i][:, c0:c1]
        assert send.size == s["elems"]
        t_out = torch.empty(send.size * G, dtype=torch.float64)
        dist.all_gather_into_tensor(t_out, torch.from_numpy(send.ravel().copy()))
        recv = t_out.numpy().reshape(G, mc, BLK, W)
        buf = self._panel_of(s["c0"])

        def deliver():
            for q in range(G):
                first, cnt = owned_blocks(q, G, s["lo"], s["hi"])
                for t in range(cnt):
                    i = first + t * G
                    buf[i * BLK:(i + 1) * BLK] = recv[q, t]
            if s["hi"] * BLK >= self.Nr:  # the piece that holds the last block row brings the y row
                self.v[s["c0"] * BLK:s["c1"] * BLK] = buf[self.N]

        if s["op"] == "TAIL":
            self.pending.append(deliver)
        else:
            deliver()

    def square(self, s, dist, torch):
        self.npanel += 1
        k = self.npanel & 1
        self.pan[k] = np.full((self.Nr, (s["c1"] - s["c0"]) * BLK), np.nan)  # poison the buffer being reused
        self.pan_cols[k] = (s["c0"], s["c1"])
        self._exchange(s, dist, torch)
        c0, c1 = s["c0"] * BLK, s["c1"] * BLK
        nv = min(c1, self.N) - c0
        blk = self.pan[k][c0:c1]
        L11 = np.linalg.cholesky(np.tril(blk[:nv, :nv]) + np.tril(blk[:nv, :nv], -1).T)
        blk[:nv, :nv] = L11
        if nv < c1 - c0:
            blk[nv:, :nv] = np.linalg.solve(L11, blk[nv:, :nv].T).T
        self.logdet += float(np.sum(np.log(np.diag(L11))))
        for i in self.rows(s):  # the owners' copies of the square's rows become L too
            self.own[i][:, c0:c1] = self.pan[k][i * BLK:(i + 1) * BLK]
        if s["hi"] * BLK >= self.Nr:
            self.v[c0:c1] = self.pan[k][self.N]

    def solve(self, s):
        c0, c1 = s["c0"] * BLK, s["c1"] * BLK
        nv = min(c1, self.N) - c0
        buf = self._panel_of(s["c0"])
        L11 = np.tril(buf[c0:c0 + nv, :nv])
        for i in self.rows(s):
            self.own[i][:, c0:c0 + nv] = np.linalg.solve(L11, self.own[i][:, c0:c0 + nv].T).T
            buf[i * BLK:(i + 1) * BLK] = self.own[i][:, c0:c1]  # (U1 reads this rank's rows in the panel buffer before the TAIL arrives)

    def panel(self, s, dist, torch):
        self._exchange(s, dist, torch)

    def update(self, s):
        c0, c1, lo, hi = s["c0"] * BLK, s["c1"] * BLK, s["lo"], s["hi"]
        buf = self._panel_of(s["c0"])
        panel_rows = buf[lo * BLK:hi * BLK]
        assert not np.isnan(panel_rows).any()
        for i in self.rows(s):
            top = min(i + 1, hi)
            if top <= lo:
                continue
            assert not np.isnan(buf[i * BLK:(i + 1) * BLK]).any()
            self.own[i][:, lo * BLK:top * BLK] -= buf[i * BLK:(i + 1) * BLK] @ panel_rows[:(top - lo) * BLK].T


def _worker(rank, world, port, N, d, w, out, capacity=False):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, y, ls = O.synthetic_table(N, d, seed=5)
        spec = O.make_spec(d, range(d), kind="Matern52")
        theta = O.pack_theta(spec, ls, 1.1, 0.3)
        me = (NumpyRankCapacity if capacity else NumpyRank)(spec, theta, X, y, rank, world)
        for s in E.dist_plan(N, rank, world, w):
            if s["op"] == "KBUILD":
                me.kbuild(s)
            elif s["op"] == "SQUARE":
                me.square(s, dist, torch)
            elif s["op"] == "SOLVE":
                me.solve(s)
            elif s["op"] in ("PANEL", "TAIL"):
                me.panel(s, dist, torch)
            elif s["op"] == "UPDATE":
                me.update(s)
            elif s["op"] == "FORK":
                me.fork()
        me.fork()  # (the last TAIL: the driver's closing wait for the communication stream)
        L_ref, v_ref = O.factorize(spec, theta, X, y, dist_mode="direct")
        if capacity:  # every rank holds its own block rows only: collect them (test only) and look at the whole
            rows = [None] * world
            dist.all_gather_object(rows, {i: a for i, a in me.own.items()})
            me.A = np.full((me.Nr, me.Np), np.nan)
            for part in rows:
                for i, a in part.items():
                    me.A[i * BLK:(i + 1) * BLK] = a
            assert sorted(i for part in rows for i in part) == list(range(me.Nr // BLK))
            me.A[N, :N] = me.v[:N]
        got = np.tril(me.A[:N, :N])  # the COMPLETE factor is on every rank
        err_L = np.max(np.abs(got - L_ref)) / np.max(np.abs(L_ref))
        err_v = np.max(np.abs(me.A[N, :N] - v_ref)) / np.max(np.abs(v_ref))
        err_ld = abs(me.logdet - np.sum(np.log(np.diag(L_ref))))
        out.put((rank, err_L, err_v, err_ld))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,N,w", [(2, 300, 1), (2, 256, 2), (2, 130, 1), (3, 700, 2), (2, 1100, 4), (3, 900, 0),
                                       (3, 100, 0), (2, 128, 1), (4, 130, 2), (3, 384, 8)])
def test_plan_replayed_with_numpy_blocks_over_gloo(world, N, w):
    _replay(world, N, w, False)


@pytest.mark.parametrize("world,N,w", [(2, 300, 1), (3, 700, 2), (2, 1100, 4), (3, 900, 0), (2, 128, 1), (4, 130, 2), (3, 1300, 3)])
def test_plan_replayed_in_capacity_mode_with_own_rows_and_two_panel_buffers(world, N, w):
    """The SAME plan with the capacity mode's storage (csrc/dist_capacity.hpp): own block rows + two panel buffers per rank;
    the assembled factor, v and log-det against LAPACK -- and no update ever reads a panel buffer that has been refilled."""
    _replay(world, N, w, True)


def _replay(world, N, w, capacity):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, 3, w, out, capacity)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == list(range(world))
    for rank, err_L, err_v, err_ld in results:
        assert err_L < 1e-12 and err_v < 1e-12 and err_ld < 1e-10
