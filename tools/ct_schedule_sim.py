"""Discrete-event model of the persistent tile Cholesky's ticket schedule (csrc/chol_tiles.hpp), CPU only.

Workers (one 8-wave workgroup per compute unit) draw tickets in a fixed order; a task walks its k-blocks in order, each one
as soon as the two tiles it reads are published, then factors (diagonal tile) or solves (tile below).  The cost constants are
read off traces of the real kernel (tools/gpu_chol_tiles.py, CT_TRACE_SAVE): contraction 15.1 us per k-block + 10 us per task,
leaf 25.5 us, strip solve of a bulk tile 18.5 us, the pipelined chain (sub-diagonal tile 7 us behind the leaf, the next diagonal
tile's last k-block 9.4 us behind that).  Used to compare ticket orders before building one:

    python tools/ct_schedule_sim.py [nct ...]          # block columns, default 41 79 128
"""
import heapq
import sys

import numpy as np

KB, K0, LEAF, STRIP, SUB_TAIL, LASTQ, STORE = 15.1, 10.0, 25.5, 18.5, 7.0, 9.4, 4.0
W = 256


def column_major(nct):
    """Today's order: whole tiles, column by column, diagonal tile first."""
    return [(i, j, 0, j, True) for j in range(nct) for i in range(j, nct)]


def chunked(nct, kmax, spread=True):
    """Tiles of column J with more than `kmax` k-blocks are cut into chunks of `kmax`; chunk c = k-blocks [c kmax, (c+1) kmax)
    can run once column (c+1) kmax - 1 is complete.  Its ticket goes among the regular tickets of the following columns
    (`spread`: dealt evenly over the kmax columns that follow, nearest columns first; otherwise in one burst)."""
    slots = [[] for _ in range(nct + 1)]
    for j in range(nct):
        n = max(1, -(-j // kmax))
        for c in range(n - 1):
            first = (c + 1) * kmax  # earliest slot: every column < first is ticketed before it
            for i in range(j, nct):
                if spread:
                    # columns first .. nct-1 hold such chunks; deal them over slots first .. min(first + kmax, j) - 1
                    span = max(1, min(kmax, nct - first))
                    s = first + min(j - first - 1, (j - first - 1) * span // max(1, nct - first - 1))
                    s = min(s, j - 1)
                else:
                    s = first
                slots[s].append((i, j, c * kmax, (c + 1) * kmax, False))
    order = []
    for j in range(nct):
        n = max(1, -(-j // kmax))
        for i in range(j, nct):
            order.append((i, j, (n - 1) * kmax if j else 0, j, True))
        order.extend(sorted(slots[j], key=lambda t: (t[1], t[0])))
    return order


def simulate(nct, order, w=W):
    pub = np.full((nct, nct), np.inf)       # publication time of the finished tile
    part = {}                               # (i, j) -> time its latest partial chunk was stored
    free = [0.0] * min(w, len(order))
    heapq.heapify(free)
    busy = 0.0
    for (i, j, k0, k1, final) in order:
        t = heapq.heappop(free)
        start = t
        if k0 > 0:
            t = max(t, part[(i, j)])
        t += K0
        for k in range(k0, k1):
            ready = max(pub[i, k], pub[j, k])
            assert np.isfinite(ready), "order is not topological"
            if final and k == j - 1 and i <= j + 1:
                t = max(t + KB, ready + LASTQ)  # the chain tiles take their last k-block quarter by quarter
            else:
                t = max(t, ready) + KB
        if not final:
            t += STORE
            part[(i, j)] = t
        elif i == j:
            t += LEAF
            pub[i, j] = t
        elif i <= j + 2:
            t = max(t + SUB_TAIL, pub[j, j] + SUB_TAIL)
            pub[i, j] = t
        else:
            t = max(t, pub[j, j]) + STRIP
            pub[i, j] = t
        busy += t - start
        heapq.heappush(free, t)
    total = max(free)
    return total, busy / (min(w, len(order)) * total)


def slanted(nct, beta):
    """Whole tiles ordered by J + beta I (ties: column, row): beta = 0 is column-major, large beta row-major.  Any beta >= 0
    is a topological order (every tile a task reads lies to the left in its own row, or in row J at or left of column J)."""
    tiles = [(j + beta * i, j, i) for j in range(nct) for i in range(j, nct)]
    tiles.sort()
    return [(i, j, 0, j, True) for _, j, i in tiles]


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [41, 79, 128]
    for nct in sizes:
        n = nct * 128
        base, occ = simulate(nct, column_major(nct))
        print(f"nct={nct} (N={n}): column-major {base:8.0f} us  ({n**3 / 3 / base / 1e6:5.1f} TF/s, workers in a task {occ:.2f})")
        for kmax in (8, 12, 16, 24, 32):
            for spread in (False, True):
                o = chunked(nct, kmax, spread)
                t, occ = simulate(nct, o)
                print(f"    chunks of {kmax:2d} {'spread' if spread else 'burst '}: {t:8.0f} us ({t / base:.3f})  tasks {len(o)}  in-task {occ:.2f}")
        print("    whole tiles ordered by J + beta I: " + "  ".join(
            f"beta {b}: {simulate(nct, slanted(nct, b))[0] / base:.3f}" for b in (0.05, 0.1, 0.2, 0.5, 1.0, 3.0, 100.0)))
