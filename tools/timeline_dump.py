"""Timeline of the LAST factorisation (or whatever followed the last covariance build) in a
rocprofv3 --kernel-trace sqlite file: per kernel start offset, duration, queue, grid; plus the
busy / idle split of the critical (leaf) chain.

    python tools/timeline_dump.py gpurun_out/tl/x_results.db [max_rows]
"""
import sqlite3
import sys


def short(name):
    name = name.replace("gmb::", "").replace("void ", "")
    return name.split("(")[0][:34]


def main(db, max_rows=400):
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end, queue_id, stream_id, grid_x, workgroup_x from kernels order by start").fetchall()
    # last cov_tile launch marks the start of the last factorisation; with a start marker in
    # argv[3] (substring of a kernel name) the dump starts at that kernel's last-but-N occurrence
    marker = sys.argv[3] if len(sys.argv) > 3 else "cov_tile_kernel"
    last = max(i for i, r in enumerate(rows) if marker in r[0])
    rows = rows[last:]
    t0 = rows[0][1]
    tend = max(r[2] for r in rows)
    print(f"{len(rows)} launches, span {(tend - t0) / 1e3:.1f} us")
    agg = {}
    for r in rows:
        a = agg.setdefault((short(r[0]), r[3]), [0, 0.0])
        a[0] += 1
        a[1] += (r[2] - r[1]) / 1e3
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {k[0]:36s} q{k[1]}  n={v[0]:5d}  {v[1]:10.1f} us")
    # the leaf chain: time from each leaf end to the next leaf start
    leaves = [r for r in rows if "potrf_leaf" in r[0]]
    if len(leaves) > 1:
        dur = [(r[2] - r[1]) / 1e3 for r in leaves]
        gap = [(b[1] - a[2]) / 1e3 for a, b in zip(leaves[:-1], leaves[1:])]
        print(f"leaf: n={len(leaves)} mean dur {sum(dur)/len(dur):.1f} us (min {min(dur):.1f}, max {max(dur):.1f}); "
              f"mean gap to next leaf {sum(gap)/len(gap):.1f} us (min {min(gap):.1f}, max {max(gap):.1f})")
        print("leaf dur :", " ".join(f"{x:.0f}" for x in dur))
        print("leaf gaps:", " ".join(f"{x:.0f}" for x in gap))
    for r in rows[:max_rows]:
        print(f"{(r[1] - t0) / 1e3:10.1f} +{(r[2] - r[1]) / 1e3:8.1f}  q{r[3]} s{r[4]}  grid {r[5] // max(r[6], 1):6d}  {short(r[0])}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 400)
