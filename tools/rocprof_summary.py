"""Turn a rocprofv3 ``*_results.db`` (rocpd sqlite, what ``--kernel-trace --stats`` writes on
ROCm 7.2) into the per-kernel summary CSV committed under ``profiles/``.

    python tools/rocprof_summary.py gpurun_out/prof_r1/bench_results.db profiles/r01_bench_c2_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db, out):
    con = sqlite3.connect(db)
    rows = con.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name order by sum(duration) desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, calls, tot, avg, mn, mx in rows:
            w.writerow([name, calls, int(tot), round(avg, 1), round(100.0 * tot / total, 3), int(mn), int(mx)])
    print(f"{len(rows)} kernels, {total/1e6:.1f} ms of GPU time -> {out}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
