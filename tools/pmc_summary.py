"""Per-kernel PMC summary from rocprofv3 counter_collection CSVs (one pass per counter group).

    python tools/pmc_summary.py gpurun_out/pmc profiles/r01_pmc_summary.csv

MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs): rocprofv3 reports
GRBM_GUI_ACTIVE summed over the 8 XCDs, so the per-SIMD busy fraction needs the per-XCD cycle count (the
round-1 summaries divided by the sum and read 8x low: the register-only MFMA microbenchmark showed 12.25 %
at 71.8 TF/s; with this normalisation it reads ~98 % of the cycles at the clock the chip actually ran);
achieved f64 MFMA rate from SQ_INSTS_VALU_MFMA_MOPS_F64 * 512 / duration.  FETCH_SIZE is doubled
per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B for wide coalesced streams);
WRITE_SIZE is reported as counted (uncalibrated).
"""
import csv
import sys
from collections import defaultdict
from pathlib import Path


def load(path):
    per = defaultdict(lambda: defaultdict(float))
    dur = defaultdict(float)
    n = defaultdict(int)
    seen = set()
    with open(path) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            per[k][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (row["Dispatch_Id"], k)
            if key not in seen:
                seen.add(key)
                dur[k] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
                n[k] += 1
    return per, dur, n


def main(root, out):
    root = Path(root)
    p1, d1, n1 = load(next((root / "p1").glob("*counter_collection.csv")))
    p2, d2, _ = load(next((root / "p2").glob("*counter_collection.csv")))
    p3, d3, _ = load(next((root / "p3").glob("*counter_collection.csv")))
    rows = []
    for k in sorted(d1, key=lambda kk: -d1[kk]):
        c = p1[k]
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        mfma_busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)
        util = 100.0 * mfma_busy / (gui / 8.0 * 1024.0) if gui else 0.0
        tf = mops * 512.0 / (d1[k] * 1e-9) / 1e12 if d1[k] else 0.0
        fetch_gb = 2.0 * p2.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0 / 1e9
        write_gb = p3.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0 / 1e9
        rows.append([k, n1[k], round(d1[k] / 1e6, 3), round(d1[k] / 1e6 / max(n1[k], 1), 5), round(mops * 512.0 / max(n1[k], 1) / 1e9, 3), round(util, 2), round(tf, 2),
                     round(fetch_gb, 4), round(fetch_gb / (d2.get(k, 0) * 1e-9 + 1e-30) if d2.get(k) else 0, 1),
                     round(write_gb, 4), round(write_gb / (d3.get(k, 0) * 1e-9 + 1e-30) if d3.get(k) else 0, 1)])
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kernel", "Launches", "TotalMs(pass1)", "AvgMs(pass1)", "MFMA_GFlopPerLaunch", "MfmaUtil%", "MFMA_F64_TFLOPs", "FetchGB(x2 corrected)",
                    "FetchGB/s", "WriteGB(raw)", "WriteGB/s"])
        w.writerows(rows)
    for r in rows[:8]:
        print(r)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
