#!/bin/bash
# One rocprofv3 --pmc pass (no tracing domains) over a small python script; prints per-kernel sums of the counters.
#   tools/gpu_pmc_script.sh TAG "COUNTER1 COUNTER2 ..." script.py [kernel-name filter]      (environment passes through)
set -u
TAG=$1
CTRS=$2
SCRIPT=$3
FILT=${4:-}
export TMPDIR=/tmp
ROOT=$PWD
OUT=/tmp/${TAG}_pmc
mkdir -p $OUT $ROOT/gpurun_out
cd /tmp
timeout 600 rocprofv3 --pmc $CTRS --output-format csv -d $OUT -o p -- python $ROOT/$SCRIPT > $OUT/run.log 2>&1
cd $ROOT
python - <<PY
import csv, glob, collections
files = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.Counter(); seen = set()
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "$FILT" and "$FILT" not in k: continue
        per[k][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (row["Dispatch_Id"], k)
        if key not in seen:
            seen.add(key); n[k] += 1; dur[k] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
for k in sorted(dur, key=lambda kk: -dur[kk])[:6]:
    print("$TAG", k[:60], "launches", n[k], "ms/launch %.3f" % (dur[k] / n[k] / 1e6), {c: "%.4g" % (v / n[k]) for c, v in per[k].items()})
PY
tail -1 $OUT/run.log
