#!/bin/bash
# rocprofv3 kernel stats of a small python script (run on the GPU box through gpurun); bounded: rocprofv3 can hang in its
# exit handler on this image after writing its output.
#   tools/gpu_prof_script.sh TAG script.py      (environment variables pass through)
set -u
TAG=$1
SCRIPT=$2
export TMPDIR=/tmp
ROOT=$PWD
mkdir -p $ROOT/gpurun_out
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_prof -o ${TAG} -- python $ROOT/$SCRIPT > $ROOT/gpurun_out/${TAG}_prof.log 2>&1
cd $ROOT
DB=$(find /tmp/${TAG}_prof -name "*_results.db" | head -1)
python tools/rocprof_summary.py $DB gpurun_out/${TAG}_kernel_stats.csv
head -${3:-12} gpurun_out/${TAG}_kernel_stats.csv | cut -c1-220
