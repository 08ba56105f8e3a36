#!/bin/bash
# usage: gpu_timeline.sh TAG N d what   -> gpurun_out/TAG_tl/*_results.db
set -u
TAG=$1; N=${2:-10000}; D=${3:-4}; WHAT=${4:-factorize}
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/${TAG}_tl -o ${TAG} -- python $ROOT/tools/gpu_timeline_run.py $N $D $WHAT > $ROOT/gpurun_out/${TAG}_tl.log 2>&1
cd $ROOT
DB=$(find gpurun_out/${TAG}_tl -name "*_results.db" | head -1)
python tools/timeline_dump.py $DB 0 | head -40
