#!/bin/bash
# Round bench + rocprofv3 kernel stats of the same command (run on the GPU box through gpurun).
set -u
TAG=${1:-r01x}
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench_c2.json 2> gpurun_out/${TAG}_bench_c2.err
cat gpurun_out/${TAG}_bench_c2.json
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
GUMBI_BENCH_NO_DIST=1 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/${TAG}_prof -o ${TAG} -- python $ROOT/bench.py --steps 1 --warmup 1 > $ROOT/gpurun_out/${TAG}_prof_bench.log 2>&1
cd $ROOT
DB=$(find gpurun_out/${TAG}_prof -name "*_results.db" | head -1)
python tools/rocprof_summary.py $DB gpurun_out/${TAG}_bench_c2_kernel_stats.csv
head -12 gpurun_out/${TAG}_bench_c2_kernel_stats.csv
tail -3 gpurun_out/${TAG}_prof_bench.log
