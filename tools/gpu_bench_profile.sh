#!/bin/bash
# Round bench + rocprofv3 kernel stats of the same command (run on the GPU box through gpurun).
#   tools/gpu_bench_profile.sh TAG [CONFIG] [STEPS] [WARMUP]      (CONFIG default c3)
set -u
TAG=${1:-r02x}
CFG=${2:-c3}
STEPS=${3:-3}
WARM=${4:-1}
mkdir -p gpurun_out
python bench.py --config $CFG --steps $STEPS --warmup $WARM > gpurun_out/${TAG}_bench_${CFG}.json 2> gpurun_out/${TAG}_bench_${CFG}.err
cat gpurun_out/${TAG}_bench_${CFG}.json
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
GUMBI_BENCH_NO_DIST=1 GUMBI_BENCH_NO_E2E=1 GUMBI_BENCH_NO_CPU=1 GUMBI_BENCH_NO_C2=1 GUMBI_BENCH_NO_C4=1 GUMBI_BENCH_NO_DEFAULT_START=1 timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/${TAG}_prof -o ${TAG} -- python $ROOT/bench.py --config $CFG --steps 1 --warmup 0 > $ROOT/gpurun_out/${TAG}_prof_bench.log 2>&1
cd $ROOT
DB=$(find gpurun_out/${TAG}_prof -name "*_results.db" | head -1)
python tools/rocprof_summary.py $DB gpurun_out/${TAG}_bench_${CFG}_kernel_stats.csv
head -14 gpurun_out/${TAG}_bench_${CFG}_kernel_stats.csv
tail -3 gpurun_out/${TAG}_prof_bench.log
rm -rf gpurun_out/${TAG}_prof
