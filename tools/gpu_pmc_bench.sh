#!/bin/bash
# Three separate rocprofv3 --pmc passes (SQ+GRBM, FETCH_SIZE, WRITE_SIZE) over ONE bench step of the chosen
# workload; no tracing domains.  Only the per-kernel summary is kept (the raw counter CSVs are large).
#   tools/gpu_pmc_bench.sh TAG [CONFIG]       (CONFIG default c3)
set -u
TAG=${1:-r02x}
CFG=${2:-c3}
export TMPDIR=/tmp
ROOT=$PWD
OUT=/tmp/${TAG}_pmc
mkdir -p $OUT $ROOT/gpurun_out
cd /tmp
# one step = a capped fit (8 evaluations: the counters are per launch, the launches of every evaluation are alike)
CMD="python $ROOT/bench.py --config $CFG --steps 1 --warmup 0 --map-evals 8"
export GUMBI_BENCH_NO_DIST=1 GUMBI_BENCH_NO_CPU=1 GUMBI_BENCH_NO_E2E=1 GUMBI_BENCH_NO_DEFAULT_START=1 GUMBI_BENCH_NO_C2=1 GUMBI_BENCH_NO_C4=1
timeout 1200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -o p1 -- $CMD > $OUT/p1.log 2>&1
timeout 1200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p2 -o p2 -- $CMD > $OUT/p2.log 2>&1
timeout 1200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p3 -o p3 -- $CMD > $OUT/p3.log 2>&1
cd $ROOT
python tools/pmc_summary.py $OUT gpurun_out/${TAG}_pmc_bench_${CFG}_summary.csv
# which kernel sources the counters were taken on (bench.py marks roofline.traffic stale when they differ from the tree's)
python - <<EOF
import json, sys
sys.path.insert(0, "$ROOT")
import bench
json.dump({"kernel_sources_sha16": bench.kernel_sources_sha16(), "command": "$CMD", "tag": "$TAG"},
          open("$ROOT/gpurun_out/${TAG}_pmc_bench_${CFG}_summary.meta.json", "w"))
EOF
tail -2 $OUT/p1.log
