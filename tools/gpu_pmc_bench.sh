#!/bin/bash
# Three separate rocprofv3 --pmc passes (SQ+GRBM, FETCH_SIZE, WRITE_SIZE) over ONE bench step of
# the C2 workload; counter CSVs under gpurun_out/<TAG>_pmc/p{1,2,3}.  No tracing domains.
set -u
TAG=${1:-r01x}
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/${TAG}_pmc
mkdir -p $OUT
cd /tmp
CMD="python $ROOT/bench.py --steps 1 --warmup 0"
GUMBI_BENCH_NO_DIST=1 GUMBI_BENCH_NO_CPU=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -o p1 -- $CMD > $OUT/p1.log 2>&1
GUMBI_BENCH_NO_DIST=1 GUMBI_BENCH_NO_CPU=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p2 -o p2 -- $CMD > $OUT/p2.log 2>&1
GUMBI_BENCH_NO_DIST=1 GUMBI_BENCH_NO_CPU=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p3 -o p3 -- $CMD > $OUT/p3.log 2>&1
cd $ROOT
find $OUT -name "*counter_collection.csv" | head
# keep only the summary (the raw CSVs are large)
python tools/pmc_summary.py $OUT gpurun_out/${TAG}_pmc_bench_c2_summary.csv
find $OUT -name "*.csv" -size +8M -delete
