#!/bin/bash
# does the GEMM's HBM-side traffic (13x the algorithmic operand + C bytes) cost anything -- time or
# clock?  A/B of the shipped 128 x 128 tile (4 waves, two workgroups per compute unit) against a 256 x 128 macro-tile
# (8 waves, one workgroup per compute unit: the same two waves per SIMD, half the B-panel traffic per flop) in the
# TUNING build (GMB_GEMM_VARIANT=0 / 4): (1) wall-clock rate on plain products and on the N = 50k factorisation,
# (2) FETCH_SIZE / WRITE_SIZE and the XCD clock (GRBM_GUI_ACTIVE / 8 / duration) of the same product under rocprofv3
# --pmc, (3) the register-only MFMA ceiling before and after.
#   tools/gpu_gemm_macro_ab.sh TAG
set -u
TAG=${1:-r03x}
export TMPDIR=/tmp
ROOT=$PWD
OUT=/tmp/${TAG}_macro
mkdir -p $OUT $ROOT/gpurun_out
RES=$ROOT/gpurun_out/${TAG}_gemm_macro_ab.txt
export GUMBI_HIP_LIB=$ROOT/gumbi_amd/lib/libgumbi_hip_tuning.so
: > $RES
python - >> $RES 2>&1 <<PY
import sys; sys.path.insert(0, "$ROOT")
from gumbi_amd import engine
print("MFMA ceiling before:", engine.mfma_f64_sustained(0, 1.0))
PY
if [ "${PMC_ONLY:-0}" != "1" ]; then
echo "== plain products (gmb_blk_gemm_nt), variants 0 = 128x128, 4 = 256x128" >> $RES
SWEEP_VARIANTS=0,4,0,4 timeout 600 python tools/gpu_gemm_sweep.py 8192,8192,8192 16384,16384,3072 24576,24576,1024 6144,44032,6144 >> $RES 2>&1
echo "== N = 50k factorisation" >> $RES
timeout 600 python tools/gpu_ab_big.py 50000 GMB_GEMM_VARIANT=0 GMB_GEMM_VARIANT=4 GMB_GEMM_VARIANT=0 GMB_GEMM_VARIANT=4 >> $RES 2>&1
fi
cat > $OUT/run.py <<PY
import sys; sys.path.insert(0, "$ROOT")
import torch
from gumbi_amd.engine import Engine
eng = Engine(0)
dev = torch.device("cuda:0")
m, n, k = 16384, 16384, 3072
A = torch.randn(k, m, dtype=torch.float64, device=dev)
B = torch.randn(k, n, dtype=torch.float64, device=dev)
Cm = torch.zeros(m, n, dtype=torch.float64, device=dev)
for rep in range(6):
    eng.blk_gemm_nt(Cm.data_ptr(), n, A.data_ptr(), m, B.data_ptr(), n, m, n, k, -1.0, 1.0)
torch.cuda.synchronize()
PY
cd /tmp
for V in 0 4; do  # one counter group per pass (the guide's HBM recipe: FETCH_SIZE and WRITE_SIZE in passes of their own)
  export GMB_GEMM_VARIANT=$V
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/v$V/a -o p -- python $OUT/run.py > $OUT/v${V}a.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/v$V/b -o p -- python $OUT/run.py > $OUT/v${V}b.log 2>&1
  timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/v$V/c -o p -- python $OUT/run.py > $OUT/v${V}c.log 2>&1
done
cd $ROOT
python - >> $RES 2>&1 <<PY
import csv, glob
from collections import defaultdict
print("== rocprofv3 --pmc on 16384 x 16384 x 3072 (6 launches each; FETCH_SIZE doubled per MI355X_MICROARCH.md)")
for V in (0, 4):
    c = defaultdict(float); dur = 0.0; n = 0
    for sub in "abc":
        seen = set()
        for f in glob.glob("$OUT/v%d/%s/**/*counter_collection.csv" % (V, sub), recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm_f64" not in r["Kernel_Name"]: continue
                c[r["Counter_Name"]] += float(r["Counter_Value"])
                if sub == "c" and r["Dispatch_Id"] not in seen:  # durations of the clock / MFMA pass
                    seen.add(r["Dispatch_Id"]); dur += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); n += 1
    if not n:
        print("variant", V, "no data"); continue
    flops = 2.0 * 16384 * 16384 * 3072 * n
    fetch = 2.0 * c["FETCH_SIZE"] * 1024; write = c["WRITE_SIZE"] * 1024
    print("variant %d: %d launches, %.3f ms each (under PMC) = %.1f TF/s; fetch %.2f GB + write %.2f GB per launch = %.1f flop/B; "
          "XCD clock %.0f MHz; MFMA pipe busy %.1f %%" % (V, n, dur / n / 1e6, flops / dur / 1e3, fetch / n / 1e9, write / n / 1e9,
          flops / (fetch + write), c["GRBM_GUI_ACTIVE"] / 8 / (dur / 1e3), 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)))
PY
python - >> $RES 2>&1 <<PY
import sys; sys.path.insert(0, "$ROOT")
from gumbi_amd import engine
print("MFMA ceiling after:", engine.mfma_f64_sustained(0, 1.0))
PY
tail -3 $OUT/v0a.log >> $RES 2>&1
cat $RES
