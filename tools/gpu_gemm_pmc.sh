#!/bin/bash
# Where do the wave cycles of the 128 x 128 MFMA GEMM go?  Two rocprofv3 --pmc passes (SQ wait / active
# breakdown; LDS) over a plain m x n x k product through gmb_blk_gemm_nt.
#   tools/gpu_gemm_pmc.sh TAG [m,n,k]
set -u
TAG=${1:-r02x}
SHAPE=${2:-8192,8192,8192}
export TMPDIR=/tmp
ROOT=$PWD
OUT=/tmp/${TAG}_gemm_pmc
mkdir -p $OUT $ROOT/gpurun_out
cat > $OUT/run.py <<PY
import sys; sys.path.insert(0, "$ROOT")
import torch
from gumbi_amd.engine import Engine
eng = Engine(0)
dev = torch.device("cuda:0")
m, n, k = (int(v) for v in "$SHAPE".split(","))
A = torch.randn(k, m, dtype=torch.float64, device=dev)
B = torch.randn(k, n, dtype=torch.float64, device=dev)
Cm = torch.zeros(m, n, dtype=torch.float64, device=dev)
for rep in range(4):
    eng.blk_gemm_nt(Cm.data_ptr(), n, A.data_ptr(), m, B.data_ptr(), n, m, n, k, -1.0, 1.0)
torch.cuda.synchronize()
PY
cd /tmp
export GMB_GEMM_VARIANT=${GMB_GEMM_VARIANT:-0}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -o p1 -- python $OUT/run.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/p2 -o p2 -- python $OUT/run.py > $OUT/p2.log 2>&1
cd $ROOT
python - <<PY
import csv, glob
from collections import defaultdict
rows = defaultdict(lambda: defaultdict(float)); dur = defaultdict(float); n = defaultdict(int); seen = set()
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_f64" not in k: continue
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); dur[(f, k)] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); n[(f, k)] += 1
with open("gpurun_out/${TAG}_gemm_pmc.txt", "w") as out:
    for k, c in rows.items():
        lines = [k] + [f"  {name:28s} {val:.4g}" for name, val in sorted(c.items())]
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1
        lines.append("  fractions of SQ_WAVE_CYCLES: wait_any %.3f  wait_inst_any %.3f  (of it LDS %.3f)  active_inst_any %.3f" % (
            c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_WAIT_INST_LDS", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc))
        gui = c.get("GRBM_GUI_ACTIVE", 0)
        if gui: lines.append("  MFMA pipe busy %.1f %% of XCD-active cycles" % (100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui / 8 * 1024)))
        for (f, kk), d in dur.items():
            if kk == k: lines.append("  %d launches, %.3f ms each (%s)" % (n[(f, kk)], d / n[(f, kk)] / 1e6, f.split("/")[-3] if "/" in f else f))
        print("\n".join(lines)); out.write("\n".join(lines) + "\n")
PY
tail -2 $OUT/p1.log
