#!/bin/bash
# PMC evidence for the trace reductions' bound (grad_interior_kernel): separate --pmc passes over a few
# evaluations at C3's size (tools/gpu_eval_pmc.py with EP_N=50000 EP_D=8 EP_KIND=Matern52).
export EP_N=${1:-50000} EP_D=${2:-8} EP_KIND=${3:-Matern52} EP_REPS=2
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM" \
           "FETCH_SIZE"; do
  tools/gpu_pmc_script.sh gr "$grp" tools/gpu_eval_pmc.py grad_interior 2>&1 | grep "^gr"
done
