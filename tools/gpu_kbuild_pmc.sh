#!/bin/bash
# PMC evidence for the covariance build's bound: separate --pmc passes over tools/gpu_kbuild_pmc.py.
#   tools/gpu_kbuild_pmc.sh [N:d:kind]
export KB_CASE=${1:-50000:8:Matern52}
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "WRITE_SIZE" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  tools/gpu_pmc_script.sh kb "$grp" tools/gpu_kbuild_pmc.py cov_tile 2>&1 | grep "^kb"
done
