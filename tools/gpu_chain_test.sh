#!/bin/bash
# fused panel-chain kernel: parity subset under a hard timeout, then A/B
export GMB_CHAIN_KERNEL=1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "factor or predict or nlml" 2>&1 | tail -5
unset GMB_CHAIN_KERNEL
timeout 600 python tools/gpu_ab_env.py default GMB_CHAIN_KERNEL=1 GMB_CHAIN_KERNEL=1,GMB_CHAIN_CUS=32 GMB_CHAIN_KERNEL=1,GMB_CHAIN_CUS=96 2>&1 | tail -14
