#!/bin/bash
# quick parity subset + A/B of Cholesky scheduling knobs
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
python tools/gpu_ab_env.py "$@" 2>&1 | tail -40
