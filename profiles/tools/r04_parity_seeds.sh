#!/bin/bash
# the committed parity record: 8 independent draws (model initialisation + batch) per BASELINE configuration and arithmetic
# -> profiles/r04/r04_parity.json (per-draw records + median / max over the draws + skip rate of ill-conditioned draws)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/parity8
rm -f gpurun_out/parity8/r04_parity.json
PTAMD_PARITY_SEEDS=${1:-8} PTAMD_PARITY_OUT=$PWD/gpurun_out/parity8/r04_parity.json timeout 3000 python -m pytest tests/test_gpu_parity_record.py -q -m gpu 2>&1 | tail -n 15 | tee gpurun_out/parity8/log.txt
