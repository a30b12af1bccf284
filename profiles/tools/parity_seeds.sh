#!/bin/bash
# the committed parity record: N independent draws (model initialisation + batch) per BASELINE configuration, arithmetic and
# regime (arbitrary / realistic angles) -> profiles/<round>/<round>_parity.json (per-draw records + median / max over the draws,
# skip rates, pass / fail counts against SURVEY 8(d) as written and against the relaxed bar)
#   usage: parity_seeds.sh <round> [arbitrary draws = 8] [realistic draws = 4]
round=${1:-r06}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/parity8
rm -f gpurun_out/parity8/${round}_parity.json
PTAMD_PARITY_PROBES=12 PTAMD_PARITY_SEEDS=${2:-8} PTAMD_PARITY_SEEDS_REALISTIC=${3:-4} PTAMD_PARITY_OUT=$PWD/gpurun_out/parity8/${round}_parity.json timeout 3400 python -m pytest tests/test_gpu_parity_record.py -q -m gpu 2>&1 | tail -n 15 | tee gpurun_out/parity8/log.txt
