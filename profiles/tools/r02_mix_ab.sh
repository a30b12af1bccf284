#!/bin/bash
# A/B on one box (rebuilds there): f16 splitting with v_fma_mixlo/mixhi_f16 (-DPT_MIX_SPLIT, 8 instructions per 4 values)
# against the compiled 12-instruction form
run() { python bench.py --no-cpu-baseline --no-mode-sweep --no-kernel-timing 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'])"; }
run compiled; run compiled
PTAMD_EXTRA_FLAGS=-DPT_MIX_SPLIT python -m protein_transformer_amd.build > /dev/null 2>&1
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm or attention" -x 2>&1 | tail -1
run mix; run mix
python -m protein_transformer_amd.build > /dev/null 2>&1
run compiled
