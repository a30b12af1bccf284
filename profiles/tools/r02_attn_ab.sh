#!/bin/bash
# A/B on one box: whole step with the attention kernels in bf16x3 (three-term) and in f16x2 (two-term) arithmetic, GEMMs in AUTO
mkdir -p gpurun_out
for m in bf16x3 f16x2 bf16x3 f16x2; do
  python bench.py --no-cpu-baseline --no-mode-sweep --no-kernel-timing --attn-mode $m 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['ms_per_step'], d['value'])"
done
