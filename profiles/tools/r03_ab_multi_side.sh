#!/bin/bash
# weight-gradient products of a small-batch step: one launch of 256 work items each on the main / one side stream (default)
# against <= 128 / 64 work items each on 4 side streams running side by side
run() { python bench.py --batch $1 --steps 20 --warmup 3 --no-cpu-baseline --no-mode-sweep --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$2 batch $1', d['ms_per_step'])"; }
for rep in 1 2; do for b in 4 8 16; do
  run $b "default         "
  PTAMD_SIDE_MIN_TOKENS=2048 PTAMD_SIDE_FEW_TOKENS=100000 PTAMD_SIDE_STREAMS=4 PTAMD_DW_SLOTS_FEW=128 run $b "4 streams, 128 slots"
  PTAMD_SIDE_MIN_TOKENS=2048 PTAMD_SIDE_FEW_TOKENS=100000 PTAMD_SIDE_STREAMS=4 PTAMD_DW_SLOTS_FEW=256 run $b "4 streams, 256 slots"
  PTAMD_SIDE_MIN_TOKENS=2048 PTAMD_SIDE_FEW_TOKENS=100000 PTAMD_SIDE_STREAMS=2 PTAMD_DW_SLOTS_FEW=128 run $b "2 streams, 128 slots"
done; done
