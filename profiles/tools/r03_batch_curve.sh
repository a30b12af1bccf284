#!/bin/bash
# step time against the batch (and the two small configurations), AUTO arithmetic, one box
for b in 4 8 16 32; do python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-mode-sweep --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $b', d['ms_per_step'], d['value'])"; done
for c in 1 2; do for m in auto f32; do python bench.py --config $c --gemm-mode $m --steps 20 --warmup 3 --no-cpu-baseline --no-mode-sweep --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config $c mode $m', d['ms_per_step'])"; done; done
