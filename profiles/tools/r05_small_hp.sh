#!/bin/bash
# Small batches at d512 (the per-GPU shares of a strong-scaling run): the products behind the LayerNorms on the staging kernel
# (the default below 4096 tokens) against the LDS-DMA kernel with 256- and 128-row tiles.  One box, alternating.
out=gpurun_out/r05_small_hp.txt; : > $out
B="python bench.py --steps 40 --warmup 10 --passes 3 --no-cpu-baseline --no-mode-sweep --no-strong --no-kernel-timing"
line() { python -c "import json,sys; e=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', e['ms_per_step'], e['passes']['ms_per_step'], e['resident']['ms_per_step'])"; }
for rep in 1 2; do
for b in 4 8 16; do
  $B --batch $b 2>/dev/null | line "b=$b default" >> $out
  PTAMD_HP_MIN_TOKENS=2048 PTAMD_HP_TILE=128 $B --batch $b 2>/dev/null | line "b=$b hp,128-row" >> $out
  [ $b = 4 ] && PTAMD_HP_MIN_TOKENS=2048 $B --batch $b 2>/dev/null | line "b=$b hp,256-row" >> $out
done; done
cat $out
