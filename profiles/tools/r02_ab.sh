#!/bin/bash
# A/B of two bench.py flag sets on one box, alternating:  r02_ab.sh "<flags A>" "<flags B>" [rounds]
for i in $(seq 1 ${3:-2}); do
  for f in "$1" "$2"; do
    python bench.py --no-cpu-baseline --no-mode-sweep --no-kernel-timing $f 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$f]', d['ms_per_step'], d['value'])"
  done
done
