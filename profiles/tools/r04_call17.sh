#!/bin/bash
# round 4, call 17: the weight-gradient products of a layer as a group: tests, then the step A/B (same box, alternating)
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r4q; rm -f gpurun_out/r4q/*.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "group or scales or gemm or model or train or dp or side_stream" 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/r4q/tests.txt
for i in 1 2 3; do
  for flag in "" "--no-group-dw"; do
    timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-mode-sweep $flag 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags [$flag]', d['ms_per_step'], d['value'], d['roofline']['frac'])" | tee -a gpurun_out/r4q/step_ab.txt
  done
done
for cfg in 1 2 3 5; do
  for flag in "" "--no-group-dw"; do
    timeout 600 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-mode-sweep $flag 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $cfg flags [$flag]', d['ms_per_step'])" | tee -a gpurun_out/r4q/step_ab.txt
  done
done
