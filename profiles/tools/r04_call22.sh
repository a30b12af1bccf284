#!/bin/bash
# round 4, call 22: attention forward with the V fragments requested in front of the soft-max: tests, kernel A/B, step A/B
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r4t; rm -f gpurun_out/r4t/*.txt
timeout 1200 python -m pytest tests -m gpu -x -q -k "attention or attn or kernels" 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r4t/tests.txt
for i in 1 2 3; do
  for tag in "" av0; do
    PTAMD_LIB_TAG=$tag timeout 300 python profiles/tools/r04_attn_bench.py 2>&1 | grep "^lib" | tee -a gpurun_out/r4t/attn_ab.txt
  done
done
for i in 1 2; do
  for tag in "" av0; do
    PTAMD_LIB_TAG=$tag timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-mode-sweep 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib $tag', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r4t/step_ab.txt
  done
done
