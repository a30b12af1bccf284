#!/bin/bash
# round 4, call 15: weight-scale kernel at 1024 threads (tests), attention forward compiled for 4 wavefronts per SIMD (A/B), step time
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r4o; rm -f gpurun_out/r4o/*.txt
timeout 1200 python -m pytest tests -m gpu -x -q -k "scales or attention or guard_measures or drmsd" 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r4o/tests.txt
for i in 1 2 3; do
  for tag in "" w4; do
    PTAMD_LIB_TAG=$tag timeout 300 python profiles/tools/r04_attn_bench.py 2>&1 | grep "^lib" | tee -a gpurun_out/r4o/attn_ab.txt
  done
done
for i in 1 2; do
  for tag in "" w4; do
    PTAMD_LIB_TAG=$tag timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-mode-sweep 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib $tag', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r4o/step_ab.txt
  done
done
