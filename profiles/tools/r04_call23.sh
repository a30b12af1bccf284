#!/bin/bash
# round 4, call 23: ping-pong variant of the LDS-DMA GEMM: tests under every PM, per-product A/B, step A/B
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r4t; rm -f gpurun_out/r4t/*.txt
for pm in 4 6; do
  PTAMD_HP_PP=$pm timeout 900 python -m pytest tests/test_gpu_gemm_hp.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3 | sed "s/^/PM=$pm: /" | tee -a gpurun_out/r4t/tests.txt
done
timeout 600 python profiles/tools/r04_hp_pp.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4t/hp_pp.txt
for i in 1 2 3; do
  for pm in "" 4 6; do
    PTAMD_HP_PP=$pm timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-mode-sweep 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PP=$pm', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r4t/step_ab.txt
  done
done
