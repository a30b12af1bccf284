#!/bin/bash
# round 4, call 13/14: dRMSD pair kernel (LDS columns, DPP row broadcasts, no clamp / select per pair), 8 or 4 chains, against the previous kernel
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r4m; rm -f gpurun_out/r4m/drmsd_ab.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "drmsd or loss" 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r4m/tests.txt
for i in 1 2 3; do
  for tag in "" l4 r4d; do
    PTAMD_LIB_TAG=$tag timeout 300 python profiles/tools/r03_drmsd_bench.py 2>&1 | grep "^lib" | tee -a gpurun_out/r4m/drmsd_ab.txt
  done
done
