#!/bin/bash
# round 4, call 12: dRMSD pair kernel with scalar column loads + DPP phase 2: parity tests, then A/B against the previous kernel
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r4l
timeout 900 python -m pytest tests -m gpu -x -q -k "drmsd or loss or smoke" 2>&1 | grep -v amdgpu.ids | tail -15 | tee gpurun_out/r4l/tests.txt
for i in 1 2 3; do
  for tag in "" r4d; do
    PTAMD_LIB_TAG=$tag timeout 300 python profiles/tools/r03_drmsd_bench.py 2>&1 | grep "^lib" | tee -a gpurun_out/r4l/drmsd_ab.txt
  done
done
