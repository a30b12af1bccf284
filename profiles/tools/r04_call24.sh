#!/bin/bash
# round 4, call 24: staggered start of the two workgroups of a CU (128-row tiles of the LDS-DMA GEMM)
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r4t; rm -f gpurun_out/r4t/*.txt
timeout 900 python profiles/tools/r04_hp_stagger.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4t/hp_stagger.txt
