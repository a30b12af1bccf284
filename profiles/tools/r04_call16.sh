#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r4p
timeout 600 python profiles/tools/r04_dw_group_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4p/dw_group_probe.txt
