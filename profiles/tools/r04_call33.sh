#!/bin/bash
# round 4, call 33: after confining the 1-bit gate to the f16x2 no-dropout instantiation: GEMM tests, arithmetic-mode sweep
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r4t; rm -f gpurun_out/r4t/*.txt
timeout 1500 python -m pytest tests/test_gpu_gemm_hp.py tests/test_gpu_kernels.py tests/test_gpu_auto_guard.py tests/test_gpu_scales.py -m gpu -x -q -k "gemm or gate or stored_decisions or side_stream or linear" 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r4t/tests.txt
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], d['arithmetic_modes'])" | tee gpurun_out/r4t/modes.txt
