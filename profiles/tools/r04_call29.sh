#!/bin/bash
# round 4, call 29: the 1-bit gate of the FFN hidden layer: tests, product A/B, step A/B
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r4t; rm -f gpurun_out/r4t/*.txt
timeout 1500 python -m pytest tests/test_gpu_gemm_hp.py tests/test_gpu_kernels.py tests/test_gpu_auto_guard.py -m gpu -x -q -k "gemm or gate or stored_decisions or side_stream" 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/r4t/tests.txt
timeout 300 python profiles/tools/r04_gate_mask_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4t/gate_mask.txt
for i in 1 2 3; do
  for fl in "" "--no-ffn-gate-mask"; do
    timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-mode-sweep --no-kernel-timing $fl 2>/dev/null | python profiles/tools/benchline.py "flags [$fl]" attn_keep_bits_layer_passes ffn_gate_mask_layer_passes | tee -a gpurun_out/r4t/step_ab3.txt
  done
done
