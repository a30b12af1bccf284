#!/bin/bash
# round 4, call 26: attention dropout decisions handed from the forward to the fused backward kernel (v_writelane through the intrinsic)
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r4t; rm -f gpurun_out/r4t/*.txt
timeout 300 python profiles/tools/r04_attn_bits_debug.py 2>&1 | grep -v amdgpu.ids | grep "words differing" | tee gpurun_out/r4t/bits_debug.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "attention or attn or keep_bits or test_abi" 2>&1 | grep -v amdgpu.ids | tail -6 | tee gpurun_out/r4t/tests.txt
timeout 300 python profiles/tools/r04_attn_bits_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4t/attn_bits.txt
for i in 1 2 3; do
  for fl in "" "--no-attn-keep-bits"; do
    timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-mode-sweep $fl 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags [$fl]', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r4t/step_ab.txt
  done
done
