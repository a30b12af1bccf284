#!/bin/bash
# round 4, call 28: step A/B of the stored attention dropout decisions, with and without the side stream
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r4t; rm -f gpurun_out/r4t/step_ab2.txt
for i in 1 2; do
  for fl in "" "--no-attn-keep-bits"; do
    timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-mode-sweep --no-kernel-timing $fl 2>/dev/null | python profiles/tools/benchline.py "flags [$fl]" attn_keep_bits_layer_passes | tee -a gpurun_out/r4t/step_ab2.txt
  done
done
