#!/bin/bash
# round 4, call 31: dX of FFN layer 2 on the LDS-DMA kernel now that its gate is one bit per element
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r4t; rm -f gpurun_out/r4t/step_ab4.txt
for i in 1 2 3; do
  for fl in "" "--hp-dx" "--no-ffn-gate-mask" "--no-ffn-gate-mask --hp-dx"; do
    timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-mode-sweep --no-kernel-timing $fl 2>/dev/null | python profiles/tools/benchline.py "flags [$fl]" | tee -a gpurun_out/r4t/step_ab4.txt
  done
done
