#!/bin/bash
# kernel-time profile of the small workloads (4 proteins x 512 at d512; config 2; config 1): where do ~190 launches go
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4s
for w in "--batch 4" "--config 2" "--config 1"; do
  tag=$(echo $w | tr -d ' -')
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4s/$tag -o s -- python bench.py $w --steps 20 --warmup 5 --no-cpu-baseline --no-mode-sweep --no-kernel-timing > gpurun_out/r4s/$tag.json 2> gpurun_out/r4s/$tag.err
  rm -f gpurun_out/r4s/$tag/*/s_kernel_trace.csv gpurun_out/r4s/$tag/s_kernel_trace.csv
  python -c "import json; d=json.load(open('gpurun_out/r4s/$tag.json')); print('$w', d['ms_per_step'])"
  python profiles/summarize.py stats $(find gpurun_out/r4s/$tag -name "*kernel_stats.csv" | head -1) 46 | head -32
done
