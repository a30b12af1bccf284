#!/bin/bash
# round 4, call 27: in-step kernel durations with / without the stored attention dropout decisions
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4k; rm -rf gpurun_out/r4k/*
for tag in bits nobits; do
  fl=""; [ $tag = nobits ] && fl="--no-attn-keep-bits"
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4k/$tag -o s -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-mode-sweep --no-kernel-timing $fl > gpurun_out/r4k/$tag.json 2> gpurun_out/r4k/$tag.err
  find gpurun_out/r4k/$tag -name "*kernel_trace.csv" -delete
  python -c "import json; d=json.loads(open('gpurun_out/r4k/$tag.json').read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'])"
  python profiles/summarize.py stats $(find gpurun_out/r4k/$tag -name "*kernel_stats.csv" | head -1) 13 | head -14
done
