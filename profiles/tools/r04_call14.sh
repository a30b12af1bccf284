#!/bin/bash
# round 4, call 14: dRMSD pair kernel: work items of at most 16 / 8 / 4 column tiles; per-kernel times of the loss
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r4n; rm -f gpurun_out/r4n/drmsd_ab.txt
for i in 1 2 3; do
  for tag in "" c8 c4; do
    PTAMD_LIB_TAG=$tag timeout 300 python profiles/tools/r03_drmsd_bench.py 2>&1 | grep "^lib" | tee -a gpurun_out/r4n/drmsd_ab.txt
  done
done
for tag in "" c8; do
  (cd /tmp && PTAMD_LIB_TAG=$tag timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_d$tag -o d -- python /root/repo/profiles/tools/r03_drmsd_bench.py > /dev/null 2>&1)
  f=$(find /tmp/prof_d$tag -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r4n/drmsd_kernel_stats_${tag:-product}.csv
  head -8 "$f" | cut -c1-150
done
