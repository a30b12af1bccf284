#!/bin/bash
# Round 6: per-step kernel tables of the per-GPU shares of a strongly scaled job (4 / 8 / 16 proteins x 512).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-r06_small_profile}; mkdir -p $out
for b in ${BATCHES:-4 8 16}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof$b -o p -- python bench.py --batch $b --steps 10 --warmup 3 --passes 1 --no-strong --no-cpu-baseline --no-mode-sweep --no-kernel-timing $EXTRA > $out/bench${b}_under_rocprof.json 2> $out/prof$b.err
  rm -f $out/prof$b/*/p_kernel_trace.csv $out/prof$b/p_kernel_trace.csv
  s=$(find $out/prof$b -name "*kernel_stats.csv" | head -1)
  python profiles/summarize.py stats $s auto > $out/per_step_table_${b}proteins.txt
  cp $s $out/bench${b}_kernel_stats.csv
  rm -rf $out/prof$b
done
head -40 $out/per_step_table_16proteins.txt
