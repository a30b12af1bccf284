#!/bin/bash
# round-4 measurement set on one MI355X box: headline bench line, rocprofv3 kernel stats of the same command (with --no-side-stream:
# kernels of two streams that overlap report durations that include their waiting for CUs), PMC traffic
# passes (separate runs, --kernel-trace only), the other BASELINE configurations.   usage: r04_collect.sh <tag> [full]
set -x
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r4_$tag
mkdir -p $out
python bench.py --steps 20 --warmup 3 > $out/bench_cfg4.json 2> $out/bench_cfg4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o r4 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-mode-sweep --no-side-stream > $out/bench_under_rocprof.json 2> $out/prof.err
rm -f $out/prof/*/r4_kernel_trace.csv $out/prof/r4_kernel_trace.csv
if [ "$2" = "full" ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-mode-sweep > $out/pmc_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-mode-sweep > $out/pmc_write.log 2>&1
  rm -f $out/pmc_*/p_kernel_trace.csv $out/pmc_*/*/p_kernel_trace.csv
  for c in 1 2 3 5; do python bench.py --config $c --steps 20 --warmup 20 > $out/bench_cfg$c.json 2> $out/bench_cfg$c.err; done   # (20 warm-up steps: the launch-bound configs 1 and 2 speed up by 10 % over the first passes of a process)
  for b in 4 8 16; do python bench.py --batch $b --steps 20 --warmup 3 --no-cpu-baseline --no-mode-sweep --no-kernel-timing > $out/bench_batch$b.json 2>/dev/null; done
fi
find $out -name "*.csv" | head -20
# summaries (the numbers bench.py and DESIGN.md quote): per-step kernel table, HBM traffic of the GEMM launches (with the
# digest of the GEMM sources it was measured on: bench.py refuses a stale record), HBM traffic per step by kernel
stats=$(find $out/prof -name "*kernel_stats.csv" | head -1)
python profiles/summarize.py stats $stats auto > $out/per_step_table.txt
if [ "$2" = "full" ]; then
  f=$(find $out/pmc_fetch -name "*counter_collection.csv" | head -1); w=$(find $out/pmc_write -name "*counter_collection.csv" | head -1)
  alg=$(python -c "import json; print(json.load(open('$out/bench_cfg4.json'))['roofline']['algorithmic_bytes_per_launch'])")
  python profiles/summarize.py traffic $f $w $alg $out/gemm_hbm_traffic.json > /dev/null
  python profiles/summarize.py step_traffic $f $w auto > $out/hbm_traffic_per_step.txt
fi
