#!/bin/bash
# The measurement set of a round on ONE MI355X box (one gpurun call = one box), parameterised - replaces the numbered one-off
# scripts of round 4.     usage: collect.sh <round, e.g. r05> <tag> [full] [sq] [small]
#   always: headline bench line; rocprofv3 --kernel-trace --stats of the same command (with --no-side-stream: kernels of two
#           streams that overlap report durations that include their waiting for CUs) -> per-step kernel table
#   full  : + PMC traffic passes (separate runs, --kernel-trace only: FETCH_SIZE, WRITE_SIZE), the other BASELINE configurations,
#           the per-GPU steps of a strongly scaled job (4 / 8 / 16 proteins)
#   sq    : + where the wave cycles go (two SQ counter passes of two bench steps)
#   small : + kernel table of the 4-protein step (the per-GPU share at 8 GPUs)
set -x
round=${1:-r05}; tag=${2:-x}; shift 2
want() { for a in "$@"; do :; done; case " $ARGS " in *" $1 "*) return 0;; esac; return 1; }
ARGS="$*"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${round}_$tag
mkdir -p $out
python bench.py --steps 20 --warmup 3 > $out/bench_cfg4.json 2> $out/bench_cfg4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python bench.py --steps 5 --warmup 2 --passes 1 --no-strong --no-cpu-baseline --no-mode-sweep --no-side-stream > $out/bench_under_rocprof.json 2> $out/prof.err
rm -f $out/prof/*/p_kernel_trace.csv $out/prof/p_kernel_trace.csv
stats=$(find $out/prof -name "*kernel_stats.csv" | head -1)
python profiles/summarize.py stats $stats auto > $out/per_step_table.txt
cp $stats $out/bench_steps5_kernel_stats.csv
if want small; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof4 -o p -- python bench.py --batch 4 --steps 5 --warmup 2 --passes 1 --no-strong --no-cpu-baseline --no-mode-sweep --no-kernel-timing > $out/bench4_under_rocprof.json 2> $out/prof4.err
  rm -f $out/prof4/*/p_kernel_trace.csv $out/prof4/p_kernel_trace.csv
  s4=$(find $out/prof4 -name "*kernel_stats.csv" | head -1)
  python profiles/summarize.py stats $s4 auto > $out/per_step_table_4proteins.txt
  cp $s4 $out/bench4_steps5_kernel_stats.csv
fi
if want full; then
  quiet="--passes 1 --no-strong --no-cpu-baseline --no-kernel-timing --no-mode-sweep"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o p -- python bench.py --steps 2 --warmup 1 $quiet > $out/pmc_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o p -- python bench.py --steps 2 --warmup 1 $quiet > $out/pmc_write.log 2>&1
  rm -f $out/pmc_*/p_kernel_trace.csv $out/pmc_*/*/p_kernel_trace.csv
  # (20 warm-up steps: the launch-bound configs 1 and 2 speed up by 10 % over the first passes of a process)
  for c in 1 2 3 5; do python bench.py --config $c --steps 20 --warmup 20 > $out/bench_cfg$c.json 2> $out/bench_cfg$c.err; done
  f=$(find $out/pmc_fetch -name "*counter_collection.csv" | head -1); w=$(find $out/pmc_write -name "*counter_collection.csv" | head -1)
  alg=$(python -c "import json; print(json.load(open('$out/bench_cfg4.json'))['roofline']['gemm_family']['algorithmic_bytes_per_launch'])")
  python profiles/summarize.py traffic $f $w $alg $out/gemm_hbm_traffic.json > /dev/null
  python profiles/summarize.py step_traffic $f $w auto > $out/hbm_traffic_per_step.txt
  gzip -f $f $w
fi
if want sq; then
  cmd="python bench.py --steps 2 --warmup 1 --passes 1 --no-strong --no-cpu-baseline --no-kernel-timing --no-mode-sweep --no-side-stream"
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $out/sq1 -o p -- $cmd > $out/sq1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --output-format csv -d $out/sq2 -o p -- $cmd > $out/sq2.log 2>&1
  rm -f $out/sq*/p_kernel_trace.csv $out/sq*/*/p_kernel_trace.csv
  f1=$(find $out/sq1 -name "*counter_collection.csv" | head -1); f2=$(find $out/sq2 -name "*counter_collection.csv" | head -1)
  python profiles/summarize.py sq $f1 > $out/sq_counters_pass1.txt; python profiles/summarize.py sq $f2 > $out/sq_counters_pass2.txt
  gzip -f $f1 $f2
fi
ls -la $out
