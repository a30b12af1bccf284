#!/bin/bash
# copy the summaries of a collection (gpurun_out/<round>_<tag>/, written by collect.sh on the GPU box) into profiles/<round>/ under
# the names DESIGN.md and profiles/README.md quote.   usage: adopt.sh <round> <tag>
round=${1:-r05}; tag=${2:-final}
src=gpurun_out/${round}_$tag; dst=profiles/$round; mkdir -p $dst
cp $src/bench_cfg4.json $dst/${round}_${tag}_bench.json
for c in 1 2 3 5; do [ -f $src/bench_cfg$c.json ] && cp $src/bench_cfg$c.json $dst/${round}_${tag}_bench_cfg$c.json; done
cp $src/bench_under_rocprof.json $dst/${round}_${tag}_bench_under_rocprof.json
cp $src/bench_steps5_kernel_stats.csv $dst/${round}_${tag}_bench_steps5_kernel_stats.csv
cp $src/per_step_table.txt $dst/${round}_${tag}_per_step_table.txt
[ -f $src/per_step_table_4proteins.txt ] && cp $src/per_step_table_4proteins.txt $dst/${round}_${tag}_per_step_table_4proteins.txt && cp $src/bench4_steps5_kernel_stats.csv $dst/${round}_${tag}_bench4_steps5_kernel_stats.csv
[ -f $src/gemm_hbm_traffic.json ] && cp $src/gemm_hbm_traffic.json $dst/${round}_gemm_hbm_traffic.json && cp $src/hbm_traffic_per_step.txt $dst/${round}_${tag}_hbm_traffic_per_step.txt
[ -f $src/sq_counters_pass1.txt ] && cp $src/sq_counters_pass1.txt $dst/${round}_${tag}_sq_counters_pass1.txt && cp $src/sq_counters_pass2.txt $dst/${round}_${tag}_sq_counters_pass2.txt
ls -la $dst
