#!/bin/bash
# round 4, call 21: where the wave cycles go (SQ counters, two PMC passes of two bench steps, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r4_sq; mkdir -p $out
cmd="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-mode-sweep --no-side-stream"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $out/p1 -o p -- $cmd > $out/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --output-format csv -d $out/p2 -o p -- $cmd > $out/p2.log 2>&1
rm -f $out/p*/p_kernel_trace.csv $out/p*/*/p_kernel_trace.csv
f1=$(find $out/p1 -name "*counter_collection.csv" | head -1); f2=$(find $out/p2 -name "*counter_collection.csv" | head -1)
python profiles/summarize.py sq $f1 > $out/sq_pass1.txt; python profiles/summarize.py sq $f2 > $out/sq_pass2.txt
tail -3 $out/p1.log $out/p2.log; cat $out/sq_pass1.txt
# keep the merged output small
gzip -f $f1 $f2
