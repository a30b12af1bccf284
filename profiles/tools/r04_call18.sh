#!/bin/bash
# round 4, call 18: grouping of the weight-gradient products: whole layer / FFN pair + attention pair / attention pair only / none, with and without the side stream
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r4r; rm -f gpurun_out/r4r/*.txt
run() {  # $1 = PTAMD_DW_GROUP, $2.. = bench flags
  g=$1; shift
  PTAMD_DW_GROUP=$g timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-mode-sweep "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('group $g flags [$*]', d['ms_per_step'], d['roofline']['frac'])" | tee -a gpurun_out/r4r/step_ab.txt
}
for i in 1 2; do
  run layer; run pairs; run tail; run layer --no-group-dw
  run layer --no-side-stream; run pairs --no-side-stream; run layer --no-group-dw --no-side-stream
done
for cfg in 3 5; do
  run layer --config $cfg; run pairs --config $cfg; run tail --config $cfg; run layer --no-group-dw --config $cfg
done
