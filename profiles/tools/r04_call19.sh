#!/bin/bash
# round 4, call 19: grouping policy "auto": tests, step times of all configs, batch sweep
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r4s; rm -f gpurun_out/r4s/*.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "grouping or group or side_stream or model or train or dp or scales" 2>&1 | grep -v amdgpu.ids | tail -30 | tee gpurun_out/r4s/tests.txt
run() {
  timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-mode-sweep "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags [$*]', d['ms_per_step'], d['roofline']['frac'])" | tee -a gpurun_out/r4s/step_ab.txt
}
for i in 1 2 3; do run; run --no-top-layer-scales; run --dw-group off --no-top-layer-scales; done
for cfg in 2 5; do for i in 1 2; do run --config $cfg; run --config $cfg --dw-group off --no-top-layer-scales; done; done

