#!/bin/bash
# Round 6: the round's records on ONE box: collection (bench line, rocprof tables, PMC traffic, SQ counters, other configs, small
# batches), soak (memory must be flat), power, DP smoke.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash profiles/tools/collect.sh r06 final full sq small > gpurun_out/r06_collect.log 2>&1
python profiles/tools/soak.py --steps 4000 > gpurun_out/r06_final/soak.txt 2>&1
bash profiles/tools/step_power.sh r06 > /dev/null 2>&1
bash profiles/tools/dp_smoke.sh > gpurun_out/r06_final/dp_smoke.txt 2>&1
BATCHES="4" EXTRA="" bash profiles/tools/r06_small_profile.sh r06_final_small > /dev/null 2>&1
tail -12 gpurun_out/r06_final/soak.txt; cat gpurun_out/r06_step_power.txt | tail -8; head -c 1500 gpurun_out/r06_final/bench_cfg4.json
