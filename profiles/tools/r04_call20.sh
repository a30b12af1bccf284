#!/bin/bash
# round 4, call 20: collection v2 (bench line, rocprof kernel stats, PMC traffic, all configs) + the 8-draw parity record
cd /root/repo
bash profiles/tools/r04_collect.sh v2 full > gpurun_out/r4_v2_collect.log 2>&1
bash profiles/tools/r04_parity_seeds.sh 8
