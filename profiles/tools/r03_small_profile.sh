#!/bin/bash
# per-kernel times of a small-batch step: bash profiles/tools/r03_small_profile.sh <batch>
cd /tmp && export TMPDIR=/tmp
B=${1:-4}
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b$B -o b$B -- python $R/bench.py ${PT_BENCH_ARGS:---batch $B} --steps 10 --warmup 3 --no-cpu-baseline --no-mode-sweep --no-kernel-timing --no-side-stream > $R/gpurun_out/prof_b$B.log 2>&1
f=$(find $R/gpurun_out/prof_b$B -name "*kernel_stats.csv" | head -n 1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over 13 steps = {tot/13e6:.3f} ms/step")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f'{r["Name"][:90]:90s} calls {int(r["Calls"]):5d} avg {float(r["AverageNs"])/1e3:8.1f} us  per step {float(r["TotalDurationNs"])/13e3:8.1f} us')
PY
