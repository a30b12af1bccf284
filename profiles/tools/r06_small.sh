#!/bin/bash
# Round 6: what moves the per-GPU share of a strongly scaled job (4 / 8 / 16 proteins x 512)?  Host-side knobs and tagged
# builds, alternating on ONE box.     usage: r06_small.sh <out tag> <variant> ...   variant = name[:ENV=VALUE[,ENV=VALUE...]]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-r06_small}; shift; mkdir -p $out
quiet="--steps 20 --warmup 5 --passes 3 --no-strong --no-cpu-baseline --no-mode-sweep --no-kernel-timing"
for rep in 1 2; do
for v in "$@"; do
  name=${v%%:*}; envs=""; [ "$v" != "$name" ] && envs=$(echo ${v#*:} | tr ',' ' ')
  for b in ${BATCHES:-4 8 16}; do
    extra=""; case " $envs " in *" BENCH_ARGS="*) extra=$(echo "$envs" | tr ' ' '\n' | grep '^BENCH_ARGS=' | cut -d= -f2- | tr '+' ' ');; esac
    env $envs python bench.py --batch $b $quiet $extra 2>$out/err_$name.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', 'b=$b', d['ms_per_step'], d['passes']['ms_per_step'], 'resident', d['resident']['ms_per_step'], flush=True)" | tee -a $out/small.txt
  done
done
done
