cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_vs_r05; mkdir -p $out
q="--steps 20 --warmup 20 --passes 5 --no-strong --no-cpu-baseline --no-mode-sweep --no-kernel-timing"
for rep in 1 2; do
for c in 1 2; do
  python bench.py --config $c $q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r06 cfg$c', d['ms_per_step'], d['passes']['ms_per_step'], d['resident']['ms_per_step'])" | tee -a $out/ab.txt
  (cd _r05tree && python bench.py --config $c $q 2>/dev/null) | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r05 cfg$c', d['ms_per_step'], d['passes']['ms_per_step'], d['resident']['ms_per_step'])" | tee -a $out/ab.txt
done
done
python bench.py $q --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r06 cfg4', d['ms_per_step'], d['passes']['ms_per_step'], d['resident']['ms_per_step'])" | tee -a $out/ab.txt
(cd _r05tree && python bench.py $q --warmup 5 2>/dev/null) | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r05 cfg4', d['ms_per_step'], d['passes']['ms_per_step'], d['resident']['ms_per_step'])" | tee -a $out/ab.txt
