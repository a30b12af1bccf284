run() { ( cd $1; python bench.py $2 --steps 20 --warmup 3 --no-cpu-baseline --no-mode-sweep --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 $2', d['ms_per_step'])" ); }
for rep in 1 2; do for args in "--config 2" "--batch 4" "--batch 16" "--config 1"; do run ab_r02 "$args"; run . "$args"; done; done
