run() { python bench.py $1 --steps 20 --warmup 3 --no-cpu-baseline --no-mode-sweep --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$1]', d['ms_per_step'])"; }
for rep in 1 2 3; do run ""; run "--no-side-stream"; run "--no-hp-forward"; done
