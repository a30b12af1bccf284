run() { python bench.py $1 --steps 20 --warmup 3 --no-cpu-baseline --no-mode-sweep --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'])"; }
for rep in 1 2; do for b in 4 8 16; do run "--batch $b"; run "--batch $b --no-side-stream"; done; done
run "--config 2"; run "--config 2 --no-side-stream"; run "--config 1"; run "--config 1 --no-side-stream"
