run() { PTAMD_DW_SLOTS=$1 python bench.py $2 --steps 20 --warmup 3 --no-cpu-baseline --no-mode-sweep --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('slots $1 [$2]', d['ms_per_step'])"; }
for rep in 1 2; do for s in 512 256 128 1024; do run $s ""; done; done
run 512 "--no-side-stream"; run 256 "--no-side-stream"
