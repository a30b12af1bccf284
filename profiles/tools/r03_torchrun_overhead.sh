#!/bin/bash
# where do 0.4 ms go when bench.py runs under torch.distributed.run with ONE rank (no collective is issued at world size 1)?
B="--steps 20 --warmup 3 --no-cpu-baseline --no-mode-sweep --no-kernel-timing"
ms() { python -c "import json,sys; print('$1', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
for rep in 1 2; do
python bench.py $B 2>/dev/null | ms "plain"
OMP_NUM_THREADS=1 python bench.py $B 2>/dev/null | ms "plain, OMP_NUM_THREADS=1"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 $B 2>/dev/null | ms "torchrun 1 rank"
OMP_NUM_THREADS=16 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 $B 2>/dev/null | ms "torchrun 1 rank, OMP_NUM_THREADS=16"
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 python bench.py --gpus 1 $B 2>/dev/null | ms "env only (process group of 1, no launcher)"
done
