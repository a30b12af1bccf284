#!/bin/bash
# bench.py through torch.distributed.run on ONE GPU: 1 rank over RCCL, then 2 ranks sharing the device over gloo
# (RCCL refuses two ranks per device) in weak and strong scaling mode - exercises the multi-rank code path of the bench
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-mode-sweep 2>&1 | tail -1 | cut -c1-400
PTAMD_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --batch 8 --no-mode-sweep 2>&1 | tail -1 | cut -c1-400
PTAMD_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 5 --warmup 2 --global-batch 16 --no-mode-sweep 2>&1 | tail -1 | cut -c1-400
PTAMD_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29514 bench.py --gpus 2 --config 5 --steps 4 --warmup 1 --no-mode-sweep 2>&1 | tail -1 | cut -c1-400
