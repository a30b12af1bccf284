#!/bin/bash
# bench.py through torch.distributed.run on ONE GPU: 1 rank over RCCL, then 2 ranks sharing the device over gloo
# (RCCL refuses two ranks per device) in weak and strong scaling mode - exercises the multi-rank code path of the bench,
# including the `communication` block (allreduce_wait_ms, comm_bytes) of round 3
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-mode-sweep 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1 rank rccl', d['ms_per_step'], d['value'], d.get('communication'))"
for args in "--batch 8" "--global-batch 16" "--config 5 --steps 4"; do
PTAMD_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 $args --no-mode-sweep 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('2 ranks gloo [$args]', d['ms_per_step'], d['value'], d['scaling'], {k: v for k, v in d['communication'].items() if k != 'what'})"
done
