#!/bin/bash
# bench.py through torch.distributed.run on ONE GPU: 1 rank over RCCL, then 2 ranks sharing the device over gloo (RCCL
# refuses two ranks per device) - the self-check block of round 4: communication.backend / rccl_version / ranks_seen,
# param_checksum_spread (must be 0), and the --verify-dp pre-pass (sum over ranks of the all-reduced gradient against the
# gradient of the whole global batch computed by rank 0 alone)
pick='import json,sys; d=json.loads(sys.stdin.read()); c=d["communication"]; print(sys.argv[1], d["ms_per_step"], d["value"], d["scaling"], json.dumps({k: c.get(k) for k in ("backend","rccl_version","world_size","ranks_ok","distinct_devices","allreduce_wait_ms","comm_bytes","param_checksum_spread","verify_dp")})); print("  ranks_seen", json.dumps(c["ranks_seen"]))'
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-mode-sweep --verify-dp 2>&1 | tail -1 | python -c "$pick" "single process"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-mode-sweep 2>&1 | tail -1 | python -c "$pick" "1 rank torchrun"
for args in "--batch 8" "--global-batch 16" "--config 5 --steps 4" "--config 3 --batch 8"; do
PTAMD_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 $args --no-mode-sweep --verify-dp 2>&1 | tail -1 | python -c "$pick" "2 ranks gloo [$args]"
done
