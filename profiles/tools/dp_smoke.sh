#!/bin/bash
# bench.py through torch.distributed.run on ONE GPU - 1 rank over RCCL, then 2 ranks sharing the device over gloo (RCCL
# refuses two ranks per device): the self-check block (param_checksum_spread must be 0, --verify-dp), the strong-scaling block
# (global batch split over the ranks) and the hook trace (bytes per reduced slice, stream each reduction was issued from)
pick='import json,sys; d=json.loads(sys.stdin.read()); c=d["communication"]; print(sys.argv[1], d["ms_per_step"], d["value"], d["scaling"], json.dumps({k: c.get(k) for k in ("backend","rccl_version","world_size","ranks_ok","distinct_devices","allreduce_wait_ms","comm_bytes","param_checksum_spread","verify_dp","allreduce_bytes_per_layer")})); print("  strong", json.dumps(d.get("strong_scaling"))); print("  hooks", json.dumps((c.get("hooks") or [])[:3]), "..."); print("  kv planes / prep launches", d.get("kv_plane_layer_passes"), d.get("weights_prep_launches"))'
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --passes 1 --no-cpu-baseline --no-mode-sweep 2>&1 | tail -1 | python -c "$pick" "1 rank torchrun"
for args in "--batch 16" "--batch 8" "--config 5 --steps 4" "--config 3 --batch 8"; do
PTAMD_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --passes 1 $args --no-mode-sweep --verify-dp 2>&1 | tail -1 | python -c "$pick" "2 ranks gloo [$args]"
done
