#!/bin/bash
# round 5: per-kernel A/B of the pre-split K / V path (rocprofv3 kernel stats of 5 bench steps each): planes with 64-key stages,
# planes with 32-key stages (PTAMD_ATTN_KVP_TPS=1), fp32 K / V (--no-kv-planes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in kv kv_tps1 nokv; do
  flag=""; [ $v = nokv ] && flag="--no-kv-planes"
  unset PTAMD_ATTN_KVP_TPS; [ $v = kv_tps1 ] && export PTAMD_ATTN_KVP_TPS=1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_ab_$v -o p -- python bench.py --steps 5 --warmup 2 --passes 1 --no-strong --no-cpu-baseline --no-mode-sweep --no-side-stream --no-kernel-timing $flag > /dev/null 2>&1
  f=$(find gpurun_out/r05_ab_$v -name "*kernel_stats.csv" | head -1)
  python profiles/summarize.py stats $f auto > gpurun_out/r05_ab_${v}_table.txt
  rm -rf gpurun_out/r05_ab_$v
done
grep -h "attn_fwd\|attn_bwd_fused\|gemm_hp3" gpurun_out/r05_ab_kv_table.txt gpurun_out/r05_ab_kv_tps1_table.txt gpurun_out/r05_ab_nokv_table.txt | cut -c1-64,75-
