#!/bin/bash
# alternating A/B of ablation builds of the attention kernels (tags = arguments; "product" = the tree's library)
out=gpurun_out/r05_attn_ab.txt; : > $out
for rep in 1 2 3; do for t in "$@"; do
  if [ $t = product ]; then python profiles/tools/r05_attn_bench.py 2>/dev/null >> $out; else PTAMD_LIB_TAG=$t python profiles/tools/r05_attn_bench.py 2>/dev/null >> $out; fi
done; done
cat $out
