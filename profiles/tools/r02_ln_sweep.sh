#!/bin/bash
# persistent-grid size of the LayerNorm backward kernels (blocks of 4 wavefronts; 256 CUs): rebuilds on the box
for nb in 512 768 1024 1536; do
  PTAMD_EXTRA_FLAGS=-DPT_LN_BWD_BLOCKS=$nb python -m protein_transformer_amd.build > /dev/null 2>&1
  echo "== LN_BWD_BLOCKS=$nb"
  python profiles/tools/r02_ln_bench.py 2>&1 | grep "minima\|no scale"
done
