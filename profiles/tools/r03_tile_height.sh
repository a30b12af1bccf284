#!/bin/bash
# 128-row against 256-row tiles of the staging GEMM, per product of a layer, at several token counts.
# Libraries: libptamd_ti4.so (-DPT_FORCE_TI=4), libptamd_ti2.so (-DPT_FORCE_TI=2), libptamd.so (chooses by the cost model).
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k gemm 2>&1 | tail -n 3
for T in 2048 4096 8192 16384; do
  for tag in ti4 ti2 ""; do
    echo "=== T=$T lib=${tag:-product}"
    PTAMD_LIB_TAG=$tag timeout 300 python profiles/tools/r03_gemm_products.py 20 $T 2>&1 | grep -v "^T ="
  done
done
