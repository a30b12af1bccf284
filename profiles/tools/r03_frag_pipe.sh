#!/bin/bash
# explicitly pipelined fragment reads of the f16x2 consumer (libptamd_pipe.so) against the product library, per product of a layer
PTAMD_LIB_TAG=pipe timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k gemm 2>&1 | tail -n 2
for T in 16384 4096; do
  for tag in "" pipe "" pipe; do
    echo "=== T=$T lib=${tag:-product}"
    PTAMD_LIB_TAG=$tag timeout 300 python profiles/tools/r03_gemm_products.py 20 $T 2>&1 | grep -v "^T =" | cut -c1-84
  done
done
