#!/bin/bash
# per-product times of a layer with two builds of the library, alternating: r03_ab_lib.sh <tagA> <tagB> [T ...]   ("" = product)
A=$1; B=$2; shift 2
for T in ${@:-16384}; do for rep in 1 2; do for tag in "$A" "$B"; do
  echo "=== T=$T lib=${tag:-product}"
  PTAMD_LIB_TAG=$tag timeout 300 python profiles/tools/r03_gemm_products.py 20 $T 2>&1 | grep -v "^T =\|amdgpu.ids" | cut -c1-84
done; done; done
