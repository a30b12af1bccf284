#!/bin/bash
# build libptamd_trace.so from a temporarily instrumented copy of the sources, then restore them:  r03_trace_build.sh {gemm|hp}
cd "$(dirname "$0")/../.."
C=protein_transformer_amd/csrc
for f in gemm_split_kernel.h gemm_hp.hip gemm_common.h gemm.hip; do cp $C/$f /tmp/_trace_bk_$f; done
python profiles/tools/r03_trace_patch_${1:-gemm}.py && PTAMD_BUILD_TAG=trace python -m protein_transformer_amd.build 2>&1 | tail -n 1
for f in gemm_split_kernel.h gemm_hp.hip gemm_common.h gemm.hip; do cp /tmp/_trace_bk_$f $C/$f; done
git status --short $C
