#!/bin/bash
# ablations of the hp GEMM, variant 2: zero operands (clock), no fragment reads, no DMA, neither
export PTAMD_HP_VARIANT=2
echo "== zero operands"; HP_BENCH_ZERO=1 python profiles/tools/r02_gemm_hp_bench.py 10 2>&1 | grep -v amdgpu.ids | cut -c1-48
for a in 0 2 4 6; do echo "== PTAMD_HP_ABLATE=$a"; PTAMD_HP_ABLATE=$a python profiles/tools/r02_gemm_hp_bench.py 10 2>&1 | grep -v amdgpu.ids | cut -c1-48; done
