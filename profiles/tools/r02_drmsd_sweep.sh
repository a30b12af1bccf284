#!/bin/bash
# dRMSD pair kernel: rows per block x unroll sweep, compiled on the GPU box
cd $GRAFT_REPO_ROOT/protein_transformer_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I../../include -x hip"
for pb in 128 64 256; do for u in 4 8; do
  hipcc $FLAGS -DPT_DRMSD_PB=$pb -DPT_DRMSD_UNROLL=$u -c drmsd.hip -o build/drmsd.hip.o 2>/dev/null && hipcc -shared -fPIC --offload-arch=gfx950 -o libptamd.so build/*.o
  echo -n "PB=$pb UNROLL=$u: "; (cd ../.. && python profiles/tools/r02_drmsd_bench.py 2>&1 | grep drmsd)
done; done
