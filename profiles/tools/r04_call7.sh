cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
timeout 900 python -m pytest tests/test_gpu_gemm_hp.py -x -q -m gpu 2>&1 | tail -n 4 | tee gpurun_out/r4g/tests1.log
for t in product r3d u4 product r3d; do PTAMD_LIB_TAG=$t python profiles/tools/r03_drmsd_bench.py 2>&1 | grep -v amdgpu.ids; done | sed 's/lib product/lib product(U8)/' | tee gpurun_out/r4g/drmsd.txt
python profiles/tools/r04_gemm_products.py 20 16384 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4g/gemm_products.txt
python profiles/tools/r04_hp_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4g/hp_sweep.txt
for i in 1 2; do
for f in "" "--no-hp-qkv"; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mode-sweep --no-kernel-timing $f 2>gpurun_out/r4g/bench_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$f]', d['ms_per_step'], d['auto_fallbacks_per_step'])"; done; done 2>&1 | tee gpurun_out/r4g/hp_ab.txt
