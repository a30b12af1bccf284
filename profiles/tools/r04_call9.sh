cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
for t in 128 256; do PTAMD_HP_TILE=$t timeout 900 python -m pytest tests/test_gpu_gemm_hp.py -x -q -m gpu 2>&1 | tail -n 3; done | tee gpurun_out/r4h/tests1.log
python profiles/tools/r04_gemm_products.py 20 16384 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4h/gemm_products.txt
for i in 1 2; do
for t in 128 256 auto; do if [ $t = auto ]; then unset PTAMD_HP_TILE; else export PTAMD_HP_TILE=$t; fi; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mode-sweep --no-kernel-timing 2>gpurun_out/r4h/bench_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[tile $t]', d['ms_per_step'])"; done; done 2>&1 | tee gpurun_out/r4h/tile_ab.txt
unset PTAMD_HP_TILE
timeout 900 python -m pytest tests/test_gpu_auto_guard.py -x -q -m gpu -k "guard_measures or side_stream" 2>&1 | tail -n 5 | tee gpurun_out/r4h/tests2.log
