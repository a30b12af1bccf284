cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
timeout 900 python -m pytest tests/test_gpu_gemm_hp.py tests/test_gpu_attention_fused.py tests/test_gpu_loss_path.py -x -q -m gpu 2>&1 | tail -n 8 | tee gpurun_out/r4f/tests1.log
python profiles/tools/r03_drmsd_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4f/drmsd.txt
PTAMD_LIB_TAG=u4 python profiles/tools/r03_drmsd_bench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4f/drmsd.txt
python profiles/tools/r04_gemm_products.py 20 16384 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4f/gemm_products.txt
for i in 1 2; do
for f in "" "--no-hp-qkv" "--no-hp-dx" "--no-hp-qkv --no-hp-dx"; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mode-sweep --no-kernel-timing $f 2>gpurun_out/r4f/bench_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$f]', d['ms_per_step'], d['auto_fallbacks_per_step'])"; done; done 2>&1 | tee gpurun_out/r4f/hp_ab.txt
tail -n 3 gpurun_out/r4f/bench_err.txt
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_scales.py -x -q -m gpu 2>&1 | tail -n 8 | tee gpurun_out/r4f/tests2.log
