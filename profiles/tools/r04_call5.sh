cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
timeout 900 python -m pytest tests/test_gpu_attention_fused.py -x -q -m gpu 2>&1 | tail -n 12 | tee gpurun_out/r4e/attn_fused.log
python profiles/tools/r04_attn_debug.py 3 537 4 0 2>&1 | tail -n 5 | tee gpurun_out/r4e/attn_debug.txt
python profiles/tools/r04_hp_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4e/hp_sweep.txt
PTAMD_LIB_TAG=nostore python profiles/tools/r04_hp_sweep.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4e/hp_sweep.txt
timeout 600 python -m pytest tests/test_gpu_loss_path.py -x -q -m gpu 2>&1 | tail -n 5 | tee gpurun_out/r4e/loss_tests.log
python profiles/tools/r03_drmsd_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4e/drmsd.txt
for i in 1 2; do
for f in 1 0; do PTAMD_ATTN_FUSED=$f python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mode-sweep --no-kernel-timing 2>gpurun_out/r4e/bench_err_$f.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused=$f', d['ms_per_step'], d['auto_fallbacks_per_step'])"; done; done 2>&1 | tee gpurun_out/r4e/attn_ab.txt
tail -n 3 gpurun_out/r4e/bench_err_1.txt
