cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
for a in "1 300 1 0" "1 537 1 0" "1 512 1 0.1" "2 512 8 0.1" "1 700 2 0"; do python profiles/tools/r04_attn_debug.py $a; done 2>&1 | tee gpurun_out/r4d/attn_debug.txt
timeout 600 python -m pytest tests/test_gpu_gemm_hp.py -x -q -m gpu 2>&1 | tail -n 5 | tee gpurun_out/r4d/hp_tests.log
python profiles/tools/r04_gemm_products.py 20 16384 2>&1 | tee gpurun_out/r4d/gemm_products.txt
python profiles/tools/r03_drmsd_bench.py 2>&1 | tee gpurun_out/r4d/drmsd.txt
PTAMD_LIB_TAG=u8 python profiles/tools/r03_drmsd_bench.py 2>&1 | tee -a gpurun_out/r4d/drmsd.txt
timeout 600 python -m pytest tests/test_gpu_loss_path.py -x -q -m gpu 2>&1 | tail -n 5 | tee gpurun_out/r4d/loss_tests.log
