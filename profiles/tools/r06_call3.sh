cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_small1
python -m pytest tests/test_gpu_configs.py::test_config4_full_size_auto_steady_state tests/test_gpu_scales.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r06_small1/tests.log
python bench.py --batch 4 --steps 20 --warmup 5 --passes 3 --no-strong --no-cpu-baseline --no-mode-sweep > gpurun_out/r06_small1/bench4_timing.json 2>/dev/null
python profiles/tools/r04_gemm_products.py 20 2048 > gpurun_out/r06_small1/gemm_products_2048.txt 2>&1
bash profiles/tools/r06_small.sh r06_small1 default ti1:PTAMD_LIB_TAG=ti1 ns4:PTAMD_LIB_TAG=ns4 gmk512:PTAMD_GROUP_MIN_K=512 gmk256:PTAMD_GROUP_MIN_K=256 side2048:PTAMD_SIDE_MIN_TOKENS=2048
cat gpurun_out/r06_small1/tests.log
