#!/bin/bash
# dk = 32 split attention: kernel tests, configuration tests, micro-benchmark, configs 2/3 bench lines
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -q -m gpu -k attention -x 2>&1 | tail -5 > gpurun_out/dk32_tests.log
python -m pytest tests/test_gpu_configs.py tests/test_gpu_conv.py tests/test_gpu_model.py -q -m gpu 2>&1 | tail -8 >> gpurun_out/dk32_tests.log
python profiles/tools/r02_attn_bench.py 20 > gpurun_out/dk32_attn_bench.txt 2>&1
python bench.py --config 2 --no-cpu-baseline > gpurun_out/dk32_bench_cfg2.json 2> gpurun_out/dk32_bench_cfg2.err
python bench.py --config 3 --no-cpu-baseline > gpurun_out/dk32_bench_cfg3.json 2> gpurun_out/dk32_bench_cfg3.err
