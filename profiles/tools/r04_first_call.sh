cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
python bench.py --steps 20 --warmup 3 > gpurun_out/r4a/bench0.json 2> gpurun_out/r4a/bench0.err
timeout 1500 python -m pytest tests/test_gpu_auto_guard.py -x -q -m gpu > gpurun_out/r4a/guard_tests.log 2>&1
bash profiles/tools/r04_dp_smoke.sh > gpurun_out/r4a/dp_smoke.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_scales.py tests/test_gpu_dp.py -x -q -m gpu > gpurun_out/r4a/other_tests.log 2>&1
tail -5 gpurun_out/r4a/*.log; cat gpurun_out/r4a/dp_smoke.txt | cut -c1-600
