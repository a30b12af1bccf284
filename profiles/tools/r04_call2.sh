cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
timeout 2400 python -m pytest tests/test_gpu_auto_guard.py -q -m gpu > gpurun_out/r4b/guard_tests.log 2>&1
bash profiles/tools/r04_dp_smoke.sh > gpurun_out/r4b/dp_smoke.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_dp.py tests/test_gpu_parity_record.py -x -q -m gpu > gpurun_out/r4b/other_tests.log 2>&1
tail -n 30 gpurun_out/r4b/guard_tests.log; tail -n 5 gpurun_out/r4b/other_tests.log; cut -c1-700 gpurun_out/r4b/dp_smoke.txt
