cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
timeout 900 python -m pytest tests/test_gpu_attention_fused.py -x -q -m gpu > gpurun_out/r4c/attn_fused.log 2>&1
tail -n 15 gpurun_out/r4c/attn_fused.log
# A/B: the default bench with and without the fused attention backward, alternating
for i in 1 2; do
for f in 1 0; do PTAMD_ATTN_FUSED=$f python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mode-sweep --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused=$f', d['ms_per_step'], d['auto_fallbacks_per_step'])"; done; done | tee gpurun_out/r4c/attn_ab.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4c/prof -o r4 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-mode-sweep --no-side-stream > gpurun_out/r4c/bench_prof.json 2> gpurun_out/r4c/prof.err
rm -f gpurun_out/r4c/prof/*/r4_kernel_trace.csv gpurun_out/r4c/prof/r4_kernel_trace.csv
python profiles/summarize.py stats $(find gpurun_out/r4c/prof -name "*kernel_stats.csv" | head -1) 7 > gpurun_out/r4c/per_step.txt; head -20 gpurun_out/r4c/per_step.txt
timeout 2400 python -m pytest tests/test_gpu_auto_guard.py -q -m gpu > gpurun_out/r4c/guard_tests.log 2>&1
tail -n 12 gpurun_out/r4c/guard_tests.log
timeout 900 python -m pytest tests/test_gpu_parity_record.py tests/test_gpu_kernels.py -q -m gpu -k "parity or attention" > gpurun_out/r4c/other_tests.log 2>&1
tail -n 8 gpurun_out/r4c/other_tests.log
