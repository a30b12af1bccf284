cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_call11; mkdir -p $out
python -m pytest tests/test_gpu_kv_planes.py tests/test_gpu_attention_fused.py tests/test_gpu_scales.py tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -4 > $out/tests.log; cat $out/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
BATCHES="4 16" bash profiles/tools/r06_small.sh r06_call11 now kvoff:BENCH_ARGS=--no-kv-planes
BATCHES="4" EXTRA="" bash profiles/tools/r06_small_profile.sh r06_call11_prof > /dev/null 2>&1
grep "layernorm_bwd_dropout" gpurun_out/r06_call11_prof/per_step_table_4proteins.txt
