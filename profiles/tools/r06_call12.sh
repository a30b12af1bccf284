cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_call12; mkdir -p $out
python -m pytest tests/test_gpu_attention_fused.py tests/test_gpu_kernels.py tests/test_gpu_kv_planes.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -4 > $out/tests.log; cat $out/tests.log
BATCHES="4" bash profiles/tools/r06_small.sh r06_call12 quarters halves:PTAMD_ATTN_FWD_QUARTERS=0 quarters2 halves2:PTAMD_ATTN_FWD_QUARTERS=0
BATCHES="4" EXTRA="" bash profiles/tools/r06_small_profile.sh r06_call12_prof > /dev/null 2>&1
grep "attn_fwd" gpurun_out/r06_call12_prof/per_step_table_4proteins.txt
