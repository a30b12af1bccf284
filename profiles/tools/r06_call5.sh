cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_attn_split; mkdir -p $out
python -m pytest tests/test_gpu_attention_fused.py tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -15 > $out/tests.log
cat $out/tests.log
BATCHES="4 8 16" bash profiles/tools/r06_small.sh r06_attn_split split three:PTAMD_ATTN_FUSED=0
