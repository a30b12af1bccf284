cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_call8; mkdir -p $out
python -m pytest tests/test_gpu_attention_fused.py -m gpu -q -x 2>&1 | tail -3 > $out/tests.log; cat $out/tests.log
BATCHES="8 16" EXTRA="--no-side-stream" bash profiles/tools/r06_small_profile.sh r06_small_profile_noside > /dev/null
BATCHES="4 8 16" bash profiles/tools/r06_small.sh r06_call8 now
