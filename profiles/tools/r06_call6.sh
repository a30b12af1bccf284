cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_attn_split; mkdir -p $out
python -m pytest tests/test_gpu_attention_fused.py tests/test_gpu_kernels.py tests/test_gpu_kv_planes.py tests/test_gpu_model.py tests/test_gpu_auto_guard.py tests/test_gpu_dp.py -m gpu -q 2>&1 | tail -25 > $out/tests2.log
cat $out/tests2.log
