cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4k
( time timeout 3400 python -m pytest tests/ -q -m gpu ) > gpurun_out/r4k/full_suite.log 2>&1
tail -n 15 gpurun_out/r4k/full_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
