#!/bin/bash
# Round 6, first call: per-test durations of the -m gpu suite (to bring it under 900 s) + the baseline bench line of the box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_first; mkdir -p $out
( time python -m pytest tests -m gpu -q --durations=100 -x ) > $out/pytest.log 2>&1
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-mode-sweep > $out/bench.json 2> $out/bench.err
tail -5 $out/pytest.log
