#!/bin/bash
# Round 6: the -m gpu suite with durations + the bench line (parity block) on one box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-r06_suite}; mkdir -p $out
( time python -m pytest tests -m gpu -q --durations=40 ) > $out/pytest.log 2>&1
python bench.py --steps 20 --warmup 3 > $out/bench.json 2> $out/bench.err
tail -5 $out/pytest.log; tail -3 $out/bench.err
