cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_soak_ab; mkdir -p $out
python profiles/tools/soak.py --steps 3000 > $out/soak_default.txt 2>&1
SOAK_GC_EVERY=64 python profiles/tools/soak.py --steps 3000 > $out/soak_gc64.txt 2>&1
SOAK_GC_OFF=1 python profiles/tools/soak.py --steps 3000 > $out/soak_gcoff.txt 2>&1
PYTORCH_HIP_ALLOC_CONF=expandable_segments:True python profiles/tools/soak.py --steps 3000 > $out/soak_expandable.txt 2>&1
for f in default gc64 gcoff expandable; do echo == $f; grep "^step" $out/soak_$f.txt | awk '{print $2, $4, $13, $15, $17, $18, $20}' | tr '\n' ';'; echo; done
