cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_longk
python profiles/tools/r06_longk.py 2048 4096 8192 > gpurun_out/r06_longk/default.txt 2>&1
PTAMD_LIB_TAG=ti1 python profiles/tools/r06_longk.py 2048 4096 8192 > gpurun_out/r06_longk/ti1.txt 2>&1
PTAMD_LIB_TAG=ns4 python profiles/tools/r06_longk.py 2048 4096 > gpurun_out/r06_longk/ns4.txt 2>&1
cat gpurun_out/r06_longk/default.txt
