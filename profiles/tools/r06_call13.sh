cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_host
python profiles/tools/host_profile.py --batch 4 --steps 200 > gpurun_out/r06_host/host_profile_b4.txt 2>&1
head -60 gpurun_out/r06_host/host_profile_b4.txt
