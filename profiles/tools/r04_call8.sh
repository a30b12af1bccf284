cd $GRAFT_REPO_ROOT
bash profiles/tools/r04_collect.sh v1 full > gpurun_out/r4_v1_collect.log 2>&1
head -30 gpurun_out/r4_v1/per_step_table.txt
python -c "
import json; d=json.load(open('gpurun_out/r4_v1/bench_cfg4.json')); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['f32_equivalent'], d['arithmetic_modes'], d['auto_guard'])"
bash profiles/tools/r04_parity_seeds.sh 8
