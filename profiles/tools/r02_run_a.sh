set -x
python -m pytest tests/test_gpu_eval_ckpt.py tests/test_gpu_kernels.py tests/test_gpu_loss_path.py tests/test_gpu_model.py -m gpu -q 2>&1 | tail -15 > gpurun_out/r2_gpu_rest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2_smoke.log 2>&1
python bench.py --steps 20 --warmup 3 > gpurun_out/r2_a_bench_cfg4.json 2> gpurun_out/r2_a_bench_cfg4.err
for c in 1 2 3 5; do python bench.py --config $c --steps 20 --warmup 3 > gpurun_out/r2_a_bench_cfg$c.json 2> gpurun_out/r2_a_bench_cfg$c.err; done
tail -3 gpurun_out/r2_gpu_rest.log gpurun_out/r2_smoke.log
for c in 1 2 3 4 5; do python - <<P
import json
try:
    d=json.load(open("gpurun_out/r2_a_bench_cfg$c.json")); print($c, d["value"], d["ms_per_step"], d["h2d_inclusive"]["ms_per_step"], d["arithmetic_modes"], d["roofline"]["achieved"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["sequential"]["value"])
except Exception as e: print($c, "ERR", e); print(open("gpurun_out/r2_a_bench_cfg$c.err").read()[-1500:])
P
done
