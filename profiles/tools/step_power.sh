#!/bin/bash
# shader / memory clock and package power while the headline training step loops (rocm-smi sampled twice a second), and idle
round=${1:-r06}
out=gpurun_out/${round}_step_power.txt
echo "idle:" > $out
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' ' >> $out; echo >> $out
python profiles/tools/soak.py --steps 2500 > gpurun_out/${round}_step_power_soak.txt 2>/dev/null &
PID=$!
sleep 12     # import, set-up, first steps
echo "while the step loops (AUTO = f16x2 GEMMs and attention, 32 x 512):" >> $out
for i in $(seq 1 16); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' ' >> $out; echo >> $out; sleep 0.5; done
wait $PID
tail -n 3 gpurun_out/${round}_step_power_soak.txt >> $out
cat $out
