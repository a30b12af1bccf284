#!/bin/bash
# residues/s of the real command-line loop (log.py's speed column, reference log.py:422-430) at BASELINE configs[3],
# to set beside bench.py's number.   usage: r02_cli_throughput.sh <outfile>
out=${1:-gpurun_out/cli_throughput.txt}
d=$(mktemp -d)
python -m protein_transformer_amd.train --synthetic 32,512,60 --name clirun -m enc-only -dm 512 -nl 6 -nh 8 -dih 2048 \
  -l drmsd -b 32 --max_seq_len 512 --train_only -e 3 --log_dir $d/logs --chkpt_dir $d/ck -opt sgd > $d/stdout.txt 2>&1
python - $d/logs/clirun.train > $out <<'P'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
DR, LN, RMSE, RMSD, COMB, LR, MODE, GRAN, TIME, SPEED = range(10)
b = [r for r in rows[1:] if r[GRAN] == "batch" and r[MODE] == "train"]
sp = [float(r[SPEED]) for r in b]
last = sp[len(sp) // 3:]                      # skip the first epoch (kernel warm-up, allocator growth)
last.sort()
print(f"train.py --synthetic 32,512,60 -l drmsd d512 nl6: {len(b)} batch rows, median speed over epochs 2-3 = "
      f"{last[len(last) // 2]:.0f} residues/s, p10 {last[len(last) // 10]:.0f}, p90 {last[9 * len(last) // 10]:.0f}")
t = [float(r[TIME]) for r in b]
n = len(b) // 3
print(f"wall clock per batch over epochs 2-3 (time column, includes checkpoint writes): {(t[-1] - t[n]) / (len(t) - 1 - n) * 1e3:.2f} ms")
P
tail -3 $d/stdout.txt >> $out
