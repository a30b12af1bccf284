"""Per-step kernel table from a `rocprofv3 --kernel-trace --stats` kernel_stats.csv of `bench.py --steps S --warmup W`
(the instrumented roofline pass repeats the S steps: 2 S + W steps per run).
usage: python profiles/tools/r02_step_table.py <kernel_stats.csv> <steps in the run>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = float(sys.argv[2])
tot = 0.0
print(f"{'kernel':<70} {'calls/step':>10} {'avg us':>9} {'ms/step':>9} {'%':>6}")
out = []
for r in rows:
    ms = float(r["TotalDurationNs"]) / nsteps / 1e6
    out.append((r["Name"][:70], float(r["Calls"]) / nsteps, float(r["AverageNs"]) / 1e3, ms))
    tot += ms
for name, c, avg, ms in sorted(out, key=lambda t: -t[3]):
    if ms / tot >= 0.001:
        print(f"{name:<70} {c:10.1f} {avg:9.1f} {ms:9.3f} {100 * ms / tot:6.2f}")
print(f"{'all kernels':<70} {'':>10} {'':>9} {tot:9.3f}")
