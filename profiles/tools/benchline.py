"""stdin: the JSON line of bench.py; prints `label ms_per_step value [extra keys]`.  python benchline.py LABEL [key ...]"""
import json
import sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d["value"], *[f"{k}={d.get(k)}" for k in sys.argv[2:]])
