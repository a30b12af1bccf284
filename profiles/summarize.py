#!/usr/bin/env python3
"""Turn raw rocprofv3 output (gpurun_out/...) into the small summaries kept under profiles/.

  python profiles/summarize.py stats  <kernel_stats.csv> <steps_in_run>         -> per-step table on stdout
  python profiles/summarize.py traffic <fetch_counter_collection.csv> <write_counter_collection.csv> \
         <algorithmic_bytes_per_launch> <out.json>                              -> HBM bytes per GEMM launch
  python profiles/summarize.py step_traffic <fetch csv> <write csv> <steps>     -> HBM bytes per step by kernel

FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane
coalesced reads as 64 bytes (MI355X_MICROARCH.md, HBM section), hence the x2 on the fetch side; WRITE_SIZE is taken
as is.  The two counters are collected in separate passes (one --pmc each, with --kernel-trace only).
"""
import csv
import json
import sys

KERNEL = ("_mfma_kernel", "gemm_hp_kernel", "gemm_hp3_kernel")   # gemm_bf16x3_mfma_kernel<.., NPROD, ..> (NPROD = 3: f16x2, 6: bf16x3), gemm_f32_mfma_kernel, and the LDS-DMA kernel


def stats(path, steps):
    rows = list(csv.DictReader(open(path)))
    if str(steps) == "auto":      # steps the process ran in all (warm-up, timed, kernel-timing passes): the loss kernel runs once per step
        steps = max(int(r["Calls"]) for r in rows if "drmsd_tri_kernel" in r["Name"])
    steps = int(steps)
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"{'kernel':70s} {'calls/step':>10s} {'avg us':>9s} {'ms/step':>9s} {'%':>6s}")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
        t = float(r["TotalDurationNs"])
        print(f"{r['Name'][:70]:70s} {int(r['Calls']) / steps:10.1f} {float(r['AverageNs']) / 1e3:9.1f} "
              f"{t / steps / 1e6:9.3f} {100 * t / total:6.2f}")
    print(f"{'all kernels':70s} {'':10s} {'':9s} {total / steps / 1e6:9.3f}")


def counter_avg(path, name, kernel=KERNEL):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if "gemm_" in r["Kernel_Name"] and any(k in r["Kernel_Name"] for k in ((kernel,) if isinstance(kernel, str) else kernel))
            and r["Counter_Name"] == name]
    return (sum(vals) / len(vals), len(vals)) if vals else (0.0, 0)


def traffic(fetch_csv, write_csv, algorithmic, out):
    f, n = counter_avg(fetch_csv, "FETCH_SIZE")
    w, _ = counter_avg(write_csv, "WRITE_SIZE")
    hbm = (2.0 * f + w) * 1024.0
    # the pass in front of an f16x2 product (row scales of both operands): its traffic per GEMM launch that has one
    fs, ns = counter_avg(fetch_csv, "FETCH_SIZE", "row_scale_kernel")
    ws, _ = counter_avg(write_csv, "WRITE_SIZE", "row_scale_kernel")
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from protein_transformer_amd.build import gemm_source_digest
    rec = {
        "gemm_source_digest": gemm_source_digest(),
        "kernel": "gemm_*" + " | ".join(KERNEL), "launches_profiled": n, "fetch_size_kb_avg": f, "write_size_kb_avg": w,
        "fetch_correction": "x2: on gfx950 FETCH_SIZE tallies 128-B requests as 64 B for 16-B/lane coalesced reads "
                            "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncorrected",
        "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": float(algorithmic),
        "ratio": hbm / float(algorithmic),
        "row_scale_pass": {"kernel": "gemm_row_scale_kernel", "launches_profiled": ns,
                           "hbm_bytes_per_launch": (2.0 * fs + ws) * 1024.0,
                           "note": "reads both operands of an f16x2 product once more; not part of hbm_bytes_per_launch"},
        "note": "counted at the L2's fabric side, so Infinity-Cache hits are included",
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- python bench.py --steps 2 --warmup 1 "
                   "--no-cpu-baseline --no-kernel-timing ; same with --pmc WRITE_SIZE (separate passes)",
    }
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


def step_traffic(fetch_csv, write_csv, steps):
    """HBM-side bytes per step by kernel from the two PMC passes (2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes)."""
    import collections
    import re
    tot, calls = collections.defaultdict(float), collections.Counter()
    if str(steps) == "auto":   # the loss kernel runs once per step, whatever passes the profiled command made
        steps = sum(1 for r in csv.DictReader(open(fetch_csv)) if r["Counter_Name"] == "FETCH_SIZE" and "drmsd_tri_kernel" in r["Kernel_Name"])
    steps = int(steps)
    for path, name, mult in ((fetch_csv, "FETCH_SIZE", 2.0), (write_csv, "WRITE_SIZE", 1.0)):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != name:
                continue
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            k = re.sub(r"^void ", "", k).split("(")[0].split("<")[0]
            tot[k] += mult * float(r["Counter_Value"]) * 1024.0
            if name == "FETCH_SIZE":
                calls[k] += 1
    print(f"# HBM-side bytes per step by kernel (2 x FETCH_SIZE + WRITE_SIZE, separate PMC passes; {steps} steps profiled)")
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:20]:
        print(f"{v / steps / 1e6:10.1f} MB/step {calls[k] / steps:7.1f} calls/step  {k}")
    print(f"total {sum(tot.values()) / steps / 1e9:.3f} GB/step")


def sq(paths):
    """Where the wave cycles of every kernel go: SQ counters of one or more PMC passes (rocprofv3 --pmc SQ_..., separate
    runs of the same command), summed over the launches of a kernel, as fractions of SQ_WAVE_CYCLES.  Per the guide's PMC
    table: WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES; these
    count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES count cycles."""
    import collections
    import re
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for path in paths:
        seen = set()
        for r in csv.DictReader(open(path)):
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            k = re.sub(r"^void ", "", k).split("(")[0]
            k = re.sub(r"^(pt\w+::)", "", k)[:58]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r.get("Dispatch_Id"))
            if path == paths[0] and key not in seen:
                seen.add(key)
                calls[k] += 1
    names = sorted({c for v in tot.values() for c in v})
    cols = [c for c in names if c != "SQ_WAVE_CYCLES"]
    print("# per kernel: launches, SQ_WAVE_CYCLES (quad-cycles, summed over waves), then each counter / SQ_WAVE_CYCLES")
    print(f"{'kernel':58s} {'n':>5s} {'wave_cyc':>10s} " + " ".join(f"{c.replace('SQ_', '')[:14]:>14s}" for c in cols))
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0))[:16]:
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        if wc <= 0:
            continue
        print(f"{k:58s} {calls[k]:5d} {wc:10.3e} " + " ".join(f"{v.get(c, 0.0) / wc:14.3f}" for c in cols))


if __name__ == "__main__":
    if sys.argv[1] == "sq":
        sq(sys.argv[2:])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "traffic":
        traffic(*sys.argv[2:6])
    elif sys.argv[1] == "step_traffic":
        step_traffic(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        sys.exit(__doc__)
