#!/bin/bash
# SQ counters of the attention kernels (one --pmc pass per group, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_attn_pmc; mkdir -p $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/g$i -o p -- python profiles/tools/r02_attn_bench.py 2 > $out/g$i.log 2>&1
done
python - <<P
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/g*/**/p_counter_collection.csv", recursive=True) + glob.glob("$out/g*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn_" in k and "split" in k and r.get("Grid_Size", "") :
            acc[k.split("(")[0]][r["Counter_Name"]].append((float(r["Counter_Value"]), r["Grid_Size"]))
for k, cs in acc.items():
    print(k)
    for c, vs in sorted(cs.items()):
        big = [v for v, g in vs if g == max(g2 for _, g2 in vs)]
        print(f"   {c:28s} {sum(big)/len(big):16.0f}  (n={len(big)})")
P
