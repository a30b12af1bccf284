#!/bin/bash
# SQ / LDS counters of the f16x2 staging GEMM on one shape and layout (one --pmc pass per group, --kernel-trace only)
# usage: r03_gemm_pmc.sh {nn|dw} M N K
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r3_gemm_pmc_$1_$2_$3_$4; mkdir -p $out
python profiles/tools/r03_gemm_one.py $1 $2 $3 $4 > $out/time.txt 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/g$i -o p -- python profiles/tools/r03_gemm_one.py $1 $2 $3 $4 4 > $out/g$i.log 2>&1
done
python - <<P
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$out/g*/**/p_counter_collection.csv", recursive=True) + glob.glob("$out/g*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "_mfma_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(open("$out/time.txt").read().strip())
for c, vs in sorted(acc.items()):
    print(f"   {c:36s} {sum(vs)/len(vs):18.0f}  (n={len(vs)})")
P
rm -rf $out/g*
