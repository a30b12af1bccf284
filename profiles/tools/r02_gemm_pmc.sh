#!/bin/bash
# SQ / LDS / cache counters of the f16x2 GEMM kernel on one shape (one --pmc pass per group, --kernel-trace only)
# usage: r02_gemm_pmc.sh M N K
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r2_gemm_pmc_$1_$2_$3; mkdir -p $out
python profiles/tools/r02_gemm_one.py $1 $2 $3 > $out/time.txt 2>&1
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/g$i -o p -- python profiles/tools/r02_gemm_one.py $1 $2 $3 4 > $out/g$i.log 2>&1
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
