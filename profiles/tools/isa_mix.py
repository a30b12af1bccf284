#!/usr/bin/env python
"""Static instruction mix of a kernel's loops from the compiler's assembly (hipcc -S --cuda-device-only): what a wavefront
ISSUES per trip of each basic-block loop - the quantity the split-arithmetic kernels are bound by at the package power cap
(profiles/r04/NOTES.md section 7: only removing instructions or bytes pays).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -x hip -S --cuda-device-only -o k.s csrc/<file>.hip
    python profiles/tools/isa_mix.py k.s <substring of the mangled kernel name> [--blocks]

Prints, for the whole kernel and for every backward-branch loop body (label .. branch back to it): instruction counts by class
(MFMA, VALU split into transcendental / packed / conversions / integer-bit / fp32 / moves / DPP-lane, LDS, VMEM, SALU, waits)
and an issue-cycle estimate per trip (wave64 on a SIMD16: 4 cycles per full-rate VALU, 16 per transcendental, 8 passes x 4
= 32 per v_mfma_f32_32x32x16, 16 per 16x16x32)."""
import re
import sys
from collections import Counter

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("ds_", )):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop") or op.startswith("s_sleep"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        if op.startswith(TRANS):
            return "valu.trans"
        if op.startswith("v_pk_"):
            return "valu.packed"
        if op.startswith("v_cvt_"):
            return "valu.cvt"
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane", "v_permlane", "v_mov_b32_dpp", "v_bpermute")) or "_dpp" in op:
            return "valu.lane"
        if op.startswith(("v_mov_", "v_accvgpr")):
            return "valu.mov"
        if op.startswith(("v_cmp", "v_cndmask")):
            return "valu.cmpsel"
        if re.match(r"v_(and|or|xor|not|lshl|lshr|ashr|bfe|bfi|add_u|sub_u|add_co|addc|sub_co|subb|mad_u|mul_u|mul_lo|mul_hi|add3|lshl_add|lshl_or|and_or|or3|xad|add_lshl|perm|alignbit|min_u|max_u|min_i|max_i|sad|mad_i|mul_i|add_nc|sub_nc|subrev|bcnt|mbcnt|ffb|sat_pk)", op):
            return "valu.int"
        return "valu.f32"
    return "other"


def cycles(cnt, mfma_ops):
    c = 0.0
    for k, v in cnt.items():
        if k == "valu.trans":
            c += 16 * v
        elif k.startswith("valu"):
            c += 4 * v
    m = 0.0
    for op, v in mfma_ops.items():
        m += v * (32 if "32x32" in op else 16)
    return c, m


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w*:", l) and key in l.split(":")[0]:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    # the last s_endpgm of the function: scan to .Lfunc_end
    for i in range(start, len(lines)):
        if lines[i].startswith(".Lfunc_end"):
            end = i
            break
    body = lines[start:end]
    labels, insts = {}, []
    for l in body:
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) and not re.match(r"^\.LBB\d+_\d+:", t):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if re.match(r"^_Z\w*:", t):
            continue
        op = t.split()[0]
        insts.append((op, t))

    def mix(lo, hi):
        cnt, mf, ops = Counter(), Counter(), Counter()
        for op, t in insts[lo:hi]:
            c = classify(op)
            cnt[c] += 1
            ops[op] += 1
            if c == "mfma":
                mf[op] += 1
        return cnt, mf, ops

    def show(name, lo, hi, detail=False):
        cnt, mf, ops = mix(lo, hi)
        valu = sum(v for k, v in cnt.items() if k.startswith("valu"))
        vc, mc = cycles(cnt, mf)
        print(f"{name}: {hi - lo} instructions; MFMA {cnt['mfma']} ({mc:.0f} cyc)  VALU {valu} ({vc:.0f} cyc)  LDS {cnt['lds']}  "
              f"VMEM {cnt['vmem']}  SALU {cnt['salu']}  wait {cnt['wait']}")
        print("    " + "  ".join(f"{k[5:]} {v}" for k, v in sorted(cnt.items()) if k.startswith("valu.")))
        if detail:
            top = [f"{o} {n}" for o, n in ops.most_common(40) if o.startswith("v_") and not o.startswith("v_mfma")]
            print("    " + ", ".join(top))

    nums = [a for a in sys.argv[3:] if a.isdigit()]
    nloops = int(nums[0]) if nums else 6
    show("whole kernel", 0, len(insts))
    loops = []
    for i, (op, t) in enumerate(insts):
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i + 1, tgt))
    for lo, hi, tgt in sorted(set(loops), key=lambda x: x[0] - x[1])[:nloops]:
        show(f"loop {tgt} [{lo}, {hi})", lo, hi, detail="--blocks" in sys.argv)


if __name__ == "__main__":
    main()
