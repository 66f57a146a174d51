#!/usr/bin/env python3
"""Instruction mix of one kernel in hipcc's -S output, per basic block (blocks of >= `min` instructions).

    hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -S --cuda-device-only -o k.s csrc/tclip_kernels.hip
    python scripts/isa_blocks.py k.s '_ZN5tclip9k_mm_liveILi16ELi4ELb0ELi1ELi64EEEvNS_6MMArgsE' [min]
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_rcp") or op.startswith("v_rsq") or op.startswith("v_sqrt") or op.startswith("v_log") or op.startswith("v_exp"):
        return "trans"
    if op.startswith("v_") and ("_f64" in op):
        return "f64"
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith("v_cmp") or op.startswith("v_cndmask"):
        return "cmp/sel"
    if op.startswith("v_mbcnt") or op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"):
        return "lane"
    if op.startswith("v_") and re.search(r"_(f32|f16)", op):
        return "f32"
    if op.startswith("v_"):
        return "vint"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, sym = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur, name = [], Counter(), "entry"
    total = Counter()
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            blocks.append((name, cur))
            cur, name = Counter(), m.group(1) + " " + (l.split(";")[-1].strip() if ";" in l else "")
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\b", l)
        if m and not l.strip().startswith("."):
            c = classify(m.group(1))
            cur[c] += 1
            total[c] += 1
    blocks.append((name, cur))
    keys = ["f32", "pk", "trans", "f64", "cmp/sel", "vint", "lane", "lds", "scratch", "vmem", "salu", "wait", "branch", "barrier"]
    print("%-60s %6s " % ("block", "n") + " ".join("%7s" % k for k in keys))
    for name, c in blocks:
        n = sum(c.values())
        if n >= min_n:
            print("%-60s %6d " % (name[:60], n) + " ".join("%7d" % c[k] for k in keys))
    print("%-60s %6d " % ("TOTAL", sum(total.values())) + " ".join("%7d" % total[k] for k in keys))
    for l in lines[end:end + 60]:
        if re.match(r"^; (codeLenInByte|NumVgprs|ScratchSize|LDSByteSize|Occupancy|TotalNumSgprs)", l):
            print(l)


if __name__ == "__main__":
    main()
