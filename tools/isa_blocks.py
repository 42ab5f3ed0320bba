#!/usr/bin/env python
"""Instruction mix per basic block of one kernel in a hipcc -save-temps assembly file (static counts; loops show up as
blocks that branch to themselves).

    python tools/isa_blocks.py <file.s> <mangled-name prefix> [min instructions per block]
"""
import re
import sys


def classify(ins):
    c = dict(valu=0, mfma=0, ds=0, vmem=0, salu=0, wait=0)
    for x in ins:
        op = x.split()[0]
        if op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith("ds_"):
            c["ds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
        elif op.startswith(("s_waitcnt", "s_barrier", "s_nop")):
            c["wait"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    return c


def main():
    path, prefix = sys.argv[1], sys.argv[2]
    floor = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(prefix) and ":" in l)
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], ["entry", []]
    for l in lines[start + 1:end]:
        t = l.strip()
        if re.match(r"^\.LBB\d+_\d+:", t):
            blocks.append(cur)
            cur = [t.split(":")[0], []]
        elif t and not t.startswith((".", ";", "//")):
            cur[1].append(t)
    blocks.append(cur)
    print("total", classify(sum((b[1] for b in blocks), [])))
    for name, ins in blocks:
        c = classify(ins)
        if sum(c.values()) >= floor:
            br = [x.split()[-1] for x in ins if x.startswith(("s_cbranch", "s_branch"))]
            loop = " LOOP" if name in br else ""
            print(f"{name:12s} {c} -> {br[-2:]}{loop}")


if __name__ == "__main__":
    main()
