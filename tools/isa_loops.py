#!/usr/bin/env python3
"""Developer tool: instruction mix of every basic block of a kernel that issues MFMAs (from `hipcc -S --cuda-device-only` output).
usage: tools/isa_loops.py build/k_gemm.s _Z7k_tgemmILi4EdE"""
import re, sys
s = open(sys.argv[1]).read().splitlines()
start = next(i for i, l in enumerate(s) if l.startswith(sys.argv[2]) and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(s)) if s[i].startswith(".Lfunc_end"))
blocks, cur = [], ["entry", []]
blocks.append(cur)
for l in s[start + 1:end]:
    l = l.strip()
    if re.match(r"^\.LBB\d+_\d+:", l):
        cur = [l.split(":")[0], []]
        blocks.append(cur)
    elif l and not l.startswith(";") and not l.startswith("."):
        cur[1].append(l)
for name, ins in blocks:
    mf = sum(i.startswith("v_mfma") for i in ins)
    if not mf:
        continue
    cls = {}
    for i in ins:
        op = i.split()[0]
        key = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
               "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "wait/nop" if op in ("s_waitcnt", "s_nop") else
               "barrier" if op == "s_barrier" else "salu")
        cls[key] = cls.get(key, 0) + 1
    print(name, len(ins), cls)
    if len(sys.argv) > 3:
        ops = {}
        for i in ins:
            op = i.split()[0]
            if op.startswith("v_") and not op.startswith("v_mfma"):
                ops[op] = ops.get(op, 0) + 1
        print("   ", sorted(ops.items(), key=lambda t: -t[1]))
