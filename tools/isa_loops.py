#!/usr/bin/env python3
"""Instruction histogram of the loops of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).
usage: tools/isa_loops.py <file.s> <kernel-name-substring>
A 'loop' is a backward branch: every label that some later s_cbranch / s_branch jumps back to; its body = label .. branch."""
import re
import sys
from collections import Counter

txt = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]
start = next(i for i, l in enumerate(txt) if (": ;" in l or l.endswith(":")) and key in l.split(":")[0] and not l.startswith("."))
end = next(i for i in range(start, len(txt)) if txt[i].startswith(".Lfunc_end"))
body = txt[start:end + 1]
labels = {}
ins = []
for l in body:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        m = re.match(r"^(\.LBB\S+):", t)
        if m:
            labels[m.group(1)] = len(ins)
        continue
    ins.append(t)
print(f"{body[0][:80]}  instructions: {len(ins)}")
loops = []
for i, t in enumerate(ins):
    m = re.match(r"s_c?branch\S*\s+(\.LBB\S+)", t)
    if m and m.group(1) in labels and labels[m.group(1)] <= i:
        loops.append((labels[m.group(1)], i, m.group(1)))
for a, b, name in sorted(loops, key=lambda x: x[0] - x[1])[:8]:
    seg = ins[a:b + 1]
    c = Counter(s.split()[0] for s in seg)
    cls = Counter()
    for op, n in c.items():
        k = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith("s_waitcnt") and not op.startswith("s_cbranch") and not op.startswith("s_branch")
             else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "wait" if op.startswith("s_waitcnt") else "branch")
        cls[k] += n
    print(f"loop {name}: {len(seg)} instr  {dict(cls)}")
    print("   top:", ", ".join(f"{op}x{n}" for op, n in c.most_common(14)))
