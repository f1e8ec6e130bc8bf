#!/usr/bin/env python3
"""Static instruction counts of tools/micro/lanesplit.hip's timed loops (the loop body = one link step), from the gfx950 assembly
hipcc writes with -save-temps: per kernel, the instructions between the two s_memtime reads, by class."""
import collections
import re
import sys

asm = open(sys.argv[1]).read()
for m in re.finditer(r"^(_Z\d+(scalar|dist)_kernelILi(\d)EEvPKfPfPyi):[^\n]*\n(.*?)\n\s*s_endpgm", asm, re.S | re.M):
    kind, joint, body = m.group(2), m.group(3), m.group(4)
    parts = body.split("s_memtime")
    loop = parts[1] if len(parts) >= 3 else ""
    cls = collections.Counter()
    for line in loop.splitlines():
        t = line.strip().split()
        if not t or t[0].startswith((";", ".")) or t[0].endswith(":"):
            continue
        op = t[0]
        if op.startswith("v_") and "dpp" in line:
            cls["VALU with a DPP operand" if not op.startswith("v_mov") else "v_mov_b32_dpp"] += 1
        elif op.startswith("v_cndmask"):
            cls["v_cndmask"] += 1
        elif op.startswith("v_"):
            cls["VALU (other)"] += 1
        elif op.startswith("s_"):
            cls["SALU / waits / branches"] += 1
        else:
            cls["other"] += 1
    valu = sum(v for k, v in cls.items() if not k.startswith(("SALU", "other")))
    print(f"joint {joint} {kind:6s}: {valu:4d} vector instructions per link step  " + ", ".join(f"{k} {v}" for k, v in sorted(cls.items())))
