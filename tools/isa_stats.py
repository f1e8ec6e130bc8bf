#!/usr/bin/env python3
"""Instruction-class census of one kernel in a hipcc -S listing: tools/isa_stats.py file.s kernel_substring"""
import re
import sys
from collections import Counter

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
ins = []
for l in lines[start + 1:end + 1]:
    l = l.strip()
    if not l or l.startswith((".", ";")) or l.endswith(":"):
        continue
    ins.append(l.split()[0])
c = Counter(ins)
g = Counter()
for k, v in c.items():
    if k.startswith("scratch_"): g["scratch"] += v
    elif k.startswith(("global_", "buffer_", "flat_")): g["vmem"] += v
    elif k.startswith("ds_"): g["lds"] += v
    elif k.startswith("s_load"): g["smem"] += v
    elif k.startswith("s_waitcnt"): g["waitcnt"] += v
    elif k.startswith(("s_cbranch", "s_branch")): g["branch"] += v
    elif k.startswith("s_"): g["salu"] += v
    elif k.startswith("v_accvgpr"): g["accvgpr_mov"] += v
    elif k.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos", "v_exp", "v_log")): g["trans"] += v
    elif k.startswith("v_"): g["valu"] += v
    else: g["other"] += v
print("kernel lines", start, "-", end, "static instructions:", len(ins))
print(dict(g))
print(c.most_common(30))
