#!/usr/bin/env python3
"""How much of each kernel's time another kernel was running beside it, from a `rocprofv3 --kernel-trace` csv (Start_Timestamp /
End_Timestamp per dispatch).  python tools/trace_overlap.py <kernel_trace.csv> [skip_fraction]

Looks at the dispatches after the first `skip_fraction` (default 0.5) of the trace — the replayed graph, not the set-up — and prints, per
kernel name: dispatches, mean duration, the share of its time with at least one OTHER dispatch in flight, and over the window: the sum of
durations, the length of their union (the time the GPU had any kernel) and the wall span."""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
rows.sort()
rows = rows[int(len(rows) * skip):]


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:64]


# sweep: events sorted by time; while `active` >= 2 every active dispatch is overlapped
events = []
for i, (s, e, _) in enumerate(rows):
    events.append((s, 1, i))
    events.append((e, 0, i))
events.sort()
active, last = set(), None
overl = [0] * len(rows)
union = 0
for t, kind, i in events:
    if last is not None and active:
        dt = t - last
        union += dt
        if len(active) >= 2:
            for j in active:
                overl[j] += dt
    if kind:
        active.add(i)
    else:
        active.discard(i)
    last = t
per = defaultdict(lambda: [0, 0, 0])
for (s, e, n), o in zip(rows, overl):
    p = per[short(n)]
    p[0] += 1
    p[1] += e - s
    p[2] += o
total = sum(e - s for s, e, _ in rows)
span = max(e for _, e, _ in rows) - min(s for s, _, _ in rows)
print(f"{len(rows)} dispatches: sum of durations {total / 1e3:.0f} us, union {union / 1e3:.0f} us, wall span {span / 1e3:.0f} us "
      f"(sum / union = {total / max(union, 1):.2f} kernels in flight on average while any is; idle {100 * (1 - union / span):.1f} % of the span)")
print(f"{'kernel':64s} {'calls':>6s} {'mean us':>8s} {'beside another':>15s}")
for n, (c, d, o) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:64s} {c:6d} {d / c / 1e3:8.2f} {100 * o / max(d, 1):14.1f} %")
