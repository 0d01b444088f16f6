#!/usr/bin/env python
"""Reads a rocprofv3 kernel trace of bench.py under VIO_SOLVE_MODE=1 and prints, for the last frames, the wall span of the phased solve
(ps_setup .. ps_final), the summed kernel time inside it and the per-kernel averages when active (duration > 8 us)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"].split("(")[0], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
frames = []
cur = None
for n, a, b in ev:
    if n == "ps_setup_kernel":
        cur = [(n, a, b)]
    elif cur is not None:
        cur.append((n, a, b))
        if n == "ps_final_kernel":
            frames.append(cur); cur = None
for fr in frames[-6:]:
    ps = [x for x in fr if x[0].startswith("ps_")]
    span = (ps[-1][2] - ps[0][1]) / 1e3
    busy = sum(b - a for _, a, b in ps) / 1e3
    gaps = [(ps[i + 1][1] - ps[i][2]) / 1e3 for i in range(len(ps) - 1)]
    print("span %.0f us  kernels %.0f us  launches %d  mean gap %.1f us  max gap %.1f us" % (span, busy, len(ps), sum(gaps) / len(gaps), max(gaps)))
acc = collections.defaultdict(list)
for fr in frames[len(frames) // 2:]:
    for n, a, b in fr:
        if n.startswith("ps_") and b - a > 8000:
            acc[n].append((b - a) / 1e3)
for n, v in sorted(acc.items()):
    print("%-22s active launches/frame %.1f  mean %.1f us  max %.1f us" % (n, len(v) / (len(frames) - len(frames) // 2), sum(v) / len(v), max(v)))
fr = frames[-1]
print("last frame:", " ".join("%s:%.0f" % (n.replace("ps_", "").replace("_kernel", ""), (b - a) / 1e3) for n, a, b in fr if n.startswith("ps_")))
