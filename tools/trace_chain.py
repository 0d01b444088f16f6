#!/usr/bin/env python
"""Back-end chain of one step from a rocprofv3 kernel trace (stats_kernel_trace.csv): per-kernel gap / duration on the queue that
carries be_marg_kernel, between two consecutive marginalisation launches.    python tools/trace_chain.py trace.csv [step_from_end] [-v]
-v: every launch of the step in order (duration, gap before it)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open([a for a in sys.argv[1:] if a != "-v"][0])))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
qs = collections.defaultdict(list)
for r in rows:
    qs[r["Queue_Id"]].append(r)
verbose = "-v" in sys.argv
args = [a for a in sys.argv[1:] if a != "-v"]
back = int(args[1]) if len(args) > 1 else 2
for q, l in sorted(qs.items()):
    ms = [i for i, r in enumerate(l) if r["Kernel_Name"].startswith("be_marg")]
    if len(ms) < back + 2:
        continue
    seg = l[ms[-back - 1] + 1:ms[-back] + 1]
    prev = l[ms[-back - 1]]["e"]
    agg = collections.OrderedDict()
    gaps = 0.0
    for r in seg:
        name = r["Kernel_Name"].split("(")[0]
        gap = max(r["s"] - prev, 0) / 1e3
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1; a[1] += (r["e"] - r["s"]) / 1e3; a[2] += gap
        gaps += gap
        if verbose:
            print("      %-24s %8.1f us   gap %6.1f us" % (name, (r["e"] - r["s"]) / 1e3, gap))
        prev = r["e"]
    print("queue %s: step span %.3f ms, gaps %.3f ms" % (q, (seg[-1]["e"] - l[ms[-back - 1]]["e"]) / 1e6, gaps / 1e3))
    for name, a in agg.items():
        print("   %-24s x%-3d busy %8.1f us  (avg %6.1f)  gaps before %7.1f us" % (name, a[0], a[1], a[1] / a[0], a[2]))
