"""Reads a rocprofv3 --kernel-trace --memory-copy-trace CSV pair of a bench.py run and prints, for the page-locked leg (the last host-to-device
copies of the run), how the uploads sit against the kernels: per copy start / duration / bytes / GB/s, and the busy time of kernels overlapping it.
    python tools/pcie_timeline.py <dir>"""
import csv, glob, sys
import numpy as np
root = sys.argv[1]
kt = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
mt = glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True)[0]
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(kt))]
KQ = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in csv.DictReader(open(kt))]
M = []
for r in csv.DictReader(open(mt)):
    M.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Name", "?")), r.get("Stream_Id", "?")))
print("columns of the copy trace:", list(csv.DictReader(open(mt)).fieldnames))
big = [m for m in M if (m[1] - m[0]) >= 100_000 and "HOST_TO_DEVICE" in m[2].upper().replace(" ", "_")]   # (this rocprofv3 writes no byte counts: the image uploads are the long ones)
print(len(M), "copies,", len(big), "large host-to-device copies")
big = big[-80:]
t0 = big[0][0]
ks = np.array([k[0] for k in K]); ke = np.array([k[1] for k in K])
for s, e, d, b in big[:24]:
    ov = np.clip(np.minimum(ke, e) - np.maximum(ks, s), 0, None).sum()
    print("stream %3s copy at %9.1f us  dur %8.1f us   kernel time overlapping it %8.1f us" % (b, (s - t0) / 1e3, (e - s) / 1e3, ov / 1e3))
span = big[-1][1] - big[0][0]
print("last %d image uploads span %.2f ms; sum of their durations %.2f ms (4 uploads = 118 MB per step)" % (len(big), span / 1e6, sum(e - s for s, e, *_ in big) / 1e6))
for st in sorted(set(b for *_, b in big)):
    ss = [s for s, e, d, b in big if b == st]
    print("stream", st, len(ss), "uploads, median spacing between consecutive ones %.1f us" % (np.median(np.diff(ss)) / 1e3 if len(ss) > 1 else 0))
# kernels of the same window by name: busy time
w0, w1 = big[0][0], big[-1][1]
import collections
busy = collections.Counter()
for s, e, k in K:
    if s >= w0 and e <= w1: busy[k] += e - s
print("kernel busy time inside the window (ms):", {k: round(v / 1e6, 2) for k, v in busy.most_common(8)})

print("back-end frames inside the window (be_ingest start -> be_marg end), by stream:")
ev = sorted([(s, e, k, q, st) for s, e, k, q, st in KQ if w0 <= s <= w0 + 20_000_000 and k in ("be_ingest_kernel", "be_marg_kernel", "fe_begin_kernel", "fe_add_kernel")])
for s, e, k, q, st in ev:
    print("  stream %3s queue %3s  %-18s %9.1f .. %9.1f us" % (st, q, k, (s - w0) / 1e3, (e - w0) / 1e3))
