"""Host-to-device bandwidth in the PATTERN of vio_feed's page-locked path (two stream groups, each uploading 64 grey + 64 depth images per step:
29.5 MB + 59 MB on its own copy stream), against the raw ceiling of tools/h2d_bandwidth.py -- which allocation / chunking reaches it.
    python tools/h2d_pattern.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vio_ct
P = vio_ct.pkg()
S, HW, steps = 128, 640 * 480, 20
dev = torch.device("cuda:0")
dg = torch.empty((2, S, HW), dtype=torch.uint8, device=dev); dd = torch.empty((2, S, HW), dtype=torch.int16, device=dev)


def run(name, hg, hd, nstreams, chunks=1):
    st = [torch.cuda.Stream() for _ in range(nstreams)]
    torch.cuda.synchronize(); t = time.perf_counter()
    for k in range(steps):
        p = k & 1
        for g in range(2):
            s0, s1 = g * 64, (g + 1) * 64
            for c in range(chunks):
                a, b = s0 + (s1 - s0) * c // chunks, s0 + (s1 - s0) * (c + 1) // chunks
                with torch.cuda.stream(st[(g * chunks + c) % nstreams]):
                    dg[p, a:b].copy_(hg[k % len(hg)][a:b], non_blocking=True)
                    dd[p, a:b].copy_(hd[k % len(hd)][a:b], non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("%-58s %5.1f GB/s  %.2f ms per step" % (name, steps * S * HW * 3 / dt / 1e9, dt / steps * 1e3), flush=True)


tg = [torch.empty((S, HW), dtype=torch.uint8).pin_memory() for _ in range(3)]
td = [torch.empty((S, HW), dtype=torch.int16).pin_memory() for _ in range(3)]
pa = [P.PinnedArray((S, HW), np.uint8) for _ in range(3)]; pd = [P.PinnedArray((S, HW), np.int16) for _ in range(3)]
vg = [torch.from_numpy(x.a) for x in pa]; vd = [torch.from_numpy(x.a) for x in pd]
print("vio_host_alloc memory is_pinned:", vg[0].is_pinned())
for name, hg, hd in (("torch pin_memory", tg, td), ("vio_host_alloc (hipHostMalloc default)", vg, vd)):
    run(name + ", 2 streams (one per group)", hg, hd, 2)
    run(name + ", 4 streams, 2 chunks per group", hg, hd, 4, 2)
    run(name + ", 8 streams, 4 chunks per group", hg, hd, 8, 4)
    run(name + ", 1 stream", hg, hd, 1)
