#!/usr/bin/env python
"""Where the literal marginalisation (vio_config.marg_exact = 1) spends its time: per sequence and frame the size m of the marginalised block, the
Jacobi sweeps (0 = the LDS-resident Householder + QL path) and the in-kernel ticks of the stages of marg_exact_finish (timers build).
    python tools/marg_exact_probe.py [--seqs 16] [--frames 24]"""
import os as _os
_os.environ.setdefault("VIO_HIP_LIB", "timers")
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vio_ct  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", type=int, default=16)
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--mode", type=int, default=1, help="vio_config.marg_exact: 1 = every eigen-decomposition literal, 2 = certified first inverse")
    a = ap.parse_args()
    P = vio_ct.pkg()
    L = P.lib()
    L.vio_debug_seq.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    cfg = P.canonical_config(marg_exact=a.mode)
    sc = vio_ct.synth_like(cfg)
    syn = P.Synth(sc)
    S, n_pre = a.seqs, 26
    F = n_pre + a.frames
    hw = cfg.height * cfg.width
    g = P.DeviceBuffer(F * S * hw)
    d = P.DeviceBuffer(F * S * hw * 2)
    times = vio_ct.frame_times(sc, F)
    for f in range(F):
        syn.render_device(S, 0, float(times[f]), g.at(f * S * hw), d.at(f * S * hw * 2))
    nimu = int(F / sc.cam_rate * sc.imu_rate) + 64
    b = P.VioBatch(cfg, S, imu_capacity=nimu + 64)
    imu = [syn.imu(s, nimu) for s in range(S)]
    b.push_imu_batch(np.stack([x[0] for x in imu]), np.stack([x[1] for x in imu]), np.stack([x[2] for x in imu]))
    rows = []
    dbg = np.zeros(16, np.int32)
    for f in range(F):
        b.feed(g.at(f * S * hw), d.at(f * S * hw * 2), np.full(S, times[f]), on_device=True)
        if f < 12:
            continue
        for s in range(S):
            L.vio_debug_seq(b.h, s, dbg.ctypes.data)
            if b.status(s).solver_flag == 1:
                rows.append((f, s, dbg[4], dbg[0], dbg[2], dbg[3] / 100.0, dbg[7] / 100.0, dbg[8] / 100.0, dbg[9] / 100.0, dbg[10] / 100.0))
    r = np.array(rows)
    print("frame seq m sweeps second_new | marg us | eig1 products eig2 prior (cumulative us)")
    for row in rows[:: max(1, len(rows) // 60)]:
        print("%3d %3d %4d %5d %d | %8.0f | %8.0f %8.0f %8.0f %8.0f" % row)
    old = r[r[:, 4] == 0]
    if len(old):
        print("MARGIN_OLD frames: m mean %.0f max %.0f; share with m <= 104: %.2f; marg us mean %.0f max %.0f" % (
            old[:, 2].mean(), old[:, 2].max(), (old[:, 2] <= 104).mean(), old[:, 5].mean(), old[:, 5].max()))
    print("all frames: marg us mean %.0f p90 %.0f max %.0f" % (r[:, 5].mean(), np.percentile(r[:, 5], 90), r[:, 5].max()))


if __name__ == "__main__":
    main()
