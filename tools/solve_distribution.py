#!/usr/bin/env python
"""Per-sequence duration of be_solve (in-kernel 100 MHz ticks, BeSeq.dbg[4]) and of be_marg (dbg[3]) on the bench workload: shows how far the
kernel time (= slowest sequence) sits above the mean.   python tools/solve_distribution.py [--seqs 128] [--frames 30]"""
import os as _os
_os.environ.setdefault("VIO_HIP_LIB", "timers")   # dbg[3] / dbg[4] are tick counts of the build with the phase timers compiled in (the default build has none)
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
    ap.add_argument("--seqs", type=int, default=128)
    ap.add_argument("--frames", type=int, default=30)
    a = ap.parse_args()
    P = vio_ct.pkg()
    L = P.lib()
    L.vio_debug_seq.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    syn = P.Synth(sc)
    S, n_pre = a.seqs, 20
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
    for f in range(F):
        b.feed(g.at(f * S * hw), d.at(f * S * hw * 2), np.full(S, times[f]), on_device=True)
        if f < n_pre:
            continue
        sol, mar, it, nr = [], [], [], []
        dbg = np.zeros(16, np.int32)
        for s in range(S):
            L.vio_debug_seq(b.h, s, dbg.ctypes.data)
            st = b.status(s)
            sol.append(dbg[4] / 100.0); mar.append(dbg[3] / 100.0); it.append(st.iterations); nr.append(st.n_residuals)
        sol, mar, it, nr = map(np.array, (sol, mar, it, nr))
        rows.append((sol.mean(), sol.max(), np.percentile(sol, 90), mar.mean(), mar.max(), it.mean(), it.max(), nr.mean(), nr.max(),
                     int(it[np.argmax(sol)]), int(nr[np.argmax(sol)])))
    r = np.array(rows)
    print("per frame (us): solve mean %.0f  p90 %.0f  max %.0f | marg mean %.0f max %.0f | iterations mean %.2f max %.0f | residuals mean %.0f max %.0f" % (
        r[:, 0].mean(), r[:, 2].mean(), r[:, 1].mean(), r[:, 3].mean(), r[:, 4].mean(), r[:, 5].mean(), r[:, 6].mean(), r[:, 7].mean(), r[:, 8].mean()))
    print("slowest sequence per frame: iterations %s" % r[:, 9].astype(int).tolist())
    print("                            residuals  %s" % r[:, 10].astype(int).tolist())


if __name__ == "__main__":
    main()
