#!/usr/bin/env python
"""Front-end alone: per-kernel HIP-event times of stream group 0 and in-kernel durations per sequence, with the estimator idle.
    [VIO_GROUP_SEQS=64] [VIO_FE_CUS=64] python tools/fe_profile.py --seqs 128 --frames 30 [--lag 1]
The window is first filled through vio_feed (so that track counts / ages are the steady-state ones), then `frames` frames go through
vio_track only.  Prints the mean per-kernel time, the wall time per frame of the track loop and the distribution over sequences of
the time one workgroup spends in fe_select / fe_add (the one-workgroup-per-sequence kernels)."""
import os as _os
_os.environ.setdefault("VIO_HIP_LIB", "timers")   # the build with the phase timers compiled in (default build has none)
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vio_ct  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", type=int, default=128)
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--lag", type=int, default=1)
    a = ap.parse_args()
    P = vio_ct.pkg()
    L = P.lib()
    L.vio_debug_phases.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.vio_debug_fe_ticks.argtypes = [C.c_void_p, C.c_void_p]
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    syn = P.Synth(sc)
    S, n_pre = a.seqs, cfg.window_size + 8
    F = n_pre + a.frames
    hw = cfg.height * cfg.width
    g = P.DeviceBuffer(F * S * hw)
    d = P.DeviceBuffer(F * S * hw * 2)
    times = vio_ct.frame_times(sc, F)
    for f in range(F):
        syn.render_device(S, 0, float(times[f]), g.at(f * S * hw), d.at(f * S * hw * 2))
    nimu = int(F / sc.cam_rate * sc.imu_rate) + 64
    b = P.VioBatch(cfg, S, imu_capacity=nimu + 64)
    if a.lag:
        b.set_tracker_lag(1)
    imu = [syn.imu(s, nimu) for s in range(S)]
    b.push_imu_batch(np.stack([x[0] for x in imu]), np.stack([x[1] for x in imu]), np.stack([x[2] for x in imu]))
    for f in range(n_pre):
        b.feed(g.at(f * S * hw), d.at(f * S * hw * 2), np.full(S, times[f]), on_device=True)
    b.sync()
    L.vio_debug_phases(b.h, None, 1)
    b.profile_begin(a.frames)
    ticks = []
    t0 = time.perf_counter()
    for f in range(n_pre, F):
        b.track(g.at(f * S * hw), np.full(S, times[f]), on_device=True)
        if f % 8 == 0:   # sample the per-sequence durations now and then (synchronises: not part of the wall figure's steady state)
            o = np.zeros((S, 4), np.float32)
            L.vio_debug_fe_ticks(b.h, o.ctypes.data)
            ticks.append(o)
    b.sync()
    wall = (time.perf_counter() - t0) / a.frames * 1e3
    n, k = b.profile_end()
    print("S = %d, groups of %s, FE CUs %s, lag %d: %.3f ms wall per track-only frame (%d profiled)" % (
        S, os.environ.get("VIO_GROUP_SEQS", "all"), os.environ.get("VIO_FE_CUS", "default"), a.lag, wall, n))
    print("group 0 kernels [ms]: " + "  ".join("%s %.3f" % (kk, v) for kk, v in k.items() if kk.startswith("fe_")))
    print("group 0 front-end sum %.3f ms" % sum(v for kk, v in k.items() if kk.startswith("fe_")))
    T = np.concatenate(ticks) / 100.0   # us
    for col, name in ((0, "fe_select"), (1, "fe_add")):
        v = T[:, col]
        print("%s workgroup time over sequences [us]: mean %.1f  median %.1f  p90 %.1f  max %.1f" % (name, v.mean(), np.median(v), np.percentile(v, 90), v.max()))
    print("deficit cells per sequence: mean %.1f max %d; ransac niters mean %.1f max %d" % (T[:, 2].mean() * 100, int(T[:, 2].max() * 100), T[:, 3].mean() * 100, int(T[:, 3].max() * 100)))
    out = np.zeros(128, np.float32)
    L.vio_debug_phases(b.h, out.ctypes.data, 0)
    us = out / 100.0 / a.frames
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import phase_profile
    for kk in sorted(phase_profile.NAMES):
        if kk >= 64 and us[kk] > 0:
            print("%3d %-28s %8.1f us/frame (sequence 0)" % (kk, phase_profile.NAMES[kk], us[kk]))


if __name__ == "__main__":
    main()
