#!/usr/bin/env python
"""In-kernel phase timers of sequence 0 (vio_debug_phases: thread 0 accumulates 100 MHz ticks per phase) on the bench workload.
    python tools/phase_profile.py [--seqs 128] [--frames 40]
Prints microseconds per frame for every phase slot of be_solve / be_marg (see the PH(k) markers in be_kernels.hip)."""
import os as _os
_os.environ.setdefault("VIO_HIP_LIB", "timers")   # the build with the phase timers compiled in (default build has none)
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vio_ct  # noqa: E402

NAMES = {0: "vector2double", 1: "lm indexing", 2: "pair lists", 4: "evaluate(first/accepted)", 5: "assemble", 6: "prepare_point", 7: "cauchy", 8: "schur",
         9: "cholesky", 10: "tri solves", 11: "dogleg/model", 12: "candidate evaluate", 13: "accept/bookkeeping",
         16: "marg prior", 17: "marg imu", 18: "marg proj eval", 19: "marg lm rows", 20: "marg frame blocks+rank update", 21: "marg 15x15 + reduce",
         22: "marg new prior + c0", 23: "marg keep data", 24: "  marg frame blocks (mfma)", 25: "  marg scatter", 27: "finish consistency check",
         48: "ser prepare_point", 49: "ser GN rhs (Hpl matvec)", 50: "ser tile load", 51: "ser cholesky", 52: "ser solves + back-subst", 53: "ser cauchy (H, Hpl matvec)",
         54: "ser dogleg/model", 55: "ser candidate", 56: "  ser chol_solve_tiles", 57: "  ser finite check", 58: "  ser y vectors + Hpl matvec",
         59: "  chol panel (wave 0 view)", 60: "  chol diag + trailing (wave 0)", 61: "  chol barrier wait",
         62: "eval block 0 (prior)", 63: "eval block 1 (imu)", 14: "eval block 3 (256 projections)", 15: "eval accept",
         28: "finish slide states", 29: "finish slide landmarks", 30: "finish removeFailures + odom",
         32: "asm_a pair (0,1) item", 33: "asm_a pair (0,W) item", 34: "asm_a imu item 0", 35: "asm_a landmark rows item 0", 36: "asm element sums", 37: "asm lm rows",
         36: "  marg 21a: 15x15 inverse", 37: "  marg 21b: T1 / Amr staging", 38: "  marg 22a: prior_H / prior_r stores", 39: "  marg 22b: tile fill",
         45: "  eval b0: X to LDS", 40: "  eval b0: prior dx", 44: "  eval b0: A dx partials", 46: "  eval b1: X + headers to LDS",
         100: "setup: init extras + vector2double", 101: "setup: landmark indexing (2 scans)", 102: "setup: relo + residual list", 103: "setup: pair counts",
         104: "setup: pair lists", 106: "ingest: hash + matching", 107: "ingest: new landmarks", 108: "ingest: parallax", 109: "ingest: IMU pre-integration",
         110: "ingest: triangulate", 111: "  imu: interval + load", 112: "  imu: stage + state recursions", 113: "  imu: F / V of the samples", 114: "  imu: J / P recursion",
         # front-end (fe_kernels.hip FE_PH markers)
         64: "select load + status cull", 65: "select lift (F input)", 66: "  ransac 7-point hypotheses", 67: "  ransac inlier counts", 68: "  ransac serial best/iters",
         69: "  ransac final inliers", 70: "select F compaction", 71: "select rank sort", 72: "select greedy setMask", 73: "select counts + stores",
         80: "add cell mask filter", 81: "add cell scan/compact", 82: "add cell top-k", 83: "add cell addPoints", 84: "add (tail of cells)",
         85: "add undistort + velocity", 86: "add ids", 87: "add packaging", 90: "lk: block 0 total", 91: "lk: launches counted",
         92: "lk: iterations x100 (block 0)", 93: "lk: levels x100 (block 0)", 94: "lk: setup per frame (block 0)", 95: "lk: iterations per frame (block 0)",
         96: "lk:   setup until staged (cumulative)", 97: "lk:   setup until derivatives (cumulative)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", type=int, default=128)
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--config5", action="store_true", help="BASELINE configs[4] shape: 1280x720, 300 features, W = 20")
    a = ap.parse_args()
    P = vio_ct.pkg()
    L = P.lib()
    L.vio_debug_phases.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    cfg = P.canonical_config()
    if a.config5:
        cfg = P.canonical_config(width=1280, height=720, max_cnt=300, window_size=20, grid_rows=7, grid_cols=8, max_landmarks=2048,
                                 fx=604.5821781259577 * 2, fy=604.2544712985845 * 1.5, cx=321.2638233484251 * 2, cy=239.70969315130674 * 1.5)
    sc = vio_ct.synth_like(cfg)
    syn = P.Synth(sc)
    S, n_pre = a.seqs, cfg.window_size + 10
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
    for f in range(n_pre):
        b.feed(g.at(f * S * hw), d.at(f * S * hw * 2), np.full(S, times[f]), on_device=True)
    L.vio_debug_phases(b.h, None, 1)
    it0 = b.status(0).iterations_total
    for f in range(n_pre, F):
        b.feed(g.at(f * S * hw), d.at(f * S * hw * 2), np.full(S, times[f]), on_device=True)
    out = np.zeros(128, np.float32)
    L.vio_debug_phases(b.h, out.ctypes.data, 0)
    st = b.status(0)
    print("sequence 0: %.2f solver iterations per frame over %d frames" % ((st.iterations_total - it0) / a.frames, a.frames))
    us = out / 100.0 / a.frames
    for k in sorted(NAMES):
        if us[k] > 0:
            print("%3d %-28s %8.1f us/frame" % (k, NAMES[k], us[k]))
    print("MARGIN_OLD frames of sequence 0: %d of %d; direct (Cholesky) pseudo-inverse in %d marginalisations" % (int(out[31]), a.frames, int(out[26])))
    print("solve top-level sum %.1f us, marg sum %.1f us" % (sum(us[k] for k in (0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13)), sum(us[16:24])))
    print("ps_serial sum %.1f us/frame" % float(sum(us[48:56]) + us[56] + us[57] + us[58]))
    print("fe_select sum %.1f us/frame, fe_add sum %.1f us/frame" % (float(sum(us[64:74])), float(sum(us[80:88]))))


if __name__ == "__main__":
    main()
