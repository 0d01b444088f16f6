#!/usr/bin/env python
"""Diagnostic: per-frame HIP-vs-oracle divergence of one synthetic sequence (positions, solver statistics, landmark counts).
    python tools/parity_trace.py --seq 704 --frames 300 > trace.txt
Test infrastructure (drives the oracle); not part of the product path."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vio_ct  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seq", type=int, default=704)
    ap.add_argument("--frames", type=int, default=300)
    a = ap.parse_args()
    P = vio_ct.pkg()
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    syn = P.Synth(sc)
    n = a.frames
    times = vio_ct.frame_times(sc, n)
    ti, ai, gi = syn.imu(a.seq, int(n / sc.cam_rate * sc.imu_rate) + 64)
    o = vio_ct.OraclePipeline(cfg)
    b = P.VioBatch(cfg, 1, imu_capacity=len(ti) + 64)
    o.push_imu(ti, ai, gi)
    b.push_imu(0, ti, ai, gi)
    W = cfg.window_size
    print("# f dP_max dP_newest it_o it_h succ_o succ_h cost0_o cost0_h cost1_o cost1_h nres_o nres_h nlm_o nlm_h marg_o marg_h dBa dBg")
    for f in range(n):
        g, d = syn.render_host(a.seq, float(times[f]))
        o.feed(g, d, float(times[f]))
        b.feed(g[None], d[None], [times[f]])
        so, sh = o.status(), b.status(0)
        wo, wh = o.window(), b.window(0)
        dp = np.abs(wo[:, :3] - wh[:, :3])
        print("%3d %.3e %.3e %d %d %d %d %.9f %.9f %.9f %.9f %d %d %d %d %d %d %.2e %.2e" % (
            f, dp.max(), dp[W].max(), so["iterations"], sh.iterations, so["successful_steps"], sh.successful_steps, so["initial_cost"], sh.initial_cost,
            so["final_cost"], sh.final_cost, so["n_residuals"], sh.n_residuals, so["n_landmarks"], sh.n_landmarks, so["marginalization_flag"],
            sh.marginalization_flag, np.abs(wo[:, 10:13] - wh[:, 10:13]).max(), np.abs(wo[:, 13:16] - wh[:, 13:16]).max()))


if __name__ == "__main__":
    main()
