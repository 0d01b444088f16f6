#!/usr/bin/env python
"""Replay a rosbag-free RGB-D + IMU recording through the MI355X hot path and write the reference's result CSV.

    python tools/replay.py --config <vio.yaml> --data <dir with rgb.txt depth.txt imu.txt> --out vins_result.csv [--gt gt.txt]

The recording layout is described in vins-rgbd-fast_amd/dataio.py (RgbdImuDirectory).  --gt: ``stamp x y z ...`` ground truth
(TUM format) for an ATE report."""
import argparse
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--data", required=True)
    ap.add_argument("--out", default="vins_result.csv")
    ap.add_argument("--gt", default=None)
    ap.add_argument("--lenient", action="store_true", help="warn instead of failing on settings outside the built hot path")
    a = ap.parse_args()
    P = importlib.import_module("vins-rgbd-fast_amd")
    io = importlib.import_module("vins-rgbd-fast_amd.dataio")
    cfg, extra = io.config_from_yaml(a.config, P, strict=not a.lenient)
    for n in extra["notes"]:
        print("note:", n, file=sys.stderr)
    rec = io.RgbdImuDirectory(a.data)
    b = P.VioBatch(cfg, 1, imu_capacity=1 << 15)
    rows = io.replay(b, rec, a.out, freq=extra["freq"], frontend_freq=extra["frontend_freq"])  # freq / frontend_freq: estimator_nodelet.cpp:264-286
    print("%d frames, %d odometry rows -> %s" % (len(rec), len(rows), a.out))
    if a.gt and len(rows) > 3:
        gt = np.loadtxt(a.gt, comments="#", ndmin=2)
        gp = np.array([gt[np.argmin(np.abs(gt[:, 0] - t)), 1:4] for t in rows[:, 0]])
        print("ATE rmse %.4f m over %d poses" % (io.ate_rmse(rows[:, 1:4], gp), len(rows)))


if __name__ == "__main__":
    main()
