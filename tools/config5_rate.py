#!/usr/bin/env python
"""BASELINE configs[4] (64 sequences of 1280x720 / 300 features / W = 20) on its own: frames/s and per-kernel HIP-event timings.
    python tools/config5_rate.py [--seqs 64] [--steps 20]          (wrap in rocprofv3 --kernel-trace --stats for the ps_* breakdown)"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", type=int, default=64)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--lag", type=int, default=1)
    a = ap.parse_args()
    import torch
    import bench
    import vio_ct
    P = vio_ct.pkg()
    os.environ.setdefault("VIO_GROUP_SEQS", str(max(1, a.seqs // 2)))
    cfg = bench.config5(P)
    sc = vio_ct.synth_like(cfg)
    r = bench.aux_rate(P, vio_ct, torch, cfg, sc, torch.device("cuda", 0), a.seqs, cfg.window_size + 8, 6, a.steps, lag=a.lag)
    print(json.dumps(r))


if __name__ == "__main__":
    main()
