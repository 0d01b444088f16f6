#!/usr/bin/env python
"""Canonical workload at a given batch size / tracker lag through bench.aux_rate (one JSON line).    python tools/lag_rate.py --seqs 128 --lag 0"""
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
    ap.add_argument("--seqs", type=int, default=128)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--lag", type=int, default=1)
    ap.add_argument("--marg-exact", type=int, default=0, help="vio_config.marg_exact (0 fast form, 1 literal, 2 certified literal)")
    a = ap.parse_args()
    import torch
    import bench
    import vio_ct
    P = vio_ct.pkg()
    if a.seqs >= 2:
        os.environ.setdefault("VIO_GROUP_SEQS", str(max(1, a.seqs // 2)))
    cfg = P.canonical_config(marg_exact=a.marg_exact) if a.marg_exact else P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    r = bench.aux_rate(P, vio_ct, torch, cfg, sc, torch.device("cuda", 0), a.seqs, cfg.window_size + 8, 6, a.steps, lag=a.lag)
    print(json.dumps(dict({k: r[k] for k in ("sequences_per_gpu", "tracker_lag", "frames_per_s", "ms_per_step", "valid")}, be_marg_ms=r["kernels_ms"]["be_marg"])))


if __name__ == "__main__":
    main()
