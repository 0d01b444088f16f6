#!/usr/bin/env python
"""Flip census: the first DECISION at which the HIP path and the oracle part ways, per sequence.
    python tools/flip_census.py [--lag 0|1] [--seqs 128] --out gpurun_out/flip_census_lag0.json
128 sequences x 300 frames of the canonical workload through vio_feed with the per-frame status read back every frame, against the
oracle's per-frame decisions (tests/golden/oracle_decisions_300.npz, written by tests/oracle_control.py from the same oracle runs as the
control experiment) and its positions (tests/golden/oracle_ate_300[_lag1].npz).  Same classification as the oracle-vs-oracle control
(tests/oracle_control.first_flip), so the two tables can be read side by side.  Test infrastructure: compares with the oracle."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vio_ct  # noqa: E402
import parity_long  # noqa: E402
import oracle_control  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lag", type=int, default=0)
    ap.add_argument("--seqs", type=int, default=128)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "flip_census.json"))
    a = ap.parse_args()
    P = vio_ct.pkg()
    fx = np.load(os.path.join(ROOT, "tests", "golden", "oracle_decisions_300.npz"))
    keys = [str(k) for k in fx["keys"]]
    seq0, n_frames, S = int(fx["seq0"]), int(fx["frames"]), min(a.seqs, fx["decisions"].shape[1])
    dec_o = fx["decisions"][a.lag, :S].astype(np.int64)
    pos_file = os.path.join(ROOT, "tests", "golden", "oracle_ate_300.npz" if a.lag == 0 else "oracle_ate_300_lag1.npz")
    pos_o = np.load(pos_file)["positions"][:S] if os.path.exists(pos_file) else None
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    dec_h = np.zeros((S, n_frames, len(keys)), np.int64)

    def per_frame(f, b):
        for i, st in enumerate(b.status_all()):
            dec_h[i, f] = [getattr(st, k) for k in keys]

    hist, stats, _ = parity_long.run_hip(P, cfg, sc, seq0, S, n_frames, lag=a.lag, per_frame=per_frame)
    times = vio_ct.frame_times(sc, n_frames)
    rows, kinds = [], {}
    for i in range(S):
        ff, kind = oracle_control.first_flip(dec_o[i], dec_h[i])
        row = dict(sequence=seq0 + i, first_flip_frame=ff, first_flip_kind=kind)
        if pos_o is not None:
            h = hist[i]
            fr = np.rint(h[:, 0] * sc.cam_rate).astype(int)
            d = np.linalg.norm(pos_o[i][fr] - h[:, 1:4], axis=1)
            sep = np.nonzero(d > 1e-6)[0]
            row.update(max_distance_m=float(d.max()), first_frame_beyond_1um=(int(fr[sep[0]]) if len(sep) else None),
                       distance_at_flip_m=(float(d[np.searchsorted(fr, ff)]) if ff is not None and ff >= fr[0] and np.searchsorted(fr, ff) < len(d) else None))
        rows.append(row)
        kinds[str(kind)] = kinds.get(str(kind), 0) + 1
    flips = [r["first_flip_frame"] for r in rows if r["first_flip_frame"] is not None]
    seps = [r["first_frame_beyond_1um"] for r in rows if r.get("first_frame_beyond_1um") is not None]
    md = [r["max_distance_m"] for r in rows if "max_distance_m" in r]
    dflip = [r["distance_at_flip_m"] for r in rows if r.get("distance_at_flip_m") is not None]
    out = dict(what="first differing per-frame decision, HIP path vs oracle", tracker_lag=a.lag, sequences=S, first_sequence=seq0, frames=n_frames,
               status_keys=keys, sequences_with_a_decision_flip=len(flips), median_first_flip_frame=(float(np.median(flips)) if flips else None),
               first_flip_census=kinds, separated_beyond_1um=len(seps), median_first_frame_beyond_1um=(float(np.median(seps)) if seps else None),
               median_max_distance_m=(float(np.median(md)) if md else None),
               median_distance_at_the_first_flip_m=(float(np.median(dflip)) if dflip else None), rows=rows)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "rows"}))


if __name__ == "__main__":
    main()
