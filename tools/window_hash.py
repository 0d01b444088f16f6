#!/usr/bin/env python
"""Fingerprint of the estimator's output on a few short runs (window states, landmarks, solver iteration counts): a refactoring that claims
bit-identical results is checked by comparing this file's output before and after.    python tools/window_hash.py > gpurun_out/hash.txt"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vio_ct  # noqa: E402
from test_gpu_batch import _drive  # noqa: E402


def main():
    P = vio_ct.pkg()
    for kw, n in ((dict(), 34), (dict(estimate_extrinsic=1, estimate_td=1), 30), (dict(window_size=20), 40), (dict(window_size=14, estimate_extrinsic=1, estimate_td=1), 32)):
        cfg = P.canonical_config(**kw)
        sc = vio_ct.synth_like(cfg)
        h = hashlib.sha256()
        its = []

        def hook(f, b):
            for i in range(3):
                h.update(np.ascontiguousarray(b.window(i)).tobytes())

        b = _drive(P, cfg, sc, [60, 61, 62], n, hook=hook)
        for i in range(3):
            h.update(np.ascontiguousarray(b.landmarks(i)).tobytes())
            its.append(b.status(i).iterations_total)
        print(kw, n, its, h.hexdigest()[:24])


if __name__ == "__main__":
    main()
