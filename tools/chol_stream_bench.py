#!/usr/bin/env python
"""The streaming tile Cholesky of ps_serial_big (windows beyond 10 keyframes) on its own: correctness against numpy, bit-equality with the
LDS-resident factorisation where both apply, in-kernel microseconds.   python tools/chol_stream_bench.py [--nb 21] [--blocks 32]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def spd(n, seed, cond=1e6):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    S = (q * np.exp(rng.uniform(0, np.log(cond), n))) @ q.T
    return 0.5 * (S + S.T), rng.standard_normal(n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nb", type=int, default=21)
    ap.add_argument("--blocks", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    import vio_ct
    L = vio_ct.pkg().lib()
    for nb in (7, 11):
        S, b = spd(16 * nb, nb)
        out = []
        for mode in (1, -7):
            Lo, x, us = np.zeros_like(S), np.zeros(16 * nb), np.zeros(5)
            assert L.vio_stage_chol(nb, 1, mode, S.ctypes.data, b.ctypes.data, Lo.ctypes.data, x.ctypes.data, us.ctypes.data) == 0
            out.append((Lo, x))
        print("nb %2d  stream == lds: factor %s  solution %s" % (nb, np.array_equal(out[0][0], out[1][0]), np.array_equal(out[0][1], out[1][1])))
    nb = a.nb
    S, b = spd(16 * nb, 100 + nb)
    Lo, x, us = np.zeros_like(S), np.zeros(16 * nb), np.zeros(5)
    assert L.vio_stage_chol(nb, a.reps, a.blocks, S.ctypes.data, b.ctypes.data, Lo.ctypes.data, x.ctypes.data, us.ctypes.data) == 0
    Lr, xr = np.linalg.cholesky(S), np.linalg.solve(S, b)
    print("nb %d x %d blocks: factor err %.2e  solution err %.2e   factorisation %.1f us  backward %.1f us   (update + diagonal %.1f, panels + stores %.1f)" %
          (nb, a.blocks, np.abs(Lo - Lr).max() / np.abs(Lr).max(), np.abs(x - xr).max() / np.abs(xr).max(), us[0], us[1], us[2], us[3]))


if __name__ == "__main__":
    main()
