"""Generates tests/golden/oracle_ate_300.npz: the ORACLE's absolute trajectory error over 300 frames of the canonical synthetic
workload for sequences 700 .. 700 + N - 1 (tracker lag 0), plus the oracle's positions for the first 128 of them.  The oracle needs no
GPU, so this runs wherever there are spare CPU hours (N = 1024: about 8 CPU hours, mostly the host renderer); the GPU-side parity test
(tests/test_gpu_parity3.py) then only has to run the HIP path.

    python tests/golden/make_oracle_ate_300.py --seqs 1024 --procs 7
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", type=int, default=1024)
    ap.add_argument("--seq0", type=int, default=700)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--procs", type=int, default=0)
    ap.add_argument("--keep", type=int, default=128, help="sequences whose positions are stored")
    ap.add_argument("--out", default=os.path.join(HERE, "oracle_ate_300.npz"))
    a = ap.parse_args()
    import parity_long
    import vio_ct
    res = parity_long.run_oracle_pool(range(a.seq0, a.seq0 + a.seqs), a.frames, lag=0, procs=a.procs or None)
    ate = np.zeros(a.seqs); nfr = np.zeros(a.seqs, np.int32); reb = np.zeros(a.seqs, np.int32); first = np.zeros(a.seqs, np.int32)
    pos = np.zeros((min(a.keep, a.seqs), a.frames, 3))
    for i in range(a.seqs):
        fr, po, gt, r = res[a.seq0 + i]
        ate[i] = vio_ct.ate_rmse(po, gt); nfr[i] = len(po); reb[i] = r; first[i] = fr[0]
        if i < len(pos):
            pos[i, fr] = po
    np.savez_compressed(a.out, seq0=a.seq0, frames=a.frames, ate=ate, n_rows=nfr, reboots=reb, first_frame=first, positions=pos)
    print("oracle: %d sequences, mean ATE %.4f mm, reboots %d" % (a.seqs, ate.mean() * 1e3, reb.sum()))


if __name__ == "__main__":
    main()
