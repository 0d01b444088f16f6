#!/usr/bin/env python
"""The timers build (libvio_hip_timers.so: the same sources with the in-kernel phase timers compiled in) must compute what the shipped library computes --
a different register allocation and different timing, so a difference between the two is a race or an undefined behaviour somewhere (round 6: it exposed a
rewritten block_scan_flags that every test of the default build had passed).    python tools/timers_sanity.py   ->  two identical lines"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import vio_ct
from test_gpu_batch import _drive
P = vio_ct.pkg()
out = []
for kw, n in ((dict(), 22), (dict(marg_exact=1), 22), (dict(window_size=14), 24)):
    cfg = P.canonical_config(**kw)
    sc = vio_ct.synth_like(cfg)
    b = _drive(P, cfg, sc, [60, 61, 62, 63], n)
    out.append([(b.status(i).has_prior, b.status(i).reboot_count, b.status(i).iterations_total) for i in range(4)])
print(out)
''' % (ROOT, ROOT)


def main():
    lines = []
    for lib in ("", "timers"):
        env = dict(os.environ, VIO_HIP_LIB=lib)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        lines.append(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED: " + r.stderr[-400:])
        print(lib or "default", lines[-1])
    sys.exit(0 if lines[0] == lines[1] and not lines[0].startswith("FAILED") else 1)


if __name__ == "__main__":
    main()
