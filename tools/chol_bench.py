"""Runs ON THE GPU BOX: the LDS-tile Cholesky of ps_serial on a synthetic reduced camera system (vio_stage_chol), checked against
numpy and timed inside the kernel.  Usage: python tools/chol_bench.py [nb] [reps] [blocks ...]"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("vins_rgbd_fast_amd", os.path.join(ROOT, "vins-rgbd-fast_amd", "__init__.py"))
P = importlib.util.module_from_spec(spec)
sys.modules["vins_rgbd_fast_amd"] = P
spec.loader.exec_module(P)


def spd(n, seed, cond=1e6):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    w = np.exp(rng.uniform(0, np.log(cond), n))
    a = (q * w) @ q.T
    return 0.5 * (a + a.T)


def run(nb, reps, blocks, seed=1):
    n = 16 * nb
    S = spd(n, seed)
    b = np.random.default_rng(seed + 1).standard_normal(n)
    L = np.zeros((n, n))
    x = np.zeros(n)
    us = np.zeros(5)
    rc = P.lib().vio_stage_chol(nb, reps, blocks, S.ctypes.data, b.ctypes.data, L.ctypes.data, x.ctypes.data, us.ctypes.data)
    assert rc == 0, rc
    Lr = np.linalg.cholesky(S)
    xr = np.linalg.solve(S, b)
    eL = np.abs(L - Lr).max() / np.abs(Lr).max()
    ex = np.abs(x - xr).max() / np.abs(xr).max()
    # bit-level fingerprint of the factor and the solution: a change of the kernel that is meant to keep the arithmetic must keep these
    import hashlib
    fp = hashlib.sha1(np.ascontiguousarray(np.tril(L)).tobytes() + x.tobytes()).hexdigest()[:12]
    return us, eL, ex, fp


if __name__ == "__main__":
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 11
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    for blocks in ([int(v) for v in sys.argv[3:]] or [1, 64, 256]):
        us, eL, ex, fp = run(nb, reps, blocks)
        print("nb %d  blocks %3d  factor+forward %.2f us (panel %.2f diag+trail %.2f wait %.2f)  backward %.2f us   |L - L_numpy| / max %.1e   |x - x_numpy| / max %.1e  bits %s" % (nb, blocks, us[0], us[2], us[3], us[4], us[1], eL, ex, fp))


def micro(nb=4, reps=200):
    n = 16 * nb
    S = spd(n, 3)
    b = np.zeros(n); L = np.zeros((n, n)); x = np.zeros(n); us = np.zeros(5)
    for mode, name in ((5, "empty (harness overhead)"), (1, "diag tile alone"), (2, "diag tile + SIMD mate updating"), (3, "three panel tiles")):
        assert P.lib().vio_stage_chol(nb, reps, -mode, S.ctypes.data, b.ctypes.data, L.ctypes.data, x.ctypes.data, us.ctypes.data) == 0
        print("micro %-34s %.3f us" % (name, us[0]) + ("   clock64 ticks inside: %.0f" % (us[1] * 100) if mode == 1 else ""))
    assert P.lib().vio_stage_chol(nb, 1, -6, S.ctypes.data, b.ctypes.data, L.ctypes.data, x.ctypes.data, us.ctypes.data) == 0
    print("micro v_mfma_f64_16x16x4 cycles each: chained on one accumulator %.1f, alternating two accumulators %.1f, MFMA -> v_mul -> MFMA %.1f per pair; "
          "dependent v_rsq_f64 + add %.1f" % (us[1] * 100 / 64, us[2] * 100 / 64, us[3] * 100 / 32, us[4] * 100 / 64))
    assert P.lib().vio_stage_chol(nb, 1, -4, S.ctypes.data, b.ctypes.data, L.ctypes.data, x.ctypes.data, us.ctypes.data) == 0
    # (usec5 slots 1, 2 come back as wall-clock microseconds of clock64 TICKS: undo the scaling with the 100 MHz wall clock)
    print("micro 256 dependent fma: %.3f us wall; clock64 ticks: fma chain %.0f, 64 rsq chain %.0f" % (us[0], us[1] * 100, us[2] * 100))


if __name__ == "__main__" and os.environ.get("CHOL_MICRO"):
    micro()
