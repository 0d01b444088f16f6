"""Round 5 test infrastructure on the CPU: (a) sincos_det, the polynomial sin / cos predictMotion uses on BOTH sides (oracle/om.h and
csrc/dmath.h hold the same text), against libm; (b) the attribution experiment's switches (oracle/oracle.h ODEV_*, environment
OVIO_DEVIATIONS; tests/oracle_control.py `attribution`, profiles/round5_deviation_attribution.json): every switch must be an EQUIVALENT
formulation of the factor / marginalisation it replaces -- equal to the reference's form up to round-off on a single evaluation -- otherwise
the experiment would attribute a modelling difference instead of an arithmetic one."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import vio_ct
from test_oracle_kat import _proj_eval, _rand_pose, pose_plus


@pytest.fixture(scope="module")
def P():
    return vio_ct.pkg()


@pytest.fixture(scope="module")
def orc():
    return vio_ct.oracle()


def test_sincos_det_against_libm_and_shared_verbatim(orc):
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-7, 7, 20000), rng.uniform(-1e-2, 1e-2, 20000), rng.uniform(-2000, 2000, 5000), [0.0, 1e-300, np.pi / 4, -np.pi / 4, np.pi / 2]])
    s, c = np.zeros_like(x), np.zeros_like(x)
    orc.ovio_sincos_det(len(x), x.ctypes.data, s.ctypes.data, c.ctypes.data)
    es, ec = np.sin(x), np.cos(x)
    big = np.abs(es) > 1e-3
    assert (np.abs(s - es)[big] <= 2.5 * np.spacing(np.abs(es[big]))).all()
    assert (np.abs(s - es)[~big] <= 4e-19 + 2.5 * np.spacing(np.abs(es[~big]))).all() or np.abs(s - es)[~big].max() < 1e-18
    bigc = np.abs(ec) > 1e-3
    assert (np.abs(c - ec)[bigc] <= 2.5 * np.spacing(np.abs(ec[bigc]))).all()
    assert np.abs(s * s + c * c - 1).max() < 1e-15
    assert s[-5] == 0.0 and c[-5] == 1.0
    # the same text on both sides of the parity tests (the qualifier macro differs: DM_HD / inline)
    def body(path):
        t = open(path).read()
        m = re.search(r"SINCOS_DET_QUAL void sincos_det\(.*?\n}\n", t, re.S)
        assert m, path
        return m.group(0)
    assert body(os.path.join(vio_ct.ROOT, "oracle", "om.h")) == body(os.path.join(vio_ct.ROOT, "vins-rgbd-fast_amd", "csrc", "dmath.h"))


def test_pair_form_projection_equals_the_reference_form(P, orc):
    cfg = P.default_config(tr=0.0)
    rng = np.random.default_rng(77)
    for use_td in (0, 1):
        for _ in range(8):
            pi = _rand_pose(rng, 0.5)
            pj = pose_plus(pi, np.r_[rng.normal(0, 0.1, 3), rng.normal(0, 0.03, 3)])
            ex = pose_plus(np.r_[np.array(cfg.tic[:]), 0.5, -0.5, 0.5, -0.5], np.r_[np.zeros(3), rng.normal(0, 0.02, 3)])
            oi = np.r_[rng.uniform(-0.4, 0.4, 2), 1.0, rng.uniform(0, 640), rng.uniform(0, 480), rng.normal(0, 0.1, 2), 0.001, 2.0]
            oj = np.r_[rng.uniform(-0.4, 0.4, 2), 1.0, rng.uniform(0, 640), rng.uniform(0, 480), rng.normal(0, 0.1, 2), -0.002, 2.0]
            inv_dep, td = 1.0 / rng.uniform(1.5, 6.0), 0.003
            r0, J0 = _proj_eval(orc, cfg, pi, pj, ex, inv_dep, td, oi, oj, use_td)
            old = orc.ovio_set_deviations(8)
            try:
                r1, J1 = _proj_eval(orc, cfg, pi, pj, ex, inv_dep, td, oi, oj, use_td)
            finally:
                orc.ovio_set_deviations(old)
            assert np.abs(r1 - r0).max() < 1e-11 * max(1.0, np.abs(r0).max())
            assert np.abs(J1 - J0).max() < 1e-10 * max(1.0, np.abs(J0).max())
            assert not np.array_equal(J1, J0)          # a different association of the same products, not the same code


def test_cholesky_whitening_gives_the_same_normal_equations(P, orc):
    """deviation 8: M = chol(cov)^-1 instead of LLT(cov^-1).L^T -- r and J differ (another square root of the same information matrix),
    |r|^2, J^T r and J^T J do not."""
    cfg = P.default_config()
    rng = np.random.default_rng(9)
    a0, g0 = np.array([0.1, -0.2, 9.7]), np.array([0.01, 0.02, -0.01])
    ba, bg = np.zeros(3), np.zeros(3)
    h = C.c_void_p(orc.ovio_preint_create(C.byref(cfg), a0.ctypes.data, g0.ctypes.data, ba.ctypes.data, bg.ctypes.data))
    for _ in range(20):
        a, g = a0 + rng.normal(0, 0.3, 3), g0 + rng.normal(0, 0.05, 3)
        orc.ovio_preint_push(h, 0.005, a.ctypes.data, g.ctypes.data)
    pi = _rand_pose(rng, 0.3)
    pj = pose_plus(pi, np.r_[rng.normal(0, 0.02, 3), rng.normal(0, 0.01, 3)])
    sbi, sbj = rng.normal(0, 0.05, 9), rng.normal(0, 0.05, 9)

    def ev():
        r, J = np.zeros(15), np.zeros(15 * 32)
        orc.ovio_eval_imu(h, cfg.g_norm, pi.ctypes.data, sbi.ctypes.data, pj.ctypes.data, sbj.ctypes.data, r.ctypes.data, J.ctypes.data)
        Jf = np.concatenate([J[:105].reshape(15, 7)[:, :6], J[105:240].reshape(15, 9), J[240:345].reshape(15, 7)[:, :6], J[345:480].reshape(15, 9)], axis=1)
        return r, Jf
    r0, J0 = ev()
    old = orc.ovio_set_deviations(1)
    try:
        r1, J1 = ev()
    finally:
        orc.ovio_set_deviations(old)
    orc.ovio_preint_destroy(h)
    assert np.abs(r1 - r0).max() > 1e-6 * np.abs(r0).max()                      # a different square root ...
    assert abs(r1 @ r1 - r0 @ r0) < 1e-9 * (r0 @ r0)                              # ... of the same quadratic form
    H0, H1 = J0.T @ J0, J1.T @ J1
    assert np.abs(H1 - H0).max() < 1e-9 * np.abs(H0).max()
    assert np.abs(J1.T @ r1 - J0.T @ r0).max() < 1e-9 * np.abs(J0.T @ r0).max()


@pytest.mark.parametrize("mask", [2, 4, 6])
def test_marginalisation_variants_give_the_same_prior(orc, mask):
    """deviations 13 (quadratic-form prior) and 10 (analytic elimination of the landmark block, which is diagonal) against the literal
    marg_finish on a system with the marginalisation's structure: 15 dense rows + F landmark rows that only meet themselves."""
    rng = np.random.default_rng(31 + mask)
    F, n = 24, 40
    m, N = 15 + F, 15 + F + n
    rows = []
    for l in range(F):                                  # each landmark: a few residual rows touching pose columns and its own column
        for _ in range(3):
            row = np.zeros(N)
            row[:15] = rng.normal(size=15) * (rng.random(15) < 0.5)
            row[m:] = rng.normal(size=n) * (rng.random(n) < 0.2)
            row[15 + l] = rng.normal() + 2.0
            rows.append(row)
    for _ in range(3 * (15 + n)):                       # prior / IMU-like rows without landmark columns
        row = np.zeros(N)
        row[:15] = rng.normal(size=15)
        row[m:] = rng.normal(size=n)
        rows.append(row)
    Jf = np.array(rows)
    rf = rng.normal(size=len(rows))
    A, b = np.ascontiguousarray(Jf.T @ Jf), np.ascontiguousarray(Jf.T @ rf)
    assert np.abs(A[15:m, 15:m] - np.diag(np.diag(A[15:m, 15:m]))).max() == 0

    def run(env):
        J, r = np.zeros((n, n)), np.zeros(n)
        if env:
            os.environ["OVIO_DEVIATIONS"] = str(env)
        try:
            orc.ovio_marg_finish(m, n, A.ctypes.data, b.ctypes.data, J.ctypes.data, r.ctypes.data)
        finally:
            os.environ.pop("OVIO_DEVIATIONS", None)
        return J, r
    J0, r0 = run(0)
    H0, g0 = J0.T @ J0, J0.T @ r0
    J1, r1 = run(mask)
    H1, g1 = (J1, r1) if mask & 2 else (J1.T @ J1, J1.T @ r1)    # quadratic mode hands back (A, b) itself
    assert np.abs(H1 - H0).max() < 1e-9 * np.abs(H0).max()
    assert np.abs(g1 - g0).max() < 1e-9 * np.abs(g0).max()


def test_deviation_variants_track_the_base_oracle_on_a_sequence(P):
    """whole pipeline, 45 frames: every switch (and all together) makes the same decisions as the base oracle while the difference is
    small and stays within micrometres -- the variants differ from the base by round-off amplified by the estimator, nothing else"""
    import oracle_control as OC
    names = ["base", "dev8", "dev13", "dev10", "dev11", "devall"]
    try:
        z = OC.run_variants(703, 45, names)
    finally:
        vio_ct.oracle().ovio_set_deviations(0)
    assert len(z["base_pos"]) >= 25
    for k in names[1:]:
        n = min(len(z["base_pos"]), len(z[k + "_pos"]))
        d = np.linalg.norm(z["base_pos"][:n] - z[k + "_pos"][:n], axis=1)
        assert 0 < d.max() < 2e-5, (k, d.max())
        assert np.array_equal(z["base_status"][:15], z[k + "_status"][:15]), k
