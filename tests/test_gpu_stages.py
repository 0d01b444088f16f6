"""GPU parity of the single stages (through the C ABI) against the CPU oracle on identical inputs.
Integer / fixed-point stages must be bit-exact; FP64 factor math agrees to round-off (tolerances stated per test)."""
import ctypes as C

import numpy as np
import pytest

import vio_ct

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def frames(P):
    cfg = P.default_config()
    sc = vio_ct.synth_like(cfg)
    syn = P.Synth(sc)
    return cfg, sc, [syn.render_host(3, t) for t in (2.5, 2.6)]


def test_pyr_down_bit_exact(P, orc, frames):
    cfg, sc, fr = frames
    g = fr[0][0]
    h, w = g.shape
    for img in (g, np.ascontiguousarray(g[:241, :333])):
        hh, ww = img.shape
        ref = np.zeros(((hh + 1) // 2, (ww + 1) // 2), np.uint8)
        out = np.zeros_like(ref)
        orc.ovio_pyr_down(img.ctypes.data, ww, hh, ref.ctypes.data)
        assert P.lib().vio_stage_pyr_down(img.ctypes.data, ww, hh, out.ctypes.data) == 0
        assert np.array_equal(ref, out)


def test_fast_roi_bit_exact(P, orc, frames):
    cfg, sc, fr = frames
    g = fr[0][0]
    H, W = g.shape
    total = 0
    for (rx, ry, rw, rh) in [(0, 0, 109, 99), (103, 93, 112, 102), (527, 381, 113, 99), (315, 189, 112, 102), (10, 20, 7, 7), (0, 0, 30, 9)]:
        cap = 4096
        a, b = np.zeros((cap, 3), np.float32), np.zeros((cap, 3), np.float32)
        na = orc.ovio_fast_roi(g.ctypes.data, W, H, rx, ry, rw, rh, cap, a.ctypes.data)
        nb = P.lib().vio_stage_fast_roi(g.ctypes.data, W, H, rx, ry, rw, rh, cap, b.ctypes.data)
        assert na == nb
        assert np.array_equal(a[:na], b[:nb])
        total += na
    assert total > 20  # the synthetic texture must actually produce corners


def _corners(orc, g, n=120):
    H, W = g.shape
    out = np.zeros((8192, 3), np.float32)
    k = orc.ovio_fast_roi(g.ctypes.data, W, H, 0, 0, W, H, 8192, out.ctypes.data)
    pts = out[:k]
    pts = pts[(pts[:, 0] > 30) & (pts[:, 0] < W - 30) & (pts[:, 1] > 30) & (pts[:, 1] < H - 30)]
    idx = np.argsort(-pts[:, 2], kind="stable")[:n]
    return np.ascontiguousarray(pts[idx, :2])


@pytest.mark.parametrize("max_level", [1, 3])
def test_lk_bit_exact(P, orc, frames, max_level):
    cfg, sc, fr = frames
    g0, g1 = fr[0][0], fr[1][0]
    H, W = g0.shape
    prev = _corners(orc, g0)
    # add border / out-of-image cases
    prev = np.vstack([prev, [[2.5, 3.5], [W - 2.0, H - 3.0], [W / 2, 1.0]]]).astype(np.float32)
    n = len(prev)
    rng = np.random.default_rng(0)
    init = (prev + rng.uniform(-2, 2, prev.shape)).astype(np.float32)
    a, b = init.copy(), init.copy()
    sa, sb = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    orc.ovio_lk(g0.ctypes.data, g1.ctypes.data, W, H, max_level, n, prev.ctypes.data, a.ctypes.data, sa.ctypes.data, 1)
    assert P.lib().vio_stage_lk(g0.ctypes.data, g1.ctypes.data, W, H, max_level, n, prev.ctypes.data, b.ctypes.data, sb.ctypes.data) == 0
    assert np.array_equal(sa, sb)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), float(np.abs(a - b).max())
    assert sa.sum() > n // 2
    moved = np.linalg.norm(a[sa > 0] - prev[sa > 0], axis=1)
    assert moved.mean() > 0.5  # the two frames really differ


def test_ransac_same_inliers(P, orc):
    cfg = P.default_config()
    rng = np.random.default_rng(1)
    n = 150
    # synthetic two-view geometry in virtual-pinhole pixels (f = 460, c = (320,240))
    X = np.c_[rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(2, 6, n)]
    th = 0.05
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    t = np.array([0.15, 0.02, 0.05])
    X2 = (R @ X.T).T + t
    p1 = np.ascontiguousarray((460 * X[:, :2] / X[:, 2:3] + [320, 240]).astype(np.float32))
    p2 = np.ascontiguousarray((460 * X2[:, :2] / X2[:, 2:3] + [320, 240]).astype(np.float32))
    p2 += rng.normal(0, 0.2, p2.shape).astype(np.float32)
    bad = rng.choice(n, 25, replace=False)
    p2[bad] += rng.uniform(8, 30, (25, 2)).astype(np.float32)
    for pts2 in (p2, p1.copy()):  # second case: zero motion (rank-deficient samples)
        sa, sb = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        orc.ovio_ransac(C.byref(cfg), n, p1.ctypes.data, pts2.ctypes.data, sa.ctypes.data)
        assert P.lib().vio_stage_ransac(C.byref(cfg), n, p1.ctypes.data, pts2.ctypes.data, sb.ctypes.data) == 0
        assert np.array_equal(sa, sb)
    sa = np.zeros(n, np.uint8)
    orc.ovio_ransac(C.byref(cfg), n, p1.ctypes.data, p2.ctypes.data, sa.ctypes.data)
    assert sa[bad].sum() <= 2 and sa.sum() >= 110


def _rand_pose(rng, scale=1.0):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    if q[3] < 0:
        q = -q
    return np.r_[rng.normal(size=3) * scale, q]  # x y z qx qy qz qw


def test_imu_factor_matches_oracle(P, orc):
    cfg = P.default_config()
    rng = np.random.default_rng(2)
    n = 20
    dt = np.full(n, 0.005)
    acc = rng.normal(0, 1.0, (n, 3)) + [0, 0, 9.8]
    gyr = rng.normal(0, 0.3, (n, 3))
    acc0, gyr0 = acc[0] + 0.01, gyr[0] - 0.01
    ba, bg = rng.normal(0, 0.02, 3), rng.normal(0, 0.002, 3)
    pi, pj = _rand_pose(rng), _rand_pose(rng)
    sbi, sbj = rng.normal(0, 0.3, 9), rng.normal(0, 0.3, 9)
    sbi[3:] *= 0.05; sbj[3:] *= 0.05
    h = C.c_void_p(orc.ovio_preint_create(C.byref(cfg), acc0.ctypes.data, gyr0.ctypes.data, ba.ctypes.data, bg.ctypes.data))
    for k in range(n):
        orc.ovio_preint_push(h, dt[k], acc[k].ctypes.data, gyr[k].ctypes.data)
    ref_pre = np.zeros(461); orc.ovio_preint_get(h, ref_pre.ctypes.data)
    ref_r, ref_J = np.zeros(15), np.zeros(480)
    orc.ovio_eval_imu(h, cfg.g_norm, pi.ctypes.data, sbi.ctypes.data, pj.ctypes.data, sbj.ctypes.data, ref_r.ctypes.data, ref_J.ctypes.data)
    orc.ovio_preint_destroy(h)
    pre, r, J = np.zeros(461), np.zeros(15), np.zeros(480)
    rc = P.lib().vio_stage_imu_factor(C.byref(cfg), n, dt.ctypes.data, acc.ctypes.data, gyr.ctypes.data, acc0.ctypes.data, gyr0.ctypes.data,
                                      ba.ctypes.data, bg.ctypes.data, pi.ctypes.data, sbi.ctypes.data, pj.ctypes.data, sbj.ctypes.data,
                                      pre.ctypes.data, r.ctypes.data, J.ctypes.data)
    assert rc == 0
    # pre-integration: same operation order -> agreement to a few ulp (tolerance 1e-12 relative to the largest entry)
    assert np.abs(pre - ref_pre).max() <= 1e-12 * max(1.0, np.abs(ref_pre).max())
    # The HIP path whitens with M = chol(cov)^-1, the reference with LLT(cov^-1).L^T: both satisfy M^T M = cov^-1, so the
    # quantities the solver consumes (|r|^2, J^T r, J^T J) must agree; tolerance 1e-7 relative (15x15 inverse + Cholesky).
    def blocks(Jf):
        return np.hstack([Jf[:105].reshape(15, 7), Jf[105:240].reshape(15, 9), Jf[240:345].reshape(15, 7), Jf[345:].reshape(15, 9)])
    Jm, Jr = blocks(J), blocks(ref_J)
    assert abs(r @ r - ref_r @ ref_r) <= 1e-7 * max(1.0, ref_r @ ref_r)
    assert np.abs(Jm.T @ r - Jr.T @ ref_r).max() <= 1e-7 * max(1.0, np.abs(Jr.T @ ref_r).max())
    assert np.abs(Jm.T @ Jm - Jr.T @ Jr).max() <= 1e-7 * max(1.0, np.abs(Jr.T @ Jr).max())


@pytest.mark.parametrize("form", ["pair", "residual"])
@pytest.mark.parametrize("use_td", [0, 1])
def test_projection_factor_matches_oracle(P, orc, use_td, form):
    """ProjectionFactor / ProjectionTdFactor::Evaluate through both device routines: the frame-pair form of the solver's hot loop
    (be_factors.h eval_projection_pair) and the per-residual form used by the marginalisation and the outlier rejection, on pairs
    with a genuinely different rotation Rj != Ri and a non-trivial extrinsic.  Tolerance 1e-10 relative (different association of
    the same products; the pair form pre-multiplies the rotations)."""
    cfg = P.default_config(tr=0.01)
    rng = np.random.default_rng(3 + use_td)
    fn = P.lib().vio_stage_projection if form == "pair" else P.lib().vio_stage_projection_residual
    worst = 0.0
    for _ in range(12):
        pi = _rand_pose(rng, 0.5); pj = pi.copy(); pj[:3] += rng.normal(0, 0.1, 3)
        dq = np.r_[rng.normal(0, 0.05, 3), 1.0]; dq /= np.linalg.norm(dq)
        # q_j = q_i * dq (x y z w): up to ~6 degrees between the two body frames
        x1, y1, z1, w1 = pi[3:]; x2, y2, z2, w2 = dq
        pj[3:] = [w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
                  w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2]
        qe = np.r_[0.5, -0.5, 0.5, -0.5] + rng.normal(0, 0.02, 4); qe /= np.linalg.norm(qe)
        ex = np.r_[np.array(cfg.tic[:]), qe]
        oi = np.r_[rng.uniform(-0.4, 0.4, 2), 1.0, rng.uniform(0, 640), rng.uniform(0, 480), rng.normal(0, 0.1, 2), 0.001, 2.0]
        oj = np.r_[rng.uniform(-0.4, 0.4, 2), 1.0, rng.uniform(0, 640), rng.uniform(0, 480), rng.normal(0, 0.1, 2), -0.002, 2.0]
        inv_dep, td = 1.0 / rng.uniform(1.5, 6.0), 0.003
        r0, J0, r1, J1 = np.zeros(2), np.zeros(46), np.zeros(2), np.zeros(46)
        orc.ovio_eval_projection(C.byref(cfg), pi.ctypes.data, pj.ctypes.data, ex.ctypes.data, inv_dep, td, oi.ctypes.data, oj.ctypes.data,
                                 use_td, r0.ctypes.data, J0.ctypes.data)
        rc = fn(C.byref(cfg), pi.ctypes.data, pj.ctypes.data, ex.ctypes.data, inv_dep, td, oi.ctypes.data, oj.ctypes.data, use_td,
                r1.ctypes.data, J1.ctypes.data)
        assert rc == 0
        assert np.abs(pj[3:] - pi[3:]).max() > 1e-3
        tol = 1e-10
        assert np.abs(r1 - r0).max() <= tol * max(1.0, np.abs(r0).max())
        assert np.abs(J1 - J0).max() <= tol * max(1.0, np.abs(J0).max())
        worst = max(worst, float(np.abs(J1 - J0).max() / max(1.0, np.abs(J0).max())))
    assert worst < 1e-10


def test_imu_block_on_the_matrix_cores_matches_oracle(P, orc):
    """The IMU factor as be_solve processes it (residual on one lane, four raw Jacobian column groups on four lanes, rows whitened on
    the fly with M = chol(cov)^-1, [J r]^T [J r] on v_mfma_f64_16x16x4) against IMUFactor::Evaluate of the oracle: the 31 x 31 Gram
    matrix (J^T J, J^T r, |r|^2) within 1e-7 relative (15x15 inverse + Cholesky of the covariance on both sides, different whitening
    factor by design -- DESIGN.md deviation 8)."""
    cfg = P.default_config()
    for seed in (5, 6, 7):
        rng = np.random.default_rng(seed)
        n = 20
        dt = np.full(n, 0.005)
        acc = rng.normal(0, 1.0, (n, 3)) + [0, 0, 9.8]
        gyr = rng.normal(0, 0.3, (n, 3))
        acc0, gyr0 = acc[0] + 0.01, gyr[0] - 0.01
        ba, bg = rng.normal(0, 0.02, 3), rng.normal(0, 0.002, 3)
        pi, pj = _rand_pose(rng), _rand_pose(rng)
        sbi, sbj = rng.normal(0, 0.3, 9), rng.normal(0, 0.3, 9)
        sbi[3:] *= 0.05; sbj[3:] *= 0.05
        h = C.c_void_p(orc.ovio_preint_create(C.byref(cfg), acc0.ctypes.data, gyr0.ctypes.data, ba.ctypes.data, bg.ctypes.data))
        for k in range(n):
            orc.ovio_preint_push(h, dt[k], acc[k].ctypes.data, gyr[k].ctypes.data)
        ref_r, ref_J = np.zeros(15), np.zeros(480)
        orc.ovio_eval_imu(h, cfg.g_norm, pi.ctypes.data, sbi.ctypes.data, pj.ctypes.data, sbj.ctypes.data, ref_r.ctypes.data, ref_J.ctypes.data)
        orc.ovio_preint_destroy(h)
        G = np.zeros((31, 31))
        rc = P.lib().vio_stage_imu_block(C.byref(cfg), n, dt.ctypes.data, acc.ctypes.data, gyr.ctypes.data, acc0.ctypes.data, gyr0.ctypes.data,
                                         ba.ctypes.data, bg.ctypes.data, pi.ctypes.data, sbi.ctypes.data, pj.ctypes.data, sbj.ctypes.data,
                                         G.ctypes.data)
        assert rc == 0
        # tangent columns of the reference Jacobians (the 7th pose column is identically zero)
        Jr = np.hstack([ref_J[:105].reshape(15, 7)[:, :6], ref_J[105:240].reshape(15, 9), ref_J[240:345].reshape(15, 7)[:, :6],
                        ref_J[345:].reshape(15, 9)])
        A = np.hstack([Jr, ref_r[:, None]])
        Gref = A.T @ A
        assert np.abs(G - G.T).max() == 0.0
        assert np.abs(G - Gref).max() <= 1e-7 * np.abs(Gref).max(), float(np.abs(G - Gref).max() / np.abs(Gref).max())


@pytest.mark.parametrize("nb", [1, 2, 7, 11])
def test_tile_cholesky_against_numpy(P, nb):
    """The dense solve of the reduced camera system (Ceres DENSE_SCHUR, estimator.cpp:1251-1263) through the LDS-tile Cholesky of
    ps_serial (be_linalg.h chol_tiles / chol_backward_tiles: diagonal blocks by rank-1 updates on v_mfma_f64_16x16x4, panels as
    A M^T with M = L^-1 of the block) against numpy on a matrix with condition 1e6: the factor to 1e-11 of its largest entry, the
    solution to 1e-9 (cond x eps); a matrix that is not positive definite is reported, not factored."""
    n = 16 * nb
    rng = np.random.default_rng(40 + nb)
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    w = np.exp(rng.uniform(0, np.log(1e6), n))
    S = (q * w) @ q.T
    S = 0.5 * (S + S.T)
    b = rng.standard_normal(n)
    L = np.zeros((n, n)); x = np.zeros(n); us = np.zeros(5)
    assert P.lib().vio_stage_chol(nb, 1, 1, S.ctypes.data, b.ctypes.data, L.ctypes.data, x.ctypes.data, us.ctypes.data) == 0
    Lr = np.linalg.cholesky(S)
    assert np.abs(np.triu(L, 1)).max() == 0.0
    assert np.abs(L - Lr).max() <= 1e-11 * np.abs(Lr).max()
    xr = np.linalg.solve(S, b)
    assert np.abs(x - xr).max() <= 1e-9 * np.abs(xr).max()
    S2 = S.copy()
    S2[n - 3, n - 3] = -1.0
    assert P.lib().vio_stage_chol(nb, 1, 1, S2.ctypes.data, b.ctypes.data, L.ctypes.data, x.ctypes.data, us.ctypes.data) == 0
    assert np.isnan(x).all()


@pytest.mark.parametrize("nb", [12, 21, 24])
def test_streaming_tile_cholesky_against_numpy_and_the_lds_path(P, nb):
    """ps_serial_big's factorisation for windows beyond 10 keyframes (be_linalg.h chol_tiles_stream / chol_backward_tiles: the tiles stay in
    HBM / L2, one block column at a time goes through LDS): against numpy at the sizes only it serves, and bit for bit against the
    LDS-resident factorisation (vio_stage_chol blocks = -7 forces the streaming path) where both apply -- the two share every arithmetic step."""
    def spd(n, seed):
        rng = np.random.default_rng(seed)
        q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        S = (q * np.exp(rng.uniform(0, np.log(1e6), n))) @ q.T
        return 0.5 * (S + S.T), rng.standard_normal(n)
    n = 16 * nb
    S, b = spd(n, 70 + nb)
    L, x, us = np.zeros((n, n)), np.zeros(n), np.zeros(5)
    assert P.lib().vio_stage_chol(nb, 1, 2, S.ctypes.data, b.ctypes.data, L.ctypes.data, x.ctypes.data, us.ctypes.data) == 0
    Lr, xr = np.linalg.cholesky(S), np.linalg.solve(S, b)
    assert np.abs(np.triu(L, 1)).max() == 0.0 and np.abs(L - Lr).max() <= 1e-11 * np.abs(Lr).max()
    assert np.abs(x - xr).max() <= 1e-9 * np.abs(xr).max()
    S2 = S.copy(); S2[n - 5, n - 5] = -1.0
    assert P.lib().vio_stage_chol(nb, 1, 1, S2.ctypes.data, b.ctypes.data, L.ctypes.data, x.ctypes.data, us.ctypes.data) == 0
    assert np.isnan(x).all()
    m = 11 if nb > 12 else 6
    S3, b3 = spd(16 * m, 90 + nb)
    out = []
    for mode in (1, -7):
        Lo, xo = np.zeros_like(S3), np.zeros(16 * m)
        assert P.lib().vio_stage_chol(m, 1, mode, S3.ctypes.data, b3.ctypes.data, Lo.ctypes.data, xo.ctypes.data, us.ctypes.data) == 0
        out.append((Lo, xo))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
