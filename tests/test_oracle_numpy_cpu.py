"""Independent numpy / scipy cross-checks of the three places where the oracle restates a third-party numerical routine of the reference
(Eigen's SelfAdjointEigenSolver in the marginalisation, Ceres' dogleg step, OpenCV's findFundamentalMat): the quantities are dumped
from a REAL oracle pipeline run (test hooks OVIO_DUMP_PRIOR / OVIO_DUMP_SOLVE in oracle/backend.cpp) and recomputed here with dense
LAPACK-backed numpy / scipy code that shares nothing with the oracle's own linear algebra (oracle/om.h)."""
import ctypes as C
import os

import numpy as np
import pytest
import scipy.linalg

import vio_ct


def _run_pipeline(P, n_frames, seq, env_key, path):
    """oracle pipeline over n_frames with the dump hook on; returns (cfg, prior after the last frame)"""
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    last = {}

    def hook(f, o):
        if f == n_frames - 1:
            last["prior"] = o.prior()
    os.environ[env_key] = path
    try:
        vio_ct.run_oracle_sequence(cfg, sc, seq, n_frames, hook=hook)
    finally:
        del os.environ[env_key]
    return cfg, last.get("prior")


def _read_prior_dump(path):
    raw = np.fromfile(path, dtype=np.float64)
    recs, o = [], 0
    while o < len(raw):
        m, n = int(raw[o]), int(raw[o + 1]); o += 2
        q = m + n
        A = raw[o:o + q * q].reshape(q, q); o += q * q
        b = raw[o:o + q]; o += q
        As = raw[o:o + n * n].reshape(n, n); o += n * n
        br = raw[o:o + n]; o += n
        recs.append((m, n, A, b, As, br))
    return recs


def test_marginalisation_against_scipy_eigh(P, tmp_path):
    """MarginalizationInfo::marginalize (marginalization_factor.cpp:258-315) on the systems of a real run: Schur complement with the
    eigenvalue-truncated inverse of A_mm, then the factorisation A = V S V^T -> J = S^1/2 V^T, r = S^-1/2 V^T b.  scipy.linalg.eigh
    (LAPACK dsyevd) replaces the oracle's cyclic Jacobi in both places."""
    path = str(tmp_path / "prior.bin")
    cfg, prior = _run_pipeline(P, 30, 6, "OVIO_DUMP_PRIOR", path)
    recs = _read_prior_dump(path)
    assert len(recs) >= 4
    kinds = set()
    eps = 1e-8
    for (m, n, A, b, As, br) in recs:
        kinds.add(m)
        Amm = 0.5 * (A[:m, :m] + A[:m, :m].T)
        w, V = scipy.linalg.eigh(Amm)
        Ainv = (V * np.where(w > eps, 1.0 / np.where(w > eps, w, 1.0), 0.0)) @ V.T
        Arm = A[m:, :m]
        Ar = A[m:, m:] - Arm @ Ainv @ A[:m, m:]
        brn = b[m:] - Arm @ Ainv @ b[:m]
        Asn = 0.5 * (Ar + Ar.T)
        # tolerance: 1e-9 of the result plus the round-off floor of the subtraction A_rr - A_rm A_mm^+ A_mr (the very first system is
        # all cancellation: nothing but the oldest frame's own factors is marginalised, the result is ~1e-13 of the input)
        tolA = 1e-9 * np.abs(Asn).max() + 1e-12 * np.abs(A).max()
        assert np.abs(As - Asn).max() < tolA, (m, n, float(np.abs(As - Asn).max()), tolA)
        assert np.abs(br - brn).max() < 1e-9 * np.abs(brn).max() + 1e-12 * np.abs(A).max() * max(np.abs(b).max() / np.abs(A).max(), 1e-3)
    # MARGIN_SECOND_NEW drops a pose (m = 6), MARGIN_OLD a pose + speed-bias + the landmarks anchored in the oldest frame (m >= 15)
    assert min(kinds) in (6, 15) and max(kinds) > 15, kinds
    # the prior the estimator holds after the last marginalisation against the factorisation of the last dumped system
    n = recs[-1][1]
    J, r = np.asarray(prior[0], float).reshape(n, n), np.asarray(prior[1], float)
    As, br = recs[-1][4], recs[-1][5]
    w, V = scipy.linalg.eigh(As)
    keep = w > eps
    Jn = (np.sqrt(w[keep])[:, None]) * V[:, keep].T
    rn = (V[:, keep].T @ br) / np.sqrt(w[keep])
    # rows (eigenvector order / sign) are not unique: compare what the solver consumes, J^T J, J^T r and |r|^2
    scale = np.abs(As).max()
    assert np.abs(J.T @ J - Jn.T @ Jn).max() < 1e-9 * scale
    assert np.abs(J.T @ r - Jn.T @ rn).max() < 1e-8 * max(np.abs(br).max(), 1.0)
    assert abs(r @ r - rn @ rn) < 1e-6 * max(rn @ rn, 1e-12)
    assert int(keep.sum()) == int((np.abs(J).sum(1) > 0).sum())      # same rank after the 1e-8 truncation


def _read_solve_dump(path):
    raw = np.fromfile(path, dtype=np.float64)
    recs, o = [], 0
    while o < len(raw):
        Pa, Fa, mu, alpha = int(raw[o]), int(raw[o + 1]), raw[o + 2], raw[o + 3]; o += 4
        def take(k, shape=None):
            nonlocal o
            v = raw[o:o + k]; o += k
            return v.reshape(shape) if shape else v
        Hs = take(Pa * Pa, (Pa, Pa)); Hpls = take(Fa * Pa, (Fa, Pa))
        Hlls = take(Fa); gs = take(Pa); gls = take(Fa); dgp = take(Pa); dgl = take(Fa); gnp = take(Pa); gnl = take(Fa)
        recs.append(dict(Pa=Pa, Fa=Fa, mu=mu, alpha=alpha, Hs=Hs, Hpls=Hpls, Hlls=Hlls, gs=gs, gls=gls, dgp=dgp, dgl=dgl, gnp=gnp, gnl=gnl))
    return recs


def test_dogleg_points_against_dense_numpy(P, tmp_path):
    """The two points Ceres' traditional dogleg interpolates between (dogleg_strategy.cc: ComputeGaussNewtonStep with the mu D^2
    regularisation, ComputeCauchyPoint), on the linear systems of a real run.  The oracle eliminates the landmarks by a Schur
    complement and uses its own Cholesky; here the full (poses + landmarks) system is solved densely with numpy."""
    path = str(tmp_path / "solve.bin")
    _run_pipeline(P, 22, 4, "OVIO_DUMP_SOLVE", path)
    recs = _read_solve_dump(path)
    assert len(recs) >= 6
    checked = 0
    for rec in recs[:12]:
        Pa, Fa = rec["Pa"], rec["Fa"]
        H = np.zeros((Pa + Fa, Pa + Fa))
        H[:Pa, :Pa] = rec["Hs"]; H[Pa:, :Pa] = rec["Hpls"]; H[:Pa, Pa:] = rec["Hpls"].T; H[Pa:, Pa:] = np.diag(rec["Hlls"])
        g = np.r_[rec["gs"], rec["gls"]]
        D = np.r_[rec["dgp"], rec["dgl"]]
        assert np.allclose(D, np.sqrt(np.clip(np.diag(H), 1e-6, 1e32)))          # Jacobi-scaling diagonal (trust_region_minimizer)
        # Gauss-Newton step of the regularised system, expressed in the D-scaled space like the oracle's gnp / gnl
        y = np.linalg.solve(H + rec["mu"] * np.diag(D * D), g)
        gn = -y * D
        ref = np.r_[rec["gnp"], rec["gnl"]]
        assert np.abs(gn - ref).max() < 1e-7 * max(np.abs(ref).max(), 1e-12), float(np.abs(gn - ref).max() / np.abs(ref).max())
        # Cauchy step length alpha = |grad|^2 / |J D^-1 grad|^2 with grad = D^-1 g
        grad = g / D
        sg = grad / D
        alpha = (grad @ grad) / (sg @ H @ sg)
        assert abs(alpha - rec["alpha"]) < 1e-9 * abs(alpha)
        checked += 1
    assert checked >= 6


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def _eight_point(x1, x2):
    """normalised 8-point algorithm (Hartley), numpy SVD"""
    def norm(x):
        c = x.mean(0)
        s = np.sqrt(2.0) / np.sqrt(((x - c) ** 2).sum(1)).mean()
        T = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1]])
        return (x - c) * s, T
    a, T1 = norm(x1); b, T2 = norm(x2)
    A = np.c_[b[:, 0] * a[:, 0], b[:, 0] * a[:, 1], b[:, 0], b[:, 1] * a[:, 0], b[:, 1] * a[:, 1], b[:, 1], a[:, 0], a[:, 1], np.ones(len(a))]
    F = np.linalg.svd(A)[2][-1].reshape(3, 3)
    U, S, Vt = np.linalg.svd(F)
    F = U @ np.diag([S[0], S[1], 0]) @ Vt
    return T2.T @ F @ T1


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fundamental_ransac_against_numpy_geometry(P, seed):
    """rejectWithF (feature_tracker.cpp:441-473, cv::findFundamentalMat FM_RANSAC, F_THRESHOLD px at FOCAL_LENGTH) on synthetic
    correspondences of a known motion with 25 % gross outliers: the status vector has to keep what the TRUE epipolar geometry keeps,
    and the fundamental matrix numpy's 8-point algorithm fits to the oracle's inliers has to be the true one."""
    rng = np.random.default_rng(seed)
    cfg = P.canonical_config()
    f, cx, cy = cfg.focal_length, cfg.width / 2.0, cfg.height / 2.0
    n = 160
    X = np.c_[rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(2.0, 8.0, n)]
    ang = np.array([0.02, -0.03, 0.015])
    R = scipy.linalg.expm(_skew(ang))
    t = np.array([0.12, -0.05, 0.03])
    X2 = X @ R.T + t
    p1 = np.c_[f * X[:, 0] / X[:, 2] + cx, f * X[:, 1] / X[:, 2] + cy]
    p2 = np.c_[f * X2[:, 0] / X2[:, 2] + cx, f * X2[:, 1] / X2[:, 2] + cy]
    p2 += rng.normal(0, 0.15, p2.shape)
    bad = rng.random(n) < 0.25
    p2[bad] += rng.uniform(-40, 40, (int(bad.sum()), 2)) + np.sign(rng.normal(size=(int(bad.sum()), 2))) * 8
    a32, b32 = np.ascontiguousarray(p1, np.float32), np.ascontiguousarray(p2, np.float32)
    st = np.zeros(n, np.uint8)
    vio_ct.oracle().ovio_ransac(C.byref(cfg), n, a32.ctypes.data, b32.ctypes.data, st.ctypes.data)
    # true epipolar distance (the error measure of cv::findFundamentalMat: distance to the epipolar line in both images)
    K = np.array([[f, 0, cx], [0, f, cy], [0, 0, 1]])
    Ft = np.linalg.inv(K).T @ _skew(t) @ R @ np.linalg.inv(K)
    h1, h2 = np.c_[a32.astype(float), np.ones(n)], np.c_[b32.astype(float), np.ones(n)]
    l2, l1 = h1 @ Ft.T, h2 @ Ft
    num = (h2 * l2).sum(1)
    d = np.maximum(np.abs(num) / np.hypot(l2[:, 0], l2[:, 1]), np.abs(num) / np.hypot(l1[:, 0], l1[:, 1]))
    # (the status comes from the best 7-point model, itself fitted to noisy points: a few true inliers fall outside its 1 px band)
    assert st[d < 0.5 * cfg.f_threshold].mean() > 0.9           # clear inliers of the true geometry are kept
    assert st[d > 3.0 * cfg.f_threshold].sum() == 0             # gross outliers are all rejected
    assert 0.6 * n < st.sum() < 0.85 * n
    Fe = _eight_point(a32[st > 0].astype(float), b32[st > 0].astype(float))
    Fe /= np.linalg.norm(Fe); Fn = Ft / np.linalg.norm(Ft)
    assert min(np.abs(Fe - Fn).max(), np.abs(Fe + Fn).max()) < 0.02


@pytest.mark.parametrize("seed", [4, 5, 6, 7])
def test_seven_point_against_numpy_nullspace_and_roots(seed):
    """The minimal solver inside the RANSAC (cv::findFundamentalMat's run7Point, restated with Gauss-Jordan elimination and a
    closed-form cubic in oracle/frontend.cpp) against the textbook formulation in numpy: null space of the 7 x 9 design matrix by SVD,
    det(F1 + l F2) = 0 by numpy.roots.  Compared up to scale: every real root has to appear on both sides, and every model has to
    satisfy the seven epipolar constraints and have rank 2."""
    rng = np.random.default_rng(seed)
    X = np.c_[rng.uniform(-1.5, 1.5, 7), rng.uniform(-1, 1, 7), rng.uniform(2.0, 6.0, 7)]
    R = scipy.linalg.expm(_skew(rng.normal(0, 0.05, 3)))
    t = rng.normal(0, 0.2, 3)
    X2 = X @ R.T + t
    x1, y1 = np.ascontiguousarray(X[:, 0] / X[:, 2]), np.ascontiguousarray(X[:, 1] / X[:, 2])
    x2, y2 = np.ascontiguousarray(X2[:, 0] / X2[:, 2]), np.ascontiguousarray(X2[:, 1] / X2[:, 2])
    Fo = np.zeros(27)
    L = vio_ct.oracle()
    L.ovio_seven_point.argtypes = [C.c_void_p] * 5
    n = L.ovio_seven_point(x1.ctypes.data, y1.ctypes.data, x2.ctypes.data, y2.ctypes.data, Fo.ctypes.data)
    assert 1 <= n <= 3
    Fo = Fo[:9 * n].reshape(n, 3, 3)
    # numpy: x2^T F x1 = 0 -> rows [x2 x1, x2 y1, x2, y2 x1, y2 y1, y2, x1, y1, 1]
    A = np.c_[x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, np.ones(7)]
    Vt = np.linalg.svd(A)[2]
    F1, F2 = Vt[-1].reshape(3, 3), Vt[-2].reshape(3, 3)
    # det(F1 + l F2) as a cubic in l through four evaluations
    ls = np.array([-1.0, 0.0, 1.0, 2.0])
    coef = np.polyfit(ls, [np.linalg.det(F1 + l * F2) for l in ls], 3)
    roots = np.roots(coef)
    real = [r.real for r in roots if abs(r.imag) < 1e-9 * max(1.0, abs(r.real))]
    Fn = [F1 + l * F2 for l in real]
    assert len(Fn) == n, (len(Fn), n)

    def unit(F):
        F = F / np.linalg.norm(F)
        k = np.argmax(np.abs(F))
        return F * np.sign(F.flat[k])
    for F in Fo:
        h1, h2 = np.c_[x1, y1, np.ones(7)], np.c_[x2, y2, np.ones(7)]
        assert np.abs(((h2 @ F) * h1).sum(1)).max() < 1e-9 * np.abs(F).max()          # the seven constraints
        assert abs(np.linalg.det(F / np.linalg.norm(F))) < 1e-10                         # rank 2
        assert min(np.abs(unit(F) - unit(G)).max() for G in Fn) < 1e-6, (unit(F), [unit(G) for G in Fn])
    # and the true fundamental matrix of the motion is one of them
    Ft = _skew(t) @ R
    assert min(np.abs(unit(F) - unit(Ft)).max() for F in Fo) < 1e-6


def _clahe_numpy(img, clip=3.0, tiles=8):
    """Independent (vectorised) restatement of cv::CLAHE for sizes that are multiples of `tiles`."""
    H, W = img.shape
    th, tw = H // tiles, W // tiles
    area = th * tw
    lim = max(int(clip * area / 256), 1)
    t = img.reshape(tiles, th, tiles, tw).transpose(0, 2, 1, 3).reshape(tiles * tiles, area)
    hist = np.stack([np.bincount(r, minlength=256) for r in t]).astype(np.int64)
    clipped = np.maximum(hist - lim, 0).sum(1)
    hist = np.minimum(hist, lim) + (clipped // 256)[:, None]
    for k, res in enumerate(clipped % 256):
        if res:
            step = max(256 // res, 1)
            idx = np.arange(0, 256, step)[:res]
            hist[k, idx] += 1
    scale = np.float32(255) / np.float32(area)
    lut = np.clip(np.rint(np.cumsum(hist, 1).astype(np.float32) * scale), 0, 255).astype(np.uint8).reshape(tiles, tiles, 256)
    f32 = np.float32
    tyf = np.arange(H, dtype=f32) * (f32(1) / f32(th)) - f32(0.5)
    txf = np.arange(W, dtype=f32) * (f32(1) / f32(tw)) - f32(0.5)
    ty1, tx1 = np.floor(tyf).astype(int), np.floor(txf).astype(int)
    ya, xa = (tyf - ty1.astype(f32))[:, None], (txf - tx1.astype(f32))[None, :]
    ya1, xa1 = f32(1) - ya, f32(1) - xa
    ty2, tx2 = np.minimum(ty1 + 1, tiles - 1)[:, None], np.minimum(tx1 + 1, tiles - 1)[None, :]
    ty1, tx1 = np.maximum(ty1, 0)[:, None], np.maximum(tx1, 0)[None, :]
    v = img.astype(int)
    g = lambda a, b: lut[a, b, v].astype(f32)
    res = (g(ty1, tx1) * xa1 + g(ty1, tx2) * xa) * ya1 + (g(ty2, tx1) * xa1 + g(ty2, tx2) * xa) * ya
    return np.clip(np.rint(res), 0, 255).astype(np.uint8)


def test_clahe_restatement_agrees_with_a_vectorised_one():
    """EQUALIZE (feature_tracker.cpp:269-275).  No OpenCV here: the oracle's loop restatement against an independent numpy one, plus the
    hand-computed case of a constant tile (4800 pixels of one value: clip 56, 4744 redistributed = 18 per bin + 1 on bins 0..135)."""
    L = vio_ct.oracle()
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[0:480, 0:640]
    imgs = [rng.integers(0, 256, (480, 640), dtype=np.uint8), (40 + 30 * np.sin(xx / 17.0) * np.cos(yy / 23.0)).astype(np.uint8),
            rng.integers(90, 110, (720, 1280), dtype=np.uint8), np.full((480, 848), 200, np.uint8)]
    for img in imgs:
        out = np.zeros_like(img)
        L.ovio_clahe(img.ctypes.data, img.shape[1], img.shape[0], out.ctypes.data)
        assert np.array_equal(out, _clahe_numpy(img))
    const = np.full((480, 640), 77, np.uint8)
    out = np.zeros_like(const)
    L.ovio_clahe(const.ctypes.data, 640, 480, out.ctypes.data)
    cdf77 = 18 * 78 + 78 + 56          # bins 0..77: 18 each, +1 each (all below 136), +56 in bin 77
    assert (out == int(np.rint(np.float32(cdf77) * (np.float32(255) / np.float32(4800))))).all()
