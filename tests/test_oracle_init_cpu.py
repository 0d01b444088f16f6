"""Known-answer tests for the oracle's visual-inertial alignment (dynamic initialisation, SURVEY.md 8f rank 1, oracle/initial.cpp):
LinearAlignmentWithDepth / RefineGravityWithDepth / TangentBasis (initial_aligment.cpp:78-91, 170-244, 337-405) and the state
hand-over at the end of visualInitialAlignWithDepth (estimator.cpp:839-869).  The inputs are built from an analytic trajectory, so
the expected gravity vector and body velocities are known in closed form.  CPU only."""
import ctypes as C

import numpy as np
import pytest

import vio_ct

G = 9.81


def rot(axis, a):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


def trajectory(t):
    """body pose / velocity in a gravity-aligned world (z up); smooth, with rotation about all axes"""
    p = np.array([0.8 * np.sin(0.9 * t), 0.5 * np.cos(0.7 * t) - 0.5, 0.3 * np.sin(1.3 * t)])
    v = np.array([0.8 * 0.9 * np.cos(0.9 * t), -0.5 * 0.7 * np.sin(0.7 * t), 0.3 * 1.3 * np.cos(1.3 * t)])
    R = rot([0, 0, 1], 0.4 * t + 0.3) @ rot([0, 1, 0], 0.25 * np.sin(1.1 * t)) @ rot([1, 0, 0], 0.2 * np.cos(0.8 * t))
    return p, v, R


def make_frames(n, dt, tic, R_cw, t0, noise=0.0, seed=0):
    """ImageFrame-like records: R = body rotation in the SfM frame c, T = camera position in it, exact pre-integration deltas"""
    rng = np.random.default_rng(seed)
    gw = np.array([0, 0, G])
    rows, truth = [], []
    prev = None
    for k in range(n):
        t = 0.3 + k * dt
        p, v, R = trajectory(t)
        Rc = R_cw @ R
        Tc = R_cw @ (p + R @ tic) + t0
        if prev is None:
            sdt, dp, dv = 0.0, np.zeros(3), np.zeros(3)
        else:
            pp, pv, pR = prev
            sdt = dt
            dp = pR.T @ (p - pp - pv * dt + 0.5 * gw * dt * dt) + noise * rng.standard_normal(3)
            dv = pR.T @ (v - pv + gw * dt) + noise * rng.standard_normal(3)
        rows.append(np.concatenate([Rc.ravel(), Tc, [sdt], dp, dv]))
        truth.append((p, v, R))
        prev = (p, v, R)
    return np.ascontiguousarray(np.array(rows)), truth


@pytest.fixture(scope="module")
def orc():
    L = vio_ct.oracle()
    L.ovio_linear_alignment_with_depth.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    L.ovio_linear_alignment_with_depth.restype = C.c_int
    L.ovio_align_window_to_gravity.argtypes = [C.c_int] + [C.c_void_p] * 6
    L.ovio_tangent_basis.argtypes = [C.c_void_p] * 3
    return L


def solve(orc, frames, tic):
    n = len(frames)
    g = np.zeros(3)
    x = np.zeros(3 * n + 3)
    tic = np.ascontiguousarray(tic, float)
    ok = orc.ovio_linear_alignment_with_depth(n, frames.ctypes.data, tic.ctypes.data, G, g.ctypes.data, x.ctypes.data)
    return ok, g, x


@pytest.mark.parametrize("g0", [[0.3, -0.2, 9.7], [0, 0, 1.0], [0, 0, -5.0], [4.0, 4.0, 0.0]])
def test_tangent_basis_is_an_orthonormal_complement(orc, g0):
    g0 = np.array(g0, float)
    b, c = np.zeros(3), np.zeros(3)
    orc.ovio_tangent_basis(g0.ctypes.data, b.ctypes.data, c.ctypes.data)
    a = g0 / np.linalg.norm(g0)
    if a[0] == 0 and a[1] == 0 and a[2] == -1:
        # reference quirk kept: only a == +e_z switches the helper axis, for a == -e_z the projection of e_z vanishes and
        # normalized() divides by zero (initial_aligment.cpp:82-85)
        assert np.isnan(b).all() and np.isnan(c).all()
        return
    M = np.stack([a, b, c])
    assert np.abs(M @ M.T - np.eye(3)).max() < 1e-14
    assert abs(np.linalg.det(M) - 1) < 1e-14          # c = a x b: right-handed
    if not (a[0] == 0 and a[1] == 0 and a[2] == 1):
        assert abs(b @ np.array([0, 0, 1.0]) - np.sqrt(1 - a[2] ** 2)) < 1e-14   # b = normalised projection of e_z
    else:
        assert np.array_equal(b, [1, 0, 0])             # the a == e_z special case switches to e_x


@pytest.mark.parametrize("n,dt", [(11, 0.1), (25, 0.05), (4, 0.2)])
def test_alignment_recovers_gravity_and_velocities(orc, n, dt):
    tic = np.array([0.05, -0.02, 0.1])
    R_cw = rot([1, 2, -1], 0.9)                       # arbitrary SfM reference frame
    frames, truth = make_frames(n, dt, tic, R_cw, np.array([0.4, -1.0, 2.0]))
    ok, g, x = solve(orc, frames, tic)
    assert ok == 1
    assert abs(np.linalg.norm(g) - G) < 1e-12          # RefineGravity keeps |g| = G exactly
    assert np.abs(g - R_cw @ np.array([0, 0, G])).max() < 1e-8
    for k, (p, v, R) in enumerate(truth):
        assert np.abs(x[3 * k:3 * k + 3] - R.T @ v).max() < 1e-8, k   # velocity of frame k in ITS body frame
    assert np.abs(x[3 * n:3 * n + 2]).max() < 1e-6      # last tangent-plane correction: already converged


def test_alignment_degrades_gracefully_with_noise_and_rejects_wrong_gravity(orc):
    tic = np.array([0.0, 0.0, 0.0])
    R_cw = rot([0, 1, 0], -0.6)
    frames, truth = make_frames(21, 0.1, tic, R_cw, np.zeros(3), noise=2e-3, seed=3)
    ok, g, x = solve(orc, frames, tic)
    assert ok == 1
    ang = np.degrees(np.arccos(np.clip(g @ (R_cw @ np.array([0, 0, G])) / G / G, -1, 1)))
    assert ang < 1.0
    v_err = max(np.abs(x[3 * k:3 * k + 3] - R.T @ v).max() for k, (p, v, R) in enumerate(truth))
    assert v_err < 0.1
    # pre-integration that corresponds to |g| = 2 G: the linear solution is more than 1 m/s^2 away from G -> the reference returns false
    bad = frames.copy()
    gw = np.array([0, 0, G])
    for k in range(1, len(bad)):
        pR = truth[k - 1][2]
        dt = bad[k, 12]
        bad[k, 13:16] += pR.T @ (0.5 * gw * dt * dt)
        bad[k, 16:19] += pR.T @ (gw * dt)
    ok2, g2, _ = solve(orc, bad, tic)
    assert ok2 == 0 and abs(np.linalg.norm(g2) - 2 * G) < 0.05


def test_window_hand_over_is_gravity_aligned_with_zero_yaw(orc):
    n, dt = 11, 0.1
    tic = np.array([0.05, -0.02, 0.1])
    R_cw = rot([1, 2, -1], 0.9)
    frames, truth = make_frames(n, dt, tic, R_cw, np.array([0.4, -1.0, 2.0]))
    ok, g, x = solve(orc, frames, tic)
    Ps = np.ascontiguousarray(frames[:, 9:12].copy())
    Rs = np.ascontiguousarray(frames[:, :9].copy())
    Vs = np.zeros((n, 3))
    gg = g.copy()
    orc.ovio_align_window_to_gravity(n, Ps.ctypes.data, Rs.ctypes.data, Vs.ctypes.data, x.ctypes.data, tic.ctypes.data, gg.ctypes.data)
    assert np.abs(gg - np.array([0, 0, G])).max() < 1e-8
    assert np.abs(Ps[0]).max() == 0.0
    R0 = Rs[0].reshape(3, 3)
    assert abs(np.arctan2(R0[1, 0], R0[0, 0])) < 1e-9                  # yaw of frame 0 removed (utility.h R2ypr convention)
    # the composite map world -> output frame is a rotation about z: heights, lengths and vertical speeds are those of the truth
    p0 = truth[0][0]
    for k, (p, v, R) in enumerate(truth):
        assert abs(Ps[k][2] - (p - p0)[2]) < 1e-8
        assert abs(np.linalg.norm(Ps[k]) - np.linalg.norm(p - p0)) < 1e-8
        assert abs(Vs[k][2] - v[2]) < 1e-8 and abs(np.linalg.norm(Vs[k]) - np.linalg.norm(v)) < 1e-8
        Rk = Rs[k].reshape(3, 3)
        assert np.abs(Rk @ Rk.T - np.eye(3)).max() < 1e-12
        # relative rotation between window frames is untouched
        assert np.abs(R0.T @ Rk - truth[0][2].T @ R).max() < 1e-9


# ------------------------------------------------------------------------------------------------ SfM front (restated OpenCV / Ceres)
@pytest.fixture(scope="module")
def sfm(orc):
    orc.ovio_solve_pnp_iterative.argtypes = [C.c_int] + [C.c_void_p] * 4
    orc.ovio_solve_pnp_ransac_epnp.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    orc.ovio_sfm_window.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 8
    return orc


def scene(n, seed):
    rng = np.random.default_rng(seed)
    X = np.c_[rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(2, 7, n)]
    Rt, tt = rot([0.3, -1, 0.2], 0.25), np.array([0.3, -0.1, 0.2])
    Y = X @ Rt.T + tt
    return rng, np.ascontiguousarray(X), np.ascontiguousarray(Y[:, :2] / Y[:, 2:3]), Rt, tt


def test_pnp_refinement_converges_from_a_rough_guess(sfm):
    """cv::solvePnP(ITERATIVE, useExtrinsicGuess) restated: Levenberg-Marquardt from the guess to the exact pose (the inputs go
    through float32 like cv::Point3f / Point2f, hence 1e-6 rather than 1e-12)."""
    rng, X, m, Rt, tt = scene(40, 1)
    R = np.ascontiguousarray(rot([0.3, -1, 0.2], 0.10))
    t = np.array([0.15, 0.05, 0.05])
    assert sfm.ovio_solve_pnp_iterative(len(X), X.ctypes.data, m.ctypes.data, R.ctypes.data, t.ctypes.data) == 1
    assert np.abs(R - Rt).max() < 1e-6 and np.abs(t - tt).max() < 1e-6
    assert np.abs(R @ R.T - np.eye(3)).max() < 1e-12
    # fewer than 4 points: cv::solvePnP asserts; the restatement reports failure
    assert sfm.ovio_solve_pnp_iterative(3, X.ctypes.data, m.ctypes.data, R.ctypes.data, t.ctypes.data) == 0


def test_epnp_ransac_rejects_outliers_and_recovers_the_pose(sfm):
    rng, X, m, Rt, tt = scene(60, 5)
    bad = rng.choice(60, 15, replace=False)
    m2 = m.copy()
    m2[bad] += rng.uniform(0.05, 0.3, (15, 2)) * rng.choice([-1, 1], (15, 2))
    R, t, inl = np.zeros((3, 3)), np.zeros(3), np.zeros(60, np.uint8)
    ok = sfm.ovio_solve_pnp_ransac_epnp(60, X.ctypes.data, m2.ctypes.data, 100, 1 / 460, 0.99, R.ctypes.data, t.ctypes.data, inl.ctypes.data)
    assert ok == 1
    assert inl[bad].sum() == 0 and inl.sum() == 45          # every outlier is 23 px or more away, every clean point is exact
    assert np.abs(R - Rt).max() < 1e-6 and np.abs(t - tt).max() < 1e-6
    # deterministic: cv::RNG(-1) restated, the same subsets every call
    R2, t2, inl2 = np.zeros((3, 3)), np.zeros(3), np.zeros(60, np.uint8)
    sfm.ovio_solve_pnp_ransac_epnp(60, X.ctypes.data, m2.ctypes.data, 100, 1 / 460, 0.99, R2.ctypes.data, t2.ctypes.data, inl2.ctypes.data)
    assert np.array_equal(R, R2) and np.array_equal(t, t2) and np.array_equal(inl, inl2)
    # fewer points than the 5-point EPnP model
    assert sfm.ovio_solve_pnp_ransac_epnp(4, X.ctypes.data, m.ctypes.data, 100, 1 / 460, 0.99, R.ctypes.data, t.ctypes.data, inl.ctypes.data) == 0


def window_tracks(W, nf, step, noise_px, depth_noise, seed, rot_rate=0.03):
    """feature tracks of a camera that moves sideways by `step` metres per frame with a slow rotation"""
    rng = np.random.default_rng(seed)
    Rwc = [rot([0, 1, 0], rot_rate * k) @ rot([1, 0, 0], rot_rate / 3 * k) for k in range(W + 1)]
    pwc = [np.array([step * k, 0.15 * step * np.sin(k), 0.25 * step * k]) for k in range(W + 1)]
    X = np.c_[rng.uniform(-3, 4, nf), rng.uniform(-2, 2, nf), rng.uniform(2.5, 8, nf)]
    start, nobs, obs = [], [], []
    for i in range(nf):
        s = int(rng.integers(0, 3))
        e = W if rng.random() < 0.8 else int(rng.integers(s + 1, W + 1))
        o = []
        for k in range(s, e + 1):
            Y = Rwc[k].T @ (X[i] - pwc[k])
            if Y[2] < 0.2:
                break
            nz = rng.normal(0, noise_px / 460, 2) if noise_px > 0 else np.zeros(2)
            o.append([Y[0] / Y[2] + nz[0], Y[1] / Y[2] + nz[1], Y[2] * (1 + (rng.normal(0, depth_noise) if depth_noise > 0 else 0))])
        if len(o) >= 2:
            start.append(s); nobs.append(len(o)); obs += o
    return (np.array(start, np.int32), np.array(nobs, np.int32), np.ascontiguousarray(np.array(obs)), Rwc, pwc, X)


def run_sfm(sfm, W, start, nobs, obs):
    nf = len(start)
    l = C.c_int(-1)
    q, T, pts, st = np.zeros((W + 1, 4)), np.zeros((W + 1, 3)), np.zeros((nf, 4)), np.zeros(4)
    rc = sfm.ovio_sfm_window(W, nf, start.ctypes.data, nobs.ctypes.data, obs.ctypes.data, C.byref(l), q.ctypes.data, T.ctypes.data,
                             pts.ctypes.data, st.ctypes.data)
    return rc, l.value, q, T, pts, st


def q2R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


@pytest.mark.parametrize("noise_px,depth_noise,tol_R,tol_T", [(0.0, 0.0, 2e-6, 2e-5), (0.3, 0.005, 4e-3, 2e-2)])
def test_sfm_window_recovers_the_camera_poses(sfm, noise_px, depth_noise, tol_R, tol_T):
    """relativePose + GlobalSFM::construct (PnP chain, depth-checked triangulation, bundle adjustment): poses of all window frames
    in the frame of camera l.  Exact data: limited by the float32 round trip of the PnP inputs; noisy data: at the noise level."""
    W = 10
    start, nobs, obs, Rwc, pwc, X = window_tracks(W, 160, 0.12, noise_px, depth_noise, 2)
    rc, l, q, T, pts, st = run_sfm(sfm, W, start, nobs, obs)
    assert rc == 0
    assert l == 0                                   # first frame whose parallax to the newest frame exceeds 30 px
    assert pts[:, 0].sum() > 0.9 * len(start)
    assert st[3] == 1 or st[2] < 5e-3               # ceres CONVERGENCE or final_cost < 5e-3 (initial_sfm.cpp:385)
    assert st[2] <= st[1] * (1 + 1e-12)
    for k in range(W + 1):
        Rrel, prel = Rwc[l].T @ Rwc[k], Rwc[l].T @ (pwc[k] - pwc[l])
        assert np.abs(q2R(q[k]) - Rrel).max() < tol_R, k
        assert np.abs(T[k] - prel).max() < tol_T, k
        assert abs(np.linalg.norm(q[k]) - 1) < 1e-9


def test_sfm_window_needs_parallax_and_correspondences(sfm):
    W = 10
    # 4 mm and 0.5 mrad per frame: < 30 px of (not rotation-compensated) parallax between any frame and the newest one
    start, nobs, obs, *_ = window_tracks(W, 160, 0.004, 0.0, 0.0, 3, rot_rate=0.0005)
    rc, *_ = run_sfm(sfm, W, start, nobs, obs)
    assert rc == 1
    start, nobs, obs, *_ = window_tracks(W, 18, 0.12, 0.0, 0.0, 4)      # at most 18 correspondences (> 20 required)
    rc, *_ = run_sfm(sfm, W, start, nobs, obs)
    assert rc == 1


# ------------------------------------------------------------------------------------------------ end to end (oracle pipeline)
@pytest.mark.parametrize("seq", [3, 11])
def test_dynamic_initialisation_end_to_end(P, seq):
    """static_init: 0 branch of processImage (estimator.cpp:230-259) in the oracle: a sequence that moves from the first frame
    initialises through SfM + visual-inertial alignment as soon as the window is full and then tracks at least as well as the
    static branch does on the same data.  (vio_config.dynamic_init carries the oracle-only switch; the product rejects such
    configurations, see dataio.config_from_yaml.)"""
    res = {}
    for dyn in (1, 0):
        cfg = P.canonical_config()
        cfg.dynamic_init = dyn
        sc = vio_ct.synth_like(cfg)
        sc.t_static = 0.0
        syn = P.Synth(sc)
        F = 40
        o = vio_ct.OraclePipeline(cfg)
        o.push_imu(*syn.imu(seq, F * 20 + 64))
        traj, gt, first, v_err = [], [], None, None
        for f, t in enumerate(vio_ct.frame_times(sc, F)):
            g, d = syn.render_host(seq, float(t))
            r = o.feed(g, d, float(t))
            st = o.status()
            if st["solver_flag"] == 1 and first is None:
                first = f
            if st["solver_flag"] == 1 and r == 1:
                traj.append(o.window()[cfg.window_size, :3].copy())
                gt.append(syn.pose(seq, float(t))[0])
        res[dyn] = (first, vio_ct.ate_rmse(np.array(traj), np.array(gt)), len(traj))
    first, ate, n = res[1]
    assert first == 13                      # first-image skip + init_pub + init_feature + 11 window frames
    assert n == 40 - 13
    assert ate < 0.03
    assert ate < 1.5 * res[0][1] + 0.005    # not worse than the static branch on the same moving-start data


def test_dynamic_initialisation_waits_for_parallax(P):
    """A sequence that rests for 1.5 s: the first full windows have no parallax (relativePose fails, the window keeps sliding and
    all_image_frame grows beyond the window: the per-image-frame solvePnP loop and the erase in slideWindow are exercised); the
    initialisation succeeds once the camera has moved."""
    cfg = P.canonical_config()
    cfg.dynamic_init = 1
    sc = vio_ct.synth_like(cfg)
    sc.t_static = 1.5
    syn = P.Synth(sc)
    F, seq = 45, 3
    o = vio_ct.OraclePipeline(cfg)
    o.push_imu(*syn.imu(seq, F * 20 + 64))
    traj, gt, first = [], [], None
    for f, t in enumerate(vio_ct.frame_times(sc, F)):
        g, d = syn.render_host(seq, float(t))
        r = o.feed(g, d, float(t))
        st = o.status()
        if st["solver_flag"] == 1 and first is None:
            first = f
        if st["solver_flag"] == 1 and r == 1:
            traj.append(o.window()[cfg.window_size, :3].copy())
            gt.append(syn.pose(seq, float(t))[0])
    assert first is not None and 15 < first < 30
    assert vio_ct.ate_rmse(np.array(traj), np.array(gt)) < 0.03
