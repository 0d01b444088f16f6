"""VO mode (imu: 0, config/tum_rgbd/tum_fr3.yaml) behind the same C ABI against the oracle: LK with maxLevel 3 and no IMU prediction
(feature_tracker.cpp:307-311), no IMU factors, oldest pose constant (estimator.cpp:1182-1185), per-frame solvePnP initial guess
(FeatureManager::initFramePoseByPnP, feature_manager.cpp:590-642), poses handed back without the gauge fix (estimator.cpp:1060-1067)."""
import numpy as np
import pytest

import vio_ct

pytestmark = pytest.mark.gpu


def _vo_cfg(P, **kw):
    cfg = P.canonical_config(**kw)
    cfg.use_imu = 0
    cfg.lk_max_level = 3
    sc = vio_ct.synth_like(cfg)
    sc.t_static = 0.0
    return cfg, sc


def test_vo_tracker_bit_exact(P):
    """readImage(img, t) without relative_R: ids, counts and coordinates identical bit for bit over 12 frames (3 pyramid levels)."""
    cfg, sc = _vo_cfg(P)
    syn = P.Synth(sc)
    times = 1.0 + np.arange(12) * 0.1
    ot = vio_ct.OracleTracker(cfg)
    b = P.VioBatch(cfg, 1)
    for t in times:
        g, _ = syn.render_host(5, float(t))
        ot.read(g, t, None, True)
        b.track(g[None], [t], publish=True)
        a, q = ot.tracks(), b.tracks(0)
        assert np.array_equal(a[0], q[0]) and np.array_equal(a[1], q[1])
        for k in (2, 3, 4):
            assert np.array_equal(a[k].view(np.uint32), q[k].view(np.uint32)), (t, k)
    assert len(a[0]) >= 100 and a[1].max() >= 8


def test_vo_pipeline_matches_oracle(P):
    """Two moving-start sequences x 40 frames without any IMU, fix_depth 1 as in config/tum_rgbd/tum_fr3.yaml (with free depths and
    no IMU the scale is a gauge freedom of the window -- identical cost, different scale -- so nothing can be compared there):
    identical state-machine decisions and landmark counts, same tracks; window positions within 5e-4 m: the first solve agrees to
    1e-13 m, afterwards every frame starts from a solvePnP pose that the two implementations reach with 1e-7 differences (both stop on
    FLT_EPSILON) and the solver's function tolerance (1e-6 of the cost) turns those into 1e-5 .. 1e-4 m."""
    cfg, sc = _vo_cfg(P, fix_depth=1, depth_max=10.0)
    seqs, n = [3, 11], 40
    oruns = [vio_ct.run_oracle_sequence(cfg, sc, s, n) for s in seqs]
    b, traj, stat = vio_ct.run_hip_batch(P, cfg, sc, seqs, n, [o["frames"] for o in oruns])
    for i, s in enumerate(seqs):
        o = oruns[i]
        for f in range(n):
            so, sh = o["status"][f], stat[i][f]
            assert (int(so["solver_flag"]), int(so["frame_count"]), int(so["n_landmarks"])) == (sh.solver_flag, sh.frame_count, sh.n_landmarks), (s, f)
            if sh.solver_flag == 1 and sh.processed:
                assert int(so["marginalization_flag"]) == sh.marginalization_flag, (s, f)
                assert (int(so["n_residuals"]), int(so["n_var_landmarks"])) == (sh.n_residuals, sh.n_var_landmarks), (s, f)
        po = np.array([x[1] for x in o["traj"]]); ph = np.array([x[1] for x in traj[i]])
        assert po.shape == ph.shape and len(po) >= 20
        assert np.abs(po[0] - ph[0]).max() < 1e-9, (s, float(np.abs(po[0] - ph[0]).max()))
        assert np.abs(po - ph).max() < 5e-4, (s, float(np.abs(po - ph).max()))
        wo, wh = o["oracle"].window(), b.window(i)
        assert np.abs(wo[:, 7:16]).max() == 0 and np.abs(wh[:, 7:16]).max() == 0     # no speed / bias states in VO mode
        a, q = o["oracle"].tracks(), b.tracks(i)
        assert np.array_equal(a[0], q[0]) and np.array_equal(a[1], q[1])
        gt = np.array(o["gt"])
        ate_o, ate_h = vio_ct.ate_rmse(po, gt), vio_ct.ate_rmse(ph, gt)
        # (the VO world frame is the first camera frame, not gravity-aligned: the yaw-only alignment of ate_rmse leaves a few cm)
        assert ate_h < 0.08 and abs(ate_h - ate_o) <= max(0.02 * ate_o, 3e-4), (ate_o, ate_h)


def test_device_solvepnp_matches_oracle(P, orc):
    """The device restatement of cv::solvePnP(ITERATIVE) (block-cooperative CvLevMarq) against the oracle's on synthetic 3-D / 2-D
    pairs with pixel-level noise and a perturbed initial guess: same pose to 1e-7 (both stop on FLT_EPSILON relative parameter change)."""
    import ctypes as C
    orc.ovio_solve_pnp_iterative.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(7)
    for trial in range(6):
        n = [40, 150, 8, 220, 60, 4][trial]
        X = np.c_[rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(2, 6, n)]
        rv = rng.normal(0, 0.2, 3); th = np.linalg.norm(rv); k = rv / th
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
        t = rng.normal(0, 0.3, 3)
        Y = (R @ X.T).T + t
        img = Y[:, :2] / Y[:, 2:3] + rng.normal(0, 1.0 / 460, (n, 2))
        # perturbed start
        rv0 = rv + rng.normal(0, 0.05, 3); t0 = t + rng.normal(0, 0.1, 3)
        th0 = np.linalg.norm(rv0); k0 = rv0 / th0
        K0 = np.array([[0, -k0[2], k0[1]], [k0[2], 0, -k0[0]], [-k0[1], k0[0], 0]])
        R0 = np.ascontiguousarray(np.eye(3) + np.sin(th0) * K0 + (1 - np.cos(th0)) * K0 @ K0)
        Ro, to = R0.copy(), t0.copy()
        obj = np.ascontiguousarray(X); im = np.ascontiguousarray(img)
        assert orc.ovio_solve_pnp_iterative(n, obj.ctypes.data, im.ctypes.data, Ro.ctypes.data, to.ctypes.data) == 1
        rd, td = rv0.copy(), t0.copy()
        assert P.lib().vio_stage_pnp(n, obj.ctypes.data, im.ctypes.data, rd.ctypes.data, td.ctypes.data) == 0
        thd = np.linalg.norm(rd); kd = rd / thd
        Kd = np.array([[0, -kd[2], kd[1]], [kd[2], 0, -kd[0]], [-kd[1], kd[0], 0]])
        Rd = np.eye(3) + np.sin(thd) * Kd + (1 - np.cos(thd)) * Kd @ Kd
        assert np.abs(Rd - Ro).max() < 1e-7 and np.abs(td - to).max() < 1e-7, (trial, float(np.abs(Rd - Ro).max()), float(np.abs(td - to).max()))
        assert np.abs(Rd - R).max() < 0.02
