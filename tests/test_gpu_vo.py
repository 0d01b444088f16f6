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


@pytest.mark.parametrize("fix_depth", [0, 1])
def test_vo_pipeline_matches_oracle(P, fix_depth):
    """Two moving-start sequences x 40 frames without any IMU: identical state-machine decisions and landmark counts, window
    positions within 2e-5 m (the per-frame solvePnP start makes the solves a little more sensitive than in IMU mode), same tracks."""
    cfg, sc = _vo_cfg(P, fix_depth=fix_depth, depth_max=10.0)
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
        assert np.abs(po - ph).max() < 2e-5, (s, float(np.abs(po - ph).max()))
        wo, wh = o["oracle"].window(), b.window(i)
        assert np.abs(wo[:, 7:16]).max() == 0 and np.abs(wh[:, 7:16]).max() == 0     # no speed / bias states in VO mode
        a, q = o["oracle"].tracks(), b.tracks(i)
        assert np.array_equal(a[0], q[0]) and np.array_equal(a[1], q[1])
        gt = np.array(o["gt"])
        assert vio_ct.ate_rmse(ph, gt) < 0.03
