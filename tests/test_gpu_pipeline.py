"""GPU parity of the whole hot path against the CPU oracle on identical rendered frames + IMU."""
import numpy as np
import pytest

import vio_ct

pytestmark = pytest.mark.gpu


def _frames(P, sc, seq, times):
    syn = P.Synth(sc)
    return [syn.render_host(seq, t) for t in times]


def test_frontend_bit_exact(P):
    """FeatureTracker::readImage x 14 frames without IMU (relative_R = I on both sides): ids, track_cnt, pixel and
    normalised coordinates and velocities must be identical bit for bit."""
    cfg = P.default_config()
    sc = vio_ct.synth_like(cfg)
    times = 2.0 + np.arange(14) * 0.1
    fr = _frames(P, sc, 5, times)
    ot = vio_ct.OracleTracker(cfg)
    b = P.VioBatch(cfg, 1)
    for (g, d), t in zip(fr, times):
        ot.read(g, t, None, True)
        b.track(g, [t], publish=True)
        a = ot.tracks()
        q = b.tracks(0)
        assert len(a[0]) == len(q[0]), (t, len(a[0]), len(q[0]))
        assert np.array_equal(a[0], q[0]) and np.array_equal(a[1], q[1])
        for k in (2, 3, 4):
            assert np.array_equal(a[k].view(np.uint32), q[k].view(np.uint32)), (t, k, float(np.abs(a[k] - q[k]).max()))
    assert len(a[0]) >= 100 and a[1].max() >= 8  # features survive: the test exercises LK, RANSAC, mask and grid-FAST


def _run_hip(P, cfg, sc, seqs, n_frames, frames, tracks=None):
    """tracks: optional list per sequence that receives the tracker's (ids, track_cnt, cur_pts, ...) after every frame"""
    syn = P.Synth(sc)
    S = len(seqs)
    b = P.VioBatch(cfg, S)
    nimu = int(n_frames / sc.cam_rate * sc.imu_rate) + 64
    imu = [syn.imu(s, nimu) for s in seqs]
    k = [0] * S
    H, W = cfg.height, cfg.width
    traj = [[] for _ in seqs]
    stat = [[] for _ in seqs]
    for f, tf in enumerate(vio_ct.frame_times(sc, n_frames)):
        for i in range(S):
            ti, ai, gi = imu[i]
            k2 = vio_ct.imu_until(ti, k[i], tf, sc.imu_rate)
            if k2 > k[i]:
                b.push_imu(i, ti[k[i]:k2], ai[k[i]:k2], gi[k[i]:k2])
            k[i] = k2
        gray = np.stack([frames[i][f][0] for i in range(S)])
        depth = np.stack([frames[i][f][1] for i in range(S)])
        b.feed(gray, depth, [tf] * S)
        for i in range(S):
            st = b.status(i)
            stat[i].append(st)
            if tracks is not None:
                tracks[i].append(b.tracks(i))
            if st.solver_flag == 1 and st.processed:
                w = b.window(i)
                traj[i].append((f, w[cfg.window_size, :3].copy(), w[cfg.window_size, 3:7].copy(), w[cfg.window_size, 7:10].copy()))
    return b, traj, stat


MIN_IDENTICAL_FRAMES = 18   # frames (static start + the first solved ones) over which the tracker's float positions equal the oracle's bit for bit


@pytest.mark.parametrize("variant", ["fix_depth", "free_depth_td", "relanded_ids"])
def test_pipeline_matches_oracle(P, variant):
    """vio_feed on 2 sequences x 40 frames vs the oracle on the same frames: identical state machine decisions,
    window poses within 1e-5 m / 1e-5 rad-equivalent, ATE of both within 3 cm of ground truth and within 1 % of each other
    (north-star tolerance) or 0.2 mm absolute."""
    kw = dict(fix_depth=1) if variant == "fix_depth" else dict(fix_depth=0, depth_max=10.0, estimate_td=int(variant == "free_depth_td"))
    cfg = P.default_config(**kw)
    sc = vio_ct.synth_like(cfg)
    seqs, n_frames = [0, 7], 40
    if variant == "relanded_ids":
        # sequence 4: outlier rejection drops 145 of 193 landmarks at frame 14 while the tracker keeps their ids; they come back as new
        # landmarks at the END of the list (ids no longer ascending in list order) -- regression test for the id -> slot lookup
        seqs, n_frames = [4], 30
    tr_o = [[] for _ in seqs]
    oruns = [vio_ct.run_oracle_sequence(cfg, sc, s, n_frames, hook=lambda f, orc, i=i: tr_o[i].append(orc.tracks())) for i, s in enumerate(seqs)]
    frames = [r["frames"] for r in oruns]
    tr_h = [[] for _ in seqs]
    b, traj, stat = _run_hip(P, cfg, sc, seqs, n_frames, frames, tracks=tr_h)
    for i, s in enumerate(seqs):
        o = oruns[i]
        assert len(traj[i]) == len(o["traj"]) > 15
        for f in range(n_frames):
            so, sh = o["status"][f], stat[i][f]
            assert int(so["solver_flag"]) == sh.solver_flag and int(so["frame_count"]) == sh.frame_count, f
            assert int(so["n_landmarks"]) == sh.n_landmarks, (f, so["n_landmarks"], sh.n_landmarks)
            if sh.solver_flag == 1 and sh.processed:
                assert int(so["marginalization_flag"]) == sh.marginalization_flag, f
        po = np.array([x[1] for x in o["traj"]]); ph = np.array([x[1] for x in traj[i]])
        qo = np.array([x[2] for x in o["traj"]]); qh = np.array([x[2] for x in traj[i]])
        assert np.abs(po - ph).max() < 1e-5, float(np.abs(po - ph).max())
        assert np.abs(np.abs((qo * qh).sum(1)) - 1).max() < 1e-9
        gt = np.array(o["gt"])
        ate_o, ate_h = vio_ct.ate_rmse(po, gt), vio_ct.ate_rmse(ph, gt)
        assert ate_o < 0.03 and ate_h < 0.03
        assert abs(ate_h - ate_o) <= max(0.01 * ate_o, 2e-4)
        # final feature tracks identical
        a, q = o["oracle"].tracks(), b.tracks(i)
        assert np.array_equal(a[0], q[0]) and np.array_equal(a[1], q[1])
        # round 5: predictMotion is bit-reproducible (one polynomial sin / cos on both sides, csrc/dmath.h = oracle/om.h), so the tracker's inputs
        # differ only through latest_Bg, i.e. through what the two back-ends disagree on.  While that is round-off (the static start and the
        # first solved frames) the tracked positions are the SAME FLOATS; once the back-ends are ~1e-9 apart a predicted point lands on the other
        # side of a float rounding boundary now and then (6e-8 px against an ulp of 3e-5 px: about one coordinate in 500) and LK, which stops at
        # 0.01 px, ends within 5e-3 px.
        first_diff = next((f for f in range(n_frames) if not np.array_equal(tr_o[i][f][2].view(np.uint32), tr_h[i][f][2].view(np.uint32))), n_frames)
        print("sequence %d (%s): tracked positions bit-identical through frame %d of %d" % (s, variant, first_diff - 1, n_frames))
        assert first_diff >= MIN_IDENTICAL_FRAMES, first_diff
        assert np.abs(a[2] - q[2]).max() < 5e-3


@pytest.mark.parametrize("seq", [781, 783, 730, 5])
def test_pipeline_equals_the_oracle_with_matched_formulations_to_round_off(P, monkeypatch, seq):
    """Round 5: the oracle with every equivalent formulation of the HIP path switched on (OVIO_DEVIATIONS = 31, oracle/oracle.h
    ODEV_*: IMU whitening by chol(cov)^-1, quadratic-form prior, analytic landmark elimination, Cholesky inverse of the remaining 15 x 15 block,
    frame-pair projection factors) and the HIP path on IDENTICAL frames: what is left is summation order, and the trajectories agree to 1e-11 ..
    6e-10 m over 50 frames (bar: 5e-9), with every solver decision equal.  Against the oracle as the reference formulates these steps the same
    run agrees to 1e-9 .. 1e-5 (test_pipeline_matches_oracle): the difference between the two is the arithmetic noise of the reference's own
    eigen-decompositions (profiles/round5_deviation_attribution.json), not an error of the HIP path.  Sequences 781 / 783 / 730 are the ones that
    separated EARLIEST in the 128-sequence runs of rounds 3 - 4 -- because those runs compared device-rendered with host-rendered frames."""
    monkeypatch.setenv("OVIO_DEVIATIONS", "31")
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    n_frames = 50
    try:
        o = vio_ct.run_oracle_sequence(cfg, sc, seq, n_frames)
    finally:
        monkeypatch.delenv("OVIO_DEVIATIONS")
        vio_ct.oracle().ovio_set_deviations(0)     # (the free factor functions read the mask of the last estimator constructed)
    b, traj, stat = _run_hip(P, cfg, sc, [seq], n_frames, [o["frames"]])
    assert len(traj[0]) == len(o["traj"]) >= 30
    for f in range(n_frames):
        so, sh = o["status"][f], stat[0][f]
        assert (int(so["solver_flag"]), int(so["frame_count"]), int(so["n_landmarks"])) == (sh.solver_flag, sh.frame_count, sh.n_landmarks), f
        if sh.solver_flag == 1 and sh.processed:
            assert (int(so["iterations"]), int(so["successful_steps"]), int(so["n_residuals"]), int(so["marginalization_flag"])) == \
                   (sh.iterations, sh.successful_steps, sh.n_residuals, sh.marginalization_flag), f
    po = np.array([x[1] for x in o["traj"]]); ph = np.array([x[1] for x in traj[0]])
    worst = float(np.abs(po - ph).max())
    print("sequence %d: HIP vs oracle with matched formulations, %d solved frames: max |dP| = %.2e m" % (seq, len(po), worst))
    assert worst < 5e-9, worst      # measured 1e-11 .. 6e-10 over the four sequences (the reference-formulation oracle: 1e-9 .. 1e-5)
    a, q = o["oracle"].tracks(), b.tracks(0)
    assert np.array_equal(a[0], q[0]) and np.array_equal(a[2].view(np.uint32), q[2].view(np.uint32))   # the trackers stay the same floats throughout


def test_pipeline_config5_shape(P):
    """BASELINE configs[4] shape on one sequence: 1280x720, 300 features, 20-keyframe window, 7x8 grid (intrinsics scaled x2 / x1.5).
    Exercises the large-window code paths (Schur complement / Cholesky in HBM instead of LDS tiles, 210 frame pairs, n_prior = 136).
    Same bar as the canonical configuration: identical decisions, window positions within 1e-5 m of the oracle."""
    cfg = P.canonical_config(width=1280, height=720, max_cnt=300, window_size=20, grid_rows=7, grid_cols=8, max_landmarks=2048,
                             fx=604.5821781259577 * 2, fy=604.2544712985845 * 1.5, cx=321.2638233484251 * 2, cy=239.70969315130674 * 1.5)
    sc = vio_ct.synth_like(cfg)
    seqs, n_frames = [1], 34
    oruns = [vio_ct.run_oracle_sequence(cfg, sc, s, n_frames) for s in seqs]
    b, traj, stat = _run_hip(P, cfg, sc, seqs, n_frames, [r["frames"] for r in oruns])
    o = oruns[0]
    assert len(traj[0]) == len(o["traj"]) >= 8
    for f in range(n_frames):
        so, sh = o["status"][f], stat[0][f]
        assert int(so["solver_flag"]) == sh.solver_flag and int(so["frame_count"]) == sh.frame_count, f
        assert int(so["n_landmarks"]) == sh.n_landmarks, (f, so["n_landmarks"], sh.n_landmarks)
        if sh.solver_flag == 1 and sh.processed:
            assert int(so["n_residuals"]) == sh.n_residuals and int(so["marginalization_flag"]) == sh.marginalization_flag, f
    po = np.array([x[1] for x in o["traj"]]); ph = np.array([x[1] for x in traj[0]])
    assert np.abs(po - ph).max() < 1e-5, float(np.abs(po - ph).max())
    assert vio_ct.ate_rmse(ph, np.array(o["gt"])) < 0.03


def test_pipeline_848x480_grid7x8(P):
    """The reference's own 150-feature configuration shape (config/realsense/vio_campus.yaml: 848x480, 7x8 grid, min_dist 15):
    non-square cell sizes, a width that is not a multiple of the pyramid tile, 56 FAST cells."""
    cfg = P.canonical_config(width=848, height=480, grid_rows=7, grid_cols=8, fx=430.0, fy=430.0, cx=424.0, cy=240.0,
                             k1=0.0, k2=0.0, p1=0.0, p2=0.0)
    sc = vio_ct.synth_like(cfg)
    seqs, n_frames = [31], 30
    oruns = [vio_ct.run_oracle_sequence(cfg, sc, s, n_frames) for s in seqs]
    b, traj, stat = _run_hip(P, cfg, sc, seqs, n_frames, [r["frames"] for r in oruns])
    o = oruns[0]
    assert len(traj[0]) == len(o["traj"]) >= 10
    for f in range(n_frames):
        so, sh = o["status"][f], stat[0][f]
        assert (int(so["solver_flag"]), int(so["frame_count"]), int(so["n_landmarks"])) == (sh.solver_flag, sh.frame_count, sh.n_landmarks), f
    po = np.array([x[1] for x in o["traj"]]); ph = np.array([x[1] for x in traj[0]])
    assert np.abs(po - ph).max() < 1e-5, float(np.abs(po - ph).max())
    a, q = o["oracle"].tracks(), b.tracks(0)
    assert np.array_equal(a[0], q[0]) and np.array_equal(a[1], q[1]) and len(a[0]) > 100
