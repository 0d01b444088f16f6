"""GPU parity where the depth image has holes (VERDICT r4 "next" item 2): landmarks triangulated WITHOUT a depth measurement
(feature_manager.cpp:465-520: DLT over the window's poses, estimate_flag 2), the upper bound on their inverse depth
(estimator.cpp:1282-1297, `SetParameterUpperBound(para_Feature, 0, 2 / DEPTH_MAX_DIST)`; here: projection of the candidate onto the box,
DESIGN.md deviation 5), the erasure of features whose depth pixel reads closer than DEPTH_MIN_DIST (feature_manager.cpp:76-80) and
movingConsistencyCheck marking landmarks dynamic (estimator.cpp:1944-2009).  The canonical workload reaches none of these branches (its
depth image is valid everywhere), so every scenario edits the rendered frames on the host -- the same edited frames go to the oracle and
through the C ABI to the HIP path -- and first proves that the ORACLE run really went through the branch.

The bar: the landmark table (ids, start frame, observation count, estimate / solve flags, is_dynamic) identical after EVERY frame, the
estimated depths within 1e-6 relative, window positions within 1e-5 m (the scenes are harder than the canonical one: the tolerance is
stated per test)."""
import numpy as np
import pytest

import vio_ct

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    return vio_ct.pkg()


def _frames(P, sc, seq, n, edit):
    syn = P.Synth(sc)
    out = []
    for f, t in enumerate(vio_ct.frame_times(sc, n)):
        g, d = syn.render_host(seq, float(t))
        out.append(edit(f, g.copy(), d.copy()))
    return out


def _moving_patch(f, g, d, size=110, depth_mm=1500):
    """a textured square that slides 5 px per frame through the image at a constant measured depth: an object moving in the world"""
    rng = np.random.RandomState(7)
    tex = rng.randint(0, 256, (size // 6 + 2, size // 6 + 2)).astype(np.float32)
    tex = np.kron(tex, np.ones((6, 6), np.float32))[:size, :size]
    pad = np.pad(tex, 1, mode="edge")
    tex = sum(pad[i:i + size, j:j + size] for i in range(3) for j in range(3)) / 9.0
    x0, y0 = 60 + 5 * f, 180 + (f % 2)
    if x0 + size < g.shape[1] - 10:
        g[y0:y0 + size, x0:x0 + size] = np.clip(tex + 0.5, 0, 255).astype(np.uint8)
        d[y0:y0 + size, x0:x0 + size] = depth_mm
    return g, d


def _run_both(P, cfg, sc, seq, n_frames, frames):
    lm_o, lm_h = [], []
    n_tracks = []

    def hook_o(f, orc):
        lm_o.append(orc.landmarks_ex())
        n_tracks.append(len(orc.tracks()[0]))
    o = vio_ct.run_oracle_sequence(cfg, sc, seq, n_frames, frames=frames, hook=hook_o)
    o["n_tracks"] = n_tracks
    b, traj, stat = vio_ct.run_hip_batch(P, cfg, sc, [seq], n_frames, [frames], hook=lambda f, bb: lm_h.append(bb.landmarks_ex(0)))
    return o, lm_o, b, traj[0], stat[0], lm_h


def _compare(o, lm_o, traj, stat, lm_h, n_frames, pos_tol, depth_tol=1e-6, exact_frames=30):
    """status and landmark tables equal after every frame (flags, dynamic marks, observation counts; depths to depth_tol), positions to pos_tol;
    returns the worst relative depth difference"""
    worst, n_tri = 0.0, 0
    for f in range(n_frames):
        so, sh = o["status"][f], stat[f]
        assert (int(so["solver_flag"]), int(so["frame_count"]), int(so["n_landmarks"])) == (sh.solver_flag, sh.frame_count, sh.n_landmarks), f
        if sh.solver_flag == 1 and sh.processed:
            assert int(so["marginalization_flag"]) == sh.marginalization_flag, f
            assert (int(so["n_in_problem"]), int(so["n_residuals"]), int(so["n_var_landmarks"])) == (sh.n_in_problem, sh.n_residuals, sh.n_var_landmarks), f
            assert (int(so["iterations"]), int(so["successful_steps"])) == (sh.iterations, sh.successful_steps), f
        a, h = lm_o[f], lm_h[f]
        assert a.shape == h.shape, (f, a.shape, h.shape)
        if len(a) == 0:
            continue
        assert np.array_equal(a[:, [0, 1, 2, 4, 5, 6]], h[:, [0, 1, 2, 4, 5, 6]]), f   # id, start, n_obs, estimate_flag, solve_flag, is_dynamic
        # first observation (x, y, z), its depth, the last depth.  Identical while the trackers are (the first ~25 frames after the static
        # start, test_gpu_pipeline.py); later a tracked position may differ in its last float digits (LK stops at 0.01 px), which moves a
        # normalised coordinate by ~1e-6 and, at a depth edge, the depth pixel it reads
        if f < exact_frames:
            assert np.array_equal(a[:, 7:12], h[:, 7:12]), f
        else:
            assert np.abs(a[:, 7:10] - h[:, 7:10]).max() < 2e-5, f
            assert (np.abs(a[:, 10:12] - h[:, 10:12]) > 1e-9).mean() < 0.02, f
        have = a[:, 3] > 0
        assert np.array_equal(have, h[:, 3] > 0), f
        if have.any():
            worst = max(worst, float((np.abs(a[have, 3] - h[have, 3]) / np.abs(a[have, 3])).max()))
            n_tri += int(have.sum())
    assert n_tri > 1000 and worst < depth_tol, (n_tri, worst)
    po = np.array([x[1] for x in o["traj"]]); ph = np.array([x[1] for x in traj])
    assert len(po) == len(ph) >= 30
    assert np.abs(po - ph).max() < pos_tol, float(np.abs(po - ph).max())
    return worst


def test_depthless_landmarks_of_a_blinded_sensor_match_the_oracle(P):
    """DEPTH_MAX_DIST = 3 m and the depth image blinded beyond it (the oracle KAT's scenario, test_oracle_kat.py): every farther landmark is
    triangulated from parallax only (flag 2) and optimised under the inverse-depth bound; the bound itself stays inactive (a landmark beyond
    3 m has an inverse depth below 1/3, half the bound)."""
    cfg = P.canonical_config(depth_max=3.0)
    sc = vio_ct.synth_like(cfg)
    seq, n = 2, 60

    def blind(f, g, d):
        d[d > 3000] = 0
        return g, d
    frames = _frames(P, sc, seq, n, blind)
    o, lm_o, b, traj, stat, lm_h = _run_both(P, cfg, sc, seq, n, frames)
    clamps, bounded = o["oracle"].bound_stats()
    assert bounded > 1000 and clamps == 0, (clamps, bounded)
    flag2_frames = sum(int((a[:, 4] == 2).sum()) for a in lm_h if len(a))
    assert flag2_frames > 1000, flag2_frames                      # the HIP tables hold the DLT landmarks too (and equal the oracle's below)
    _compare(o, lm_o, traj, stat, lm_h, n, pos_tol=1e-5)


def test_inverse_depth_bound_engages_identically(P):
    """A sensor declared to reach 10 m but blind beyond 2.5 m: parallax-only landmarks at 2.5 .. 5 m violate `depth >= DEPTH_MAX_DIST / 2`
    all the time, so the projection onto the box cuts candidate steps in almost every solve (tens of thousands of times on the oracle), the
    clamped depths are inconsistent with the images and movingConsistencyCheck marks dozens of landmarks dynamic.  The HIP path must
    make the same cuts: landmarks sitting EXACTLY on the bound (depth = 5 m to the bit) on both sides, identical tables."""
    cfg = P.canonical_config(depth_max=10.0)
    sc = vio_ct.synth_like(cfg)
    seq, n = 2, 60

    def blind(f, g, d):
        d[d > 2500] = 0
        return g, d
    frames = _frames(P, sc, seq, n, blind)
    o, lm_o, b, traj, stat, lm_h = _run_both(P, cfg, sc, seq, n, frames)
    clamps, bounded = o["oracle"].bound_stats()
    assert clamps > 1000 and bounded > 1000, (clamps, bounded)
    on_bound_o = sum(int(((a[:, 4] == 2) & (a[:, 3] == 5.0)).sum()) for a in lm_o if len(a))
    on_bound_h = sum(int(((a[:, 4] == 2) & (a[:, 3] == 5.0)).sum()) for a in lm_h if len(a))
    dyn_o = max(int((a[:, 6] != 0).sum()) for a in lm_o if len(a))
    assert on_bound_o > 500 and dyn_o > 20, (on_bound_o, dyn_o)
    assert on_bound_h == on_bound_o, (on_bound_h, on_bound_o)
    # (49 000 clamped candidates and up to 100 dynamic landmarks make this the least well conditioned scene of the suite: the depths agree
    # to 1.4e-5 relative, not to the 1e-6 of the scenes above)
    _compare(o, lm_o, traj, stat, lm_h, n, pos_tol=1e-4, depth_tol=1e-4)


def test_near_depth_erasure_depth_holes_and_a_moving_object(P):
    """One scene with the three remaining depth-image branches: the left quarter of the depth image reads 0.1 m (< DEPTH_MIN_DIST = 0.3 m:
    those features are erased from the map before they reach the landmark table, feature_manager.cpp:76-80), the lower right corner has no
    return (parallax-only landmarks), and a textured square slides through the image at a constant measured depth (an object moving in the
    world: its landmarks fail movingConsistencyCheck, estimator.cpp:1944-2009, and leave the problem)."""
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    seq, n = 5, 60

    def edit(f, g, d):
        d[:, :160] = 100
        d[300:, 400:] = 0
        return _moving_patch(f, g, d)
    frames = _frames(P, sc, seq, n, edit)
    o, lm_o, b, traj, stat, lm_h = _run_both(P, cfg, sc, seq, n, frames)
    # the oracle went through the branches: fewer landmarks than tracked features (erasure), flag-2 landmarks, dynamic landmarks
    f0 = next(f for f in range(n) if len(lm_o[f]))                       # the first frame that fills the landmark table: one landmark per surviving feature
    assert lm_o[f0].shape[0] < o["n_tracks"][f0] - 10, (lm_o[f0].shape, o["n_tracks"][f0])
    assert max(int((a[:, 4] == 2).sum()) for a in lm_o if len(a)) > 20
    assert max(int((a[:, 6] != 0).sum()) for a in lm_o if len(a)) > 10
    _compare(o, lm_o, traj, stat, lm_h, n, pos_tol=1e-4, depth_tol=1e-5)
