"""EQUALIZE (vio_config.equalize): cv::createCLAHE(3.0, Size(8, 8)) before tracking, FeatureTracker::readImage
(feature_tracker.cpp:269-275).  Kernel level: bit-exact against the oracle's restatement; pipeline level: the same frame-by-frame checks
the unequalised pipeline passes."""
import numpy as np
import pytest

import vio_ct

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    return vio_ct.pkg()


@pytest.fixture(scope="module")
def orc():
    return vio_ct.oracle()


def _images(P):
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    syn = P.Synth(sc)
    g = syn.render_host(2, 2.5)[0]
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:480, 0:640]
    dark = (g.astype(np.int32) // 6 + 3).astype(np.uint8)                       # under-exposed: what the option is for
    ramp = ((xx + 2 * yy) % 256).astype(np.uint8)
    return [g, dark, ramp, rng.integers(0, 256, (480, 640), dtype=np.uint8), np.full((480, 640), 77, np.uint8),
            np.ascontiguousarray(g[:75, :100]), np.ascontiguousarray(g[:200, :333]), np.ascontiguousarray(dark[:96, :848 // 2])]


def test_clahe_kernels_bit_exact(P, orc):
    changed = 0
    for img in _images(P):
        h, w = img.shape
        ref, out = np.zeros_like(img), np.zeros_like(img)
        orc.ovio_clahe(img.ctypes.data, w, h, ref.ctypes.data)
        assert P.lib().vio_stage_clahe(img.ctypes.data, w, h, out.ctypes.data) == 0
        assert np.array_equal(ref, out), (w, h, int(np.abs(ref.astype(int) - out).max()))
        changed += int((ref != img).any())
    assert changed >= 6


def test_pipeline_with_equalize_follows_the_oracle(P):
    n, seqs = 24, [41, 42]
    cfg = P.canonical_config(equalize=1)
    sc = vio_ct.synth_like(cfg)
    oruns = [vio_ct.run_oracle_sequence(cfg, sc, s, n) for s in seqs]
    frames = [o["frames"] for o in oruns]
    b, traj, stat = vio_ct.run_hip_batch(P, cfg, sc, seqs, n, frames)
    nl = 0
    for i in range(len(seqs)):
        for f in range(n):
            so, sh = oruns[i]["status"][f], stat[i][f]
            assert (int(so["solver_flag"]), int(so["frame_count"]), int(so["n_landmarks"]), int(so["marginalization_flag"]), int(so["last_track_num"])) == \
                   (sh.solver_flag, sh.frame_count, sh.n_landmarks, sh.marginalization_flag, sh.last_track_num), (i, f)
            if sh.solver_flag == 1 and sh.processed:
                nl += 1
                assert (int(so["iterations"]), int(so["successful_steps"])) == (sh.iterations, sh.successful_steps), (i, f)
        wo, wh = oruns[i]["oracle"].window(), b.window(i)
        assert np.abs(wo[:, :3] - wh[:, :3]).max() < 1e-6
    assert nl >= 16
    # the option is live: the unequalised run of the same frames tracks a different feature set
    b0, _, stat0 = vio_ct.run_hip_batch(P, P.canonical_config(), sc, seqs, n, frames)
    assert any(stat0[0][f].n_tracks != stat[0][f].n_tracks or stat0[0][f].n_landmarks != stat[0][f].n_landmarks for f in range(n))
    b.close(); b0.close()
    with pytest.raises(Exception):
        P.Batch(P.canonical_config(equalize=2), 1)


def test_pipeline_with_a_fisheye_mask_follows_the_oracle(P):
    """FISHEYE (estimator.cpp:29-36, feature_tracker.cpp:175-176): setMask starts from fisheye_mask.  A disk of 255 with a grey rim (a
    decoded JPEG has one: `== 255` for tracked / added points, `!= 0` for the FAST filter) and 0 outside."""
    n, seqs = 24, [43, 44]
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    yy, xx = np.mgrid[0:cfg.height, 0:cfg.width]
    rr = np.hypot(xx - 320.0, yy - 240.0)
    mask = np.where(rr < 210, 255, np.where(rr < 222, 128, 0)).astype(np.uint8)
    oruns = [vio_ct.run_oracle_sequence(cfg, sc, s, n, fisheye_mask=mask) for s in seqs]
    frames = [o["frames"] for o in oruns]
    b, traj, stat = vio_ct.run_hip_batch(P, cfg, sc, seqs, n, frames, fisheye_mask=mask)
    nl = 0
    for i in range(len(seqs)):
        for f in range(n):
            so, sh = oruns[i]["status"][f], stat[i][f]
            assert (int(so["solver_flag"]), int(so["frame_count"]), int(so["n_landmarks"]), int(so["marginalization_flag"]), int(so["last_track_num"])) == \
                   (sh.solver_flag, sh.frame_count, sh.n_landmarks, sh.marginalization_flag, sh.last_track_num), (i, f)
            nl += int(sh.solver_flag == 1 and sh.processed)
        io, co, po = oruns[i]["oracle"].tracks()[:3]
        ih, ch, ph = b.tracks(i)[:3]
        assert np.array_equal(io, ih) and np.array_equal(co, ch) and np.array_equal(po, ph)
        assert len(ih) > 30 and (mask[np.rint(ph[:, 1]).astype(int), np.rint(ph[:, 0]).astype(int)] == 255).all()
        wo, wh = oruns[i]["oracle"].window(), b.window(i)
        assert np.abs(wo[:, :3] - wh[:, :3]).max() < 1e-6
    assert nl >= 16
    # turning the option off again restores the unmasked tracker
    b.set_fisheye_mask(None)
    b.close()
