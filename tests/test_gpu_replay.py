"""Recording -> PNG / text files -> dataio.replay() -> C ABI against the ORACLE pipeline on the same frames, with the nodelet's
frequency control on both sides (GPU)."""
import importlib

import numpy as np
import pytest

import vio_ct

pytestmark = pytest.mark.gpu

INDOOR_YAML = """%YAML:1.0
imu: 1
static_init: 1
depth_min_dist: 0.3
depth_max_dist: 10
frontend_freq: 30
num_grid_rows: 7
num_grid_cols: 8
model_type: PINHOLE
image_width: 848
image_height: 480
distortion_parameters:
   k1: 0.0
   k2: 0.0
   p1: 0.0
   p2: 0.0
projection_parameters:
   fx: 430.0
   fy: 430.0
   cx: 424.0
   cy: 240.0
estimate_extrinsic: 0
max_cnt: 150
min_dist: 25
freq: 10
F_threshold: 1.0
max_num_iterations: 8
keyframe_parallax: 10.0
acc_n: 0.1
gyr_n: 0.01
acc_w: 0.001
gyr_w: 0.0001
g_norm: 9.805
estimate_td: 1
td: 0.0
rolling_shutter: 0
"""


def test_replay_of_a_30hz_recording_matches_the_oracle(P, tmp_path):
    """The parameter set of config/realsense/vio_indoor.yaml (848x480, 7x8 grid, 150 features, min_dist 25, estimate_td, freq 10 /
    frontend_freq 30) replayed from a written 30 Hz recording: dataio.replay applies the frame gate (estimator_nodelet.cpp:264-286)
    and feeds the C ABI; the oracle pipeline runs the same frames under the oracle's restatement of the gate."""
    io = importlib.import_module("vins-rgbd-fast_amd.dataio")
    cfg, extra = io.config_from_yaml(INDOOR_YAML, P)
    assert (extra["freq"], extra["frontend_freq"]) == (10, 30) and cfg.estimate_td == 1 and cfg.max_cnt == 150
    sc = vio_ct.synth_like(cfg, cam_rate=30.0)
    syn = P.Synth(sc)
    seq, n = 6, 96
    stamps = vio_ct.frame_times(sc, n)
    frames = [syn.render_host(seq, float(t)) for t in stamps]
    ti, ai, gi = syn.imu(seq, int(n / sc.cam_rate * sc.imu_rate) + 64)
    io.write_recording(str(tmp_path / "rec"), stamps, [f[0] for f in frames], [f[1] for f in frames], ti, ai, gi)
    rec = io.RgbdImuDirectory(str(tmp_path / "rec"))
    assert len(rec) == n
    b = P.VioBatch(cfg, 1)
    csv = str(tmp_path / "vins_result.csv")
    rows = io.replay(b, rec, csv, freq=extra["freq"], frontend_freq=extra["frontend_freq"])
    assert len(rows) >= 8 and b.status(0).solver_flag == 1
    # the oracle on the same frames (PNG round trip is lossless) with its own gate
    modes = vio_ct.gate_modes(vio_ct.OracleGate(extra["freq"], extra["frontend_freq"]), stamps)
    assert modes.count(1) >= n // 2       # two of three frames are tracked without being published
    o = vio_ct.OraclePipeline(cfg)
    k, ref = 0, []
    for f, tf in enumerate(stamps):
        k2 = k
        while k2 < len(ti) and ti[k2] <= tf + 1e-9:
            k2 += 1
        k2 = min(len(ti), k2 + 1)
        o.push_imu(ti[k:k2], ai[k:k2], gi[k:k2]); k = k2
        r = o.feed(frames[f][0], frames[f][1], tf, modes[f])
        if r == 1 and o.status()["solver_flag"] == 1:
            w = o.window()[cfg.window_size]
            ref.append(np.r_[tf, w[:3], w[3:7], w[7:10]])
    ref = np.array(ref)
    assert rows.shape == ref.shape, (rows.shape, ref.shape)
    assert np.abs(rows[:, 0] - ref[:, 0]).max() < 1e-9
    assert np.abs(rows[:, 1:4] - ref[:, 1:4]).max() < 1e-5, float(np.abs(rows[:, 1:4] - ref[:, 1:4]).max())
    back = io.read_odometry_csv(csv)
    assert back.shape == rows.shape and np.abs(back[:, 1:] - rows[:, 1:]).max() <= 5.1e-6  # 5 decimals in the file
    gt = np.array([syn.pose(seq, float(t))[0] for t in rows[:, 0]])
    assert io.ate_rmse(rows[:, 1:4], gt) < 0.03


def test_stream_discontinuity_restarts_the_estimator_and_keeps_the_tracker(P, tmp_path):
    """estimator_nodelet.cpp:243-262: a gap of more than one second (or a stamp going backwards) empties feature_buf and restarts the
    ESTIMATOR (clearState + setParameter); trackerData keeps its points, ids, track counts and previous image, init_pub / init_feature
    keep their values, and the frame after the restart only sets the time base again (first_image_flag).  Replay of a recording with a
    1.5 s hole against the oracle driven through the same branch (Pipeline::restart), and the tracker's ids must continue, not restart."""
    io = importlib.import_module("vins-rgbd-fast_amd.dataio")
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    syn = P.Synth(sc)
    seq, n = 8, 70
    stamps = vio_ct.frame_times(sc, n)
    keep = [f for f in range(n) if not (22 <= f < 37)]
    frames = {f: syn.render_host(seq, float(stamps[f])) for f in keep}
    ti, ai, gi = syn.imu(seq, int(n / sc.cam_rate * sc.imu_rate) + 64)
    io.write_recording(str(tmp_path / "whole"), [stamps[f] for f in keep], [frames[f][0] for f in keep], [frames[f][1] for f in keep], ti, ai, gi)
    rec = io.RgbdImuDirectory(str(tmp_path / "whole"))
    b = P.VioBatch(cfg, 1)
    seen = {}
    def grab(f, st):
        seen[f] = (st.solver_flag, st.frame_count, st.n_tracks, int(b.tracks(0)[0].max(initial=-1)))
    whole = io.replay(b, rec, freq=10, frontend_freq=30, on_frame=grab)
    # the oracle through the same gate decisions
    o = vio_ct.OraclePipeline(cfg)
    gate = vio_ct.OracleGate(10, 30)
    k, ref, init_pub, init_feature, n_reset = 0, [], False, False, 0
    for i, f in enumerate(keep):
        tf = float(stamps[f])
        k2 = k
        while k2 < len(ti) and ti[k2] <= tf + 1e-9:
            k2 += 1
        k2 = min(len(ti), k2 + 1)
        o.push_imu(ti[k:k2], ai[k:k2], gi[k:k2]); k = k2
        d = gate.step(tf)
        if d == vio_ct.OracleGate.RESET:
            o.restart()
            n_reset += 1
            assert seen[i][0] == 0 and seen[i][1] == 0 and seen[i][2] > 50      # estimator INITIAL again, the tracker still holds its points
            continue
        mode = 2 if d == vio_ct.OracleGate.FIRST else d
        r = o.feed(frames[f][0], frames[f][1], tf, mode)
        if mode == 2 and d != vio_ct.OracleGate.FIRST:
            if not init_pub:
                init_pub = True
            elif not init_feature:
                init_feature = True
            elif r == 0:
                gate.empty_map(tf)
        so = o.status()
        assert (int(so["solver_flag"]), int(so["frame_count"])) == seen[i][:2], (i, f)
        ot = o.tracks()[0]
        assert len(ot) == seen[i][2] and int(ot.max(initial=-1)) == seen[i][3], (i, f)       # same tracker population and ids on both sides
        if r == 1 and so["solver_flag"] == 1:
            w = o.window()[cfg.window_size]
            ref.append(np.r_[tf, w[:3], w[3:7], w[7:10]])
    ref = np.array(ref)
    assert n_reset == 1
    assert whole.shape == ref.shape and len(whole) >= 20, (whole.shape, ref.shape)
    assert np.abs(whole[:, 0] - ref[:, 0]).max() < 1e-9
    assert np.abs(whole[:, 1:4] - ref[:, 1:4]).max() < 1e-5, float(np.abs(whole[:, 1:4] - ref[:, 1:4]).max())
    # both halves produced rows (the estimator re-initialised after the hole), and feature ids kept counting across it
    assert (whole[:, 0] < stamps[22]).sum() >= 5 and (whole[:, 0] > stamps[37]).sum() >= 5
    i_reset = keep.index(37)
    assert seen[len(keep) - 1][3] > seen[i_reset - 1][3] > 100
