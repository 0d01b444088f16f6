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


def test_stream_discontinuity_restarts_the_sequence(P, tmp_path):
    """estimator_nodelet.cpp:243-262: a gap of more than one second (or a stamp going backwards) restarts the tracker and the
    estimator.  Replay of a recording with a 1.5 s hole must equal two independent replays of its halves."""
    io = importlib.import_module("vins-rgbd-fast_amd.dataio")
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    syn = P.Synth(sc)
    seq, n = 8, 56
    stamps = vio_ct.frame_times(sc, n)
    keep = [f for f in range(n) if not (22 <= f < 37)]
    frames = {f: syn.render_host(seq, float(stamps[f])) for f in keep}
    ti, ai, gi = syn.imu(seq, int(n / sc.cam_rate * sc.imu_rate) + 64)

    def rec_of(fs, name, k0=0):
        io.write_recording(str(tmp_path / name), [stamps[f] for f in fs], [frames[f][0] for f in fs], [frames[f][1] for f in fs], ti[k0:], ai[k0:], gi[k0:])
        return io.RgbdImuDirectory(str(tmp_path / name))
    whole = io.replay(P.VioBatch(cfg, 1), rec_of(keep, "whole"), freq=10, frontend_freq=30)
    first = io.replay(P.VioBatch(cfg, 1), rec_of([f for f in keep if f < 22], "a"), freq=10, frontend_freq=30)
    # the frame right after the hole triggers the restart and is dropped; the one after it is the new first image
    # (clearState() empties imu_buf: the samples pushed up to the restart frame, one beyond its stamp, are gone)
    k0 = int(np.searchsorted(ti, stamps[37] + 1e-9, side="right")) + 1
    second = io.replay(P.VioBatch(cfg, 1), rec_of([f for f in keep if f >= 38], "b", k0), freq=10, frontend_freq=30)
    assert len(first) >= 5 and len(second) >= 3
    assert len(whole) == len(first) + len(second)
    assert np.array_equal(whole[:len(first)], first)
    assert np.array_equal(whole[len(first):], second)
