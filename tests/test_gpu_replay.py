"""Recording -> PNG / text files -> dataio.replay() -> C ABI equals feeding the same frames directly (GPU)."""
import importlib

import numpy as np
import pytest

import vio_ct

pytestmark = pytest.mark.gpu


def test_replay_of_a_written_recording_matches_direct_feed(P, tmp_path):
    io = importlib.import_module("vins-rgbd-fast_amd.dataio")
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    syn = P.Synth(sc)
    seq, n = 6, 24
    stamps = vio_ct.frame_times(sc, n)
    frames = [syn.render_host(seq, float(t)) for t in stamps]
    ti, ai, gi = syn.imu(seq, int(n / sc.cam_rate * sc.imu_rate) + 64)
    io.write_recording(str(tmp_path / "rec"), stamps, [f[0] for f in frames], [f[1] for f in frames], ti, ai, gi)
    rec = io.RgbdImuDirectory(str(tmp_path / "rec"))
    assert len(rec) == n
    b = P.VioBatch(cfg, 1)
    csv = str(tmp_path / "vins_result.csv")
    rows = io.replay(b, rec, csv)
    assert len(rows) >= 8 and b.status(0).solver_flag == 1
    # direct feed of the same frames
    d = P.VioBatch(cfg, 1)
    k, ref = 0, []
    for f, tf in enumerate(stamps):
        k2 = vio_ct.imu_until(ti, k, tf, sc.imu_rate)
        if k2 > k:
            d.push_imu(0, ti[k:k2], ai[k:k2], gi[k:k2])
        k = k2
        d.feed(frames[f][0][None], frames[f][1][None], [tf])
        st = d.status(0)
        if st.solver_flag == 1 and st.processed:
            ref.append(d.odometry()[0].copy())
    ref = np.array(ref)
    assert rows.shape == ref.shape and np.abs(rows - ref).max() < 1e-12
    back = io.read_odometry_csv(csv)
    assert back.shape == rows.shape and np.abs(back[:, 1:] - rows[:, 1:]).max() <= 5.1e-6  # 5 decimals in the file
    gt = np.array([syn.pose(seq, float(t))[0] for t in rows[:, 0]])
    assert io.ate_rmse(rows[:, 1:4], gt) < 0.03
