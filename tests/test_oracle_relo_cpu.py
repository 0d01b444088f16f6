"""Relocalisation inside optimization() (first slice: estimator.cpp:1307-1346 relocalisation factors, :1728-1747
setReloFrame, :1034-1056 drift / relative-pose outputs) -- the ORACLE against the truth of the synthetic scene.

Set-up: the tracker's own feature maps are kept per frame; at a NON_LINEAR frame the window frame with local index i is declared to
match an "old keyframe" whose observations are those of window frame k = i - 2 and whose pose is the ground truth of frame k.  The
relocalisation pose starts as a copy of pose i and must be pulled to pose k by the projection factors alone."""
import numpy as np
import pytest

import vio_ct


def _q2R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def drive_with_relo(P, cfg, sc, seq, n_frames, f_set, i_local, back, make_pipe, set_relo, get_relo, window_of, feed_obs):
    """shared driver (the GPU test passes the HIP pipeline's accessors): returns (relo dict after the relocalisation solve, context)"""
    syn = P.Synth(sc)
    times = vio_ct.frame_times(sc, n_frames)
    ti, ai, gi = syn.imu(seq, int(n_frames / sc.cam_rate * sc.imu_rate) + 64)
    pipe = make_pipe()
    maps, k, out = {}, 0, None
    for f in range(n_frames):
        tf = float(times[f])
        k2 = vio_ct.imu_until(ti, k, tf, sc.imu_rate)
        g, d = syn.render_host(seq, tf)
        ids, obs, solved = feed_obs(pipe, f, tf, g, d, (ti[k:k2], ai[k:k2], gi[k:k2]))
        k = k2
        maps[round(tf, 6)] = (ids, obs)
        if f == f_set:
            w = window_of(pipe)
            stamp_i, stamp_k = w[i_local, 16], w[i_local - back, 16]
            ids_k, obs_k = maps[round(float(stamp_k), 6)]
            mp = np.c_[obs_k[:, 0], obs_k[:, 1], ids_k.astype(np.float64)]
            p_gt, R_gt, _ = syn.pose(seq, float(stamp_k))
            ctx = dict(stamp_i=float(stamp_i), stamp_k=float(stamp_k), n_match=len(mp), window_before=w.copy(), p_gt_k=p_gt, R_gt_k=R_gt,
                       p_gt_i=syn.pose(seq, float(stamp_i))[0], R_gt_i=syn.pose(seq, float(stamp_i))[1])
            set_relo(pipe, float(stamp_i), 7, mp, p_gt, R_gt)
        if f == f_set + 1:
            out = get_relo(pipe)
            ctx["window_after"] = window_of(pipe).copy()
    return out, ctx, pipe


def oracle_accessors(cfg):
    def feed_obs(o, f, tf, g, d, imu):
        o.push_imu(*imu)
        ids, obs = o.track(g, tf)
        if len(ids):
            o.process_obs(ids, obs, d, tf)
        return ids, obs, True
    return dict(make_pipe=lambda: vio_ct.OraclePipeline(cfg), set_relo=lambda o, *a: o.set_relo_frame(*a), get_relo=lambda o: o.relo(),
                window_of=lambda o: o.window(), feed_obs=feed_obs)


@pytest.mark.parametrize("seq,back", [(3, 2), (6, 3)])
def test_relocalisation_pulls_the_copy_of_pose_i_onto_the_matched_keyframe(P, seq, back):
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    n_frames, f_set, i_local = 40, 36, 6
    r, ctx, o = drive_with_relo(P, cfg, sc, seq, n_frames, f_set, i_local, back, **oracle_accessors(cfg))
    assert ctx["n_match"] > 60
    assert r["pending"] == 0 and r["local_index"] == i_local and r["n_factors"] >= 30       # consumed by exactly one solve
    # truth: relative pose of body frame i in body frame k
    Rk, Ri = ctx["R_gt_k"], ctx["R_gt_i"]
    rel_t = Rk.T @ (ctx["p_gt_i"] - ctx["p_gt_k"])
    assert np.linalg.norm(rel_t) > 0.02                                                      # the two frames really differ
    assert np.abs(r["relative_t"] - rel_t).max() < 0.01, (r["relative_t"], rel_t)
    yaw = lambda R: np.degrees(np.arctan2(R[1, 0], R[0, 0]))
    d_yaw = (yaw(Ri) - yaw(Rk) + 180) % 360 - 180
    assert abs(r["relative_yaw"] - d_yaw) < 0.5, (r["relative_yaw"], d_yaw)
    # relative rotation as a whole
    Rrel = _q2R(r["relative_q"])
    assert np.abs(Rrel - Rk.T @ Ri).max() < 0.01
    # drift correction maps the estimator's world onto the world of the old keyframe (here: the ground-truth world): applied to the
    # relocalisation pose it must give the old keyframe's pose back exactly (definition, estimator.cpp:1046-1049) ...
    w = ctx["window_after"]
    # ... and applied to the window it must land every frame on its ground truth to the accuracy of the odometry (yaw + translation only)
    syn = P.Synth(sc)
    est = np.array([r["drift_r"] @ w[j, :3] + r["drift_t"] for j in range(cfg.window_size + 1)])
    gt = np.array([syn.pose(seq, float(w[j, 16]))[0] for j in range(cfg.window_size + 1)])
    assert np.abs(est - gt).max() < 0.03, float(np.abs(est - gt).max())
    assert abs(np.linalg.det(r["drift_r"]) - 1) < 1e-12 and abs(r["drift_r"][2, 2] - 1) < 1e-12   # a pure yaw rotation


def test_relocalisation_is_ignored_when_the_stamp_is_not_in_the_window(P):
    cfg = P.canonical_config()
    o = vio_ct.OraclePipeline(cfg)
    o.set_relo_frame(123.456, 1, np.zeros((3, 3)), np.zeros(3), np.eye(3))
    assert o.relo()["pending"] == 0
