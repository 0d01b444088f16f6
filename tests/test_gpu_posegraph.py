"""pose_graph slice on the GPU (include/vio_posegraph.h): HIP kernels against the oracle bit for bit, the host half
against the oracle and the truth, and the whole chain -- keyframe descriptors -> descriptor search -> PnP RANSAC -> match list ->
vio_set_relo_frame -> relocalisation factors inside optimization() -- on a live estimator."""
import importlib

import numpy as np
import pytest

import vio_ct
import test_oracle_posegraph_cpu as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    return vio_ct.pkg()


@pytest.fixture(scope="module")
def PG():
    return importlib.import_module("vins-rgbd-fast_amd.posegraph")


def _q2R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def test_descriptors_keypoints_and_search_are_bit_exact(P, PG):
    """KeyFrame::computeWindowBRIEFPoint / computeBRIEFPoint / searchByBRIEFDes (keyframe.cpp:80-169): blur, FAST(20, NMS) keypoints in
    row-major order, 256-bit descriptors, normalised keypoints and the Hamming search -- identical to the oracle on rendered frames and on a
    noise image (tens of thousands of keypoints, more than the caller made room for)."""
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    syn = P.Synth(sc)
    pat = O.pattern()
    assert np.array_equal(PG.load_brief_pattern(O.os.path.join(O.GOLD, "brief_pattern.npz")), pat)
    imgs = [syn.render_host(5, 2.0)[0], syn.render_host(5, 2.3)[0], np.random.default_rng(1).integers(0, 256, (cfg.height, cfg.width), dtype=np.uint8)]
    rng = np.random.default_rng(2)
    descs = []
    for k, img in enumerate(imgs):
        uv = np.c_[rng.uniform(-2, cfg.width + 2, 150), rng.uniform(-2, cfg.height + 2, 150)].astype(np.float32)   # incl. points at / beyond the border
        # the noise image has ~30 000 keypoints: cv::FAST has no cap, so PG.describe, started with room for 1000, must come back with ALL of them
        # (round 5, ADVICE r4: it used to truncate in row-major order, i.e. drop the bottom of the image); the oracle gets room for all at once
        cap = 8192 if k < 2 else 1000
        wd_h, kxy_h, kd_h, kn_h = PG.describe(cfg, img, uv, pat, cap=cap)
        wd_o, kxy_o, kd_o, kn_o = O.o_describe(cfg, img, uv, pat, cap=cap if k < 2 else 40000)
        assert len(kxy_h) == len(kxy_o) and (len(kxy_h) > 100 if k < 2 else len(kxy_h) > 20000)
        assert np.array_equal(wd_h, wd_o) and np.array_equal(kxy_h, kxy_o) and np.array_equal(kd_h, kd_o)
        assert np.array_equal(kn_h.view(np.uint32), kn_o.view(np.uint32))        # liftProjective, bitwise
        descs.append((wd_h, kd_h))
    # the blurred image itself
    L = P.lib()
    out = np.zeros_like(imgs[0])
    assert L.vio_pg_stage_blur(imgs[0].ctypes.data, cfg.width, cfg.height, out.ctypes.data) == 0
    ref = np.zeros_like(imgs[0])
    O.olib().ovio_pg_blur(imgs[0].ctypes.data, cfg.width, cfg.height, ref.ctypes.data)
    assert np.array_equal(out, ref)
    # Hamming search: keypoints of frame 1 against keypoints of frame 0 (real near-duplicates), and against nothing
    a, b = descs[1][1][:300], descs[0][1]
    bi_h, bd_h = PG.match(a, b)
    bi_o, bd_o = O.o_match(a, b)
    assert np.array_equal(bi_h, bi_o) and np.array_equal(bd_h, bd_o) and (bi_h >= 0).sum() > 50
    bi_e, bd_e = PG.match(a[:5], np.zeros((0, 4), np.uint64))
    assert np.all(bi_e == -1) and np.all(bd_e == 128)


def test_find_connection_and_optimize4dof_against_oracle_and_truth(P, PG):
    """KeyFrame::PnPRANSAC / findConnection (keyframe.cpp:195-528) and PoseGraph::optimize4DoF (pose_graph.cpp:410-581): the product's host code
    against the oracle's independent restatement and against the generating truth."""
    s = O._loop_scene(np.random.default_rng(11))
    ok_h, info_h, mp_h, pT_h, pR_h = PG.find_connection(s["p3"], s["ids"], s["match"], s["old_norm"], s["T"], s["R"], s["qic"], s["tic"])
    ok_o, info_o, mp_o, pT_o, pR_o = O.o_find_connection(s["p3"], s["pn"], s["ids"], s["match"], s["old_norm"], s["T"], s["R"], s["qic"], s["tic"])
    assert ok_h and ok_o and np.array_equal(mp_h[:, 2], mp_o[:, 2]) and np.abs(mp_h - mp_o).max() == 0
    assert np.abs(info_h - info_o).max() < 1e-6 and np.abs(pT_h - pT_o).max() < 1e-6 and np.abs(pR_h - pR_o).max() < 1e-7
    assert np.abs(info_h[:3] - s["R_old"].T @ (s["T"] - s["T_old"])).max() < 0.02 and abs(info_h[7] - 8.0) < 0.3
    few = s["match"].copy(); few[20:] = -1
    assert not PG.find_connection(s["p3"], s["ids"], few, s["old_norm"], s["T"], s["R"], s["qic"], s["tic"])[0]
    s2 = O._loop_scene(np.random.default_rng(12), yaw_deg=35.0)
    assert not PG.find_connection(s2["p3"], s2["ids"], s2["match"], s2["old_norm"], s2["T"], s2["R"], s2["qic"], s2["tic"])[0]
    # 4-DoF pose graph
    t_true, R_true, t_vio, R_vio, seq, loop_to, info = O._drift_graph()
    to_h, Ro_h, (yd_h, td_h) = PG.optimize4DoF(t_vio, R_vio, seq, loop_to, info)
    to_o, Ro_o, dr_o = O.o_optimize4dof(t_vio, R_vio, seq, loop_to, info)
    assert np.abs(to_h - to_o).max() < 1e-6 and np.abs(Ro_h - Ro_o).max() < 1e-7 and abs(yd_h - dr_o[0]) < 1e-6 and np.abs(td_h - dr_o[1:]).max() < 1e-6
    e0, e1 = np.linalg.norm(t_vio - t_true, axis=1), np.linalg.norm(to_h - t_true, axis=1)
    assert e1[-1] < 0.4 * e0[-1]
    # a second sequence-0 block stays fixed, mixed sequences have no sequential edges across the boundary
    seq2 = seq.copy(); seq2[:10] = 0
    to2, Ro2, _ = PG.optimize4DoF(t_vio, R_vio, seq2, loop_to, info)
    assert np.abs(to2[:10] - t_vio[:10]).max() == 0
    # 6-DoF pose graph (the `imu: 0` variant, pose_graph.cpp:583-740)
    t_true, R_true, t_vio, R_vio, seq, loop_to, info = O._drift_graph6()
    to_h, Ro_h, (rd_h, td_h) = PG.optimize6DoF(t_vio, R_vio, seq, loop_to, info)
    to_o, Ro_o, dr_o = O.o_optimize6dof(t_vio, R_vio, seq, loop_to, info)
    assert np.abs(to_h - to_o).max() < 1e-6 and np.abs(Ro_h - Ro_o).max() < 1e-7
    assert np.abs(rd_h.ravel() - dr_o[:9]).max() < 1e-7 and np.abs(td_h - dr_o[9:]).max() < 1e-6
    assert np.linalg.norm(to_h[-1] - t_true[-1]) < 0.85 * np.linalg.norm(t_vio[-1] - t_true[-1])


def test_loop_verification_feeds_the_relocalisation_of_a_live_estimator(P, PG):
    """The whole chain on the HIP path: a window frame of a running sequence is the current keyframe (window points = its tracked features with
    their estimated world positions), the image of the frame `back` frames earlier the loop candidate (pose = ground truth, standing in for an
    old map).  KeyFrame descriptors -> descriptor search -> PnP RANSAC -> match list -> vio_set_relo_frame; the next optimisation carries the
    relocalisation factors and reports the relative pose of the two frames -- checked against the truth of the scene."""
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    syn = P.Synth(sc)
    pat = O.pattern()
    seq, n_frames, f_set, i_local, back = 3, 40, 36, 6, 2
    times = vio_ct.frame_times(sc, n_frames)
    ti, ai, gi = syn.imu(seq, int(n_frames / sc.cam_rate * sc.imu_rate) + 64)
    b = P.VioBatch(cfg, 1)
    ric, tic = np.array(list(cfg.ric)).reshape(3, 3), np.array(list(cfg.tic))
    maps, imgs, k, out, ctx = {}, {}, 0, None, {}
    for f in range(n_frames):
        tf = float(times[f])
        k2 = vio_ct.imu_until(ti, k, tf, sc.imu_rate)
        g, d = syn.render_host(seq, tf)
        b.push_imu(0, ti[k:k2], ai[k:k2], gi[k:k2]); k = k2
        b.feed(g[None], d[None], [tf])
        maps[round(tf, 6)] = b.packaged(0)
        imgs[round(tf, 6)] = g
        if f == f_set:
            w = b.window(0)
            stamp_i, stamp_k = float(w[i_local, 16]), float(w[i_local - back, 16])
            ids_i, obs_i = maps[round(stamp_i, 6)]
            # world points of the tracked features (pubKeyframe, visualization.cpp:400-452): first observation scaled by the estimated depth
            lm = {int(r[0]): r for r in b.landmarks_ex(0)}
            keep, p3 = [], []
            for j, fid in enumerate(ids_i):
                r = lm.get(int(fid))
                if r is None or not (r[3] > 0) or int(r[5]) != 1:
                    continue
                s0 = int(r[1])
                Rs, Ps = _q2R(w[s0, 3:7]), w[s0, :3]
                p3.append(Rs @ (ric @ (r[3] * r[7:10]) + tic) + Ps)
                keep.append(j)
            keep = np.array(keep)
            assert len(keep) > 60
            p_gt_k, R_gt_k, _ = syn.pose(seq, stamp_k)
            cur = PG.KeyFrame(cfg, pat, stamp_i, 7, w[i_local, :3], _q2R(w[i_local, 3:7]), imgs[round(stamp_i, 6)], np.array(p3), obs_i[keep, 3:5], obs_i[keep, 0:2],
                              ids_i[keep].astype(np.float64))
            old = PG.KeyFrame(cfg, pat, stamp_k, 3, p_gt_k, R_gt_k, imgs[round(stamp_k, 6)], np.zeros((0, 3)), np.zeros((0, 2)), np.zeros((0, 2)), np.zeros(0))
            assert len(old.keypoints) > 200
            assert cur.findConnection(old, ric, tic)
            assert len(cur.match_points) > PG.MIN_LOOP_NUM and np.all(np.diff(cur.match_points[:, 2]) > 0)
            # the verified relative pose against the truth: frame i seen from the old keyframe
            p_gt_i, R_gt_i, _ = syn.pose(seq, stamp_i)
            rel_truth = R_gt_k.T @ (p_gt_i - p_gt_k)
            ctx = dict(rel_truth=rel_truth, R_rel=R_gt_k.T @ R_gt_i, n_match=len(cur.match_points))
            # (PnP_T_old is the old keyframe's pose in the ESTIMATOR's world; loop_info relates the two frames)
            assert np.abs(cur.loop_info[:3] - rel_truth).max() < 0.03, (cur.loop_info[:3], rel_truth)
            b.set_relo_frame(0, stamp_i, cur.index, cur.match_points, old.T_w_i, old.R_w_i)
        if f == f_set + 1:
            out = b.relo(0)
    assert out["pending"] == 0 and out["local_index"] == i_local and out["n_factors"] >= PG.MIN_LOOP_NUM
    assert b.status(0).overflow_flags == 0
    assert np.abs(out["relative_t"] - ctx["rel_truth"]).max() < 0.02, (out["relative_t"], ctx["rel_truth"])
    assert np.abs(_q2R(out["relative_q"]) - ctx["R_rel"]).max() < 0.02
