"""Oracle of the pose_graph slice (oracle/posegraph.cpp) against definitions and ground truth -- no GPU, no product code except the ctypes
structs.  KeyFrame::computeBRIEFPoint / searchByBRIEFDes / PnPRANSAC / findConnection (pose_graph/src/keyframe/keyframe.cpp:80-528),
PoseGraph::optimize4DoF (pose_graph/src/pose_graph/pose_graph.cpp:410-581)."""
import ctypes as C
import os

import numpy as np
import pytest

import vio_ct

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pattern():
    z = np.load(os.path.join(GOLD, "brief_pattern.npz"))
    return np.ascontiguousarray(np.concatenate([z[k].astype(np.int32) for k in ("x1", "y1", "x2", "y2")]))


def olib():
    L = vio_ct.oracle()
    L.ovio_pg_describe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ovio_pg_blur.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.ovio_pg_match.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.ovio_pg_find_connection.argtypes = [C.c_int] + [C.c_void_p] * 9 + [C.c_int] + [C.c_void_p] * 5
    L.ovio_pg_optimize4dof.argtypes = [C.c_int] + [C.c_void_p] * 8
    L.ovio_pg_optimize6dof.argtypes = [C.c_int] + [C.c_void_p] * 8
    return L


def o_describe(cfg, gray, uv, pat, thr=20, cap=8192):
    L = olib()
    uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    wd = np.zeros((max(len(uv), 1), 4), np.uint64)
    kxy, kd, kn = np.zeros((cap, 2), np.float32), np.zeros((cap, 4), np.uint64), np.zeros((cap, 2), np.float32)
    m = L.ovio_pg_describe(C.byref(cfg), gray.ctypes.data, len(uv), uv.ctypes.data, pat.ctypes.data, thr, wd.ctypes.data, cap, kxy.ctypes.data, kd.ctypes.data,
                           kn.ctypes.data)
    m = min(m, cap)
    return wd[:len(uv)], kxy[:m], kd[:m], kn[:m]


def o_match(a, b):
    L = olib()
    a, b = np.ascontiguousarray(a, np.uint64), np.ascontiguousarray(b, np.uint64)
    bi, bd = np.zeros(max(len(a), 1), np.int32), np.zeros(max(len(a), 1), np.int32)
    L.ovio_pg_match(a.ctypes.data, len(a), b.ctypes.data, len(b), bi.ctypes.data, bd.ctypes.data)
    return bi[:len(a)], bd[:len(a)]


def o_find_connection(p3, pn, ids, mi, old_norm, T, R, qic, tic, min_loop=25):
    L = olib()
    p3 = np.ascontiguousarray(p3, np.float32); pn = np.ascontiguousarray(pn, np.float32); ids = np.ascontiguousarray(ids, np.float64)
    mi = np.ascontiguousarray(mi, np.int32); on = np.ascontiguousarray(old_norm, np.float32)
    T, R, q, t = (np.ascontiguousarray(x, np.float64) for x in (T, R, qic, tic))
    info, mp, nm, pT, pR = np.zeros(8), np.zeros((max(len(p3), 1), 3)), np.zeros(1, np.int32), np.zeros(3), np.zeros((3, 3))
    rc = L.ovio_pg_find_connection(len(p3), p3.ctypes.data, pn.ctypes.data, ids.ctypes.data, mi.ctypes.data, on.ctypes.data, T.ctypes.data, R.ctypes.data,
                                   q.ctypes.data, t.ctypes.data, min_loop, info.ctypes.data, mp.ctypes.data, nm.ctypes.data, pT.ctypes.data, pR.ctypes.data)
    return rc == 1, info, mp[:int(nm[0])].copy(), pT, pR


def o_optimize4dof(t, R, seq, loop_to, loop_info):
    L = olib()
    t, R = np.ascontiguousarray(t, np.float64), np.ascontiguousarray(R, np.float64).reshape(-1, 9)
    n = len(t)
    sq, lt, li = np.ascontiguousarray(seq, np.int32), np.ascontiguousarray(loop_to, np.int32), np.ascontiguousarray(loop_info, np.float64)
    to, Ro, dr = np.zeros((n, 3)), np.zeros((n, 9)), np.zeros(4)
    L.ovio_pg_optimize4dof(n, t.ctypes.data, R.ctypes.data, sq.ctypes.data, lt.ctypes.data, li.ctypes.data, to.ctypes.data, Ro.ctypes.data, dr.ctypes.data)
    return to, Ro.reshape(n, 3, 3), dr


def rot_zyx(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    return np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]]) @ np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]]) @ np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])


def test_blur_is_the_fixed_point_gaussian():
    L = olib()
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (60, 80), dtype=np.uint8)
    out = np.zeros_like(img)
    L.ovio_pg_blur(img.ctypes.data, 80, 60, out.ctypes.data)
    # definition: exp(-x^2 / (2 sigma^2)) normalised, 8 fractional bits per pass, REFLECT_101, one rounding at the end
    k = np.exp(-np.arange(-4, 5) ** 2 / 8.0)
    ki = np.rint(k / k.sum() * 256).astype(np.int64)
    assert ki.sum() == 256 and list(ki) == [7, 17, 32, 46, 52, 46, 32, 17, 7]
    pad = np.pad(img.astype(np.int64), 4, mode="reflect")
    h = sum(ki[i] * pad[:, i:i + 80] for i in range(9))
    v = sum(ki[i] * h[i:i + 60, :] for i in range(9))
    assert np.array_equal(out, ((v + (1 << 15)) >> 16).astype(np.uint8))
    flat = np.full((40, 40), 93, np.uint8); o2 = np.zeros_like(flat)
    L.ovio_pg_blur(flat.ctypes.data, 40, 40, o2.ctypes.data)
    assert np.all(o2 == 93)


def test_brief_bits_and_hamming_search_follow_their_definitions():
    P = vio_ct.pkg()
    cfg = P.canonical_config()
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (cfg.height, cfg.width), dtype=np.uint8)
    pat = pattern()
    uv = np.array([[320.4, 240.7], [3.2, 5.9], [636.5, 470.1], [100.0, 200.0]], np.float32)   # interior, two corners (out-of-image pairs), integer
    wd, kxy, kd, kn = o_describe(cfg, img, uv, pat)
    L = olib()
    blur = np.zeros_like(img)
    L.ovio_pg_blur(img.ctypes.data, cfg.width, cfg.height, blur.ctypes.data)
    for p in range(len(uv)):
        bits = 0
        for i in range(256):
            x1, y1 = int(np.float32(uv[p, 0]) + np.float32(pat[i])), int(np.float32(uv[p, 1]) + np.float32(pat[256 + i]))
            x2, y2 = int(np.float32(uv[p, 0]) + np.float32(pat[512 + i])), int(np.float32(uv[p, 1]) + np.float32(pat[768 + i]))
            if 0 <= x1 < cfg.width and 0 <= y1 < cfg.height and 0 <= x2 < cfg.width and 0 <= y2 < cfg.height and blur[y1, x1] < blur[y2, x2]:
                bits |= 1 << i
        got = sum(int(wd[p, q]) << (64 * q) for q in range(4))
        assert got == bits, p
    assert bin(int(wd[1, 0])).count("1") + bin(int(wd[1, 1])).count("1") < 128          # the corner point lost its out-of-image pairs
    # keypoints: row-major order, FAST threshold 20 on the RAW image; a random image has thousands
    assert len(kxy) > 500 and np.all(np.diff(kxy[:, 1] * cfg.width + kxy[:, 0]) > 0)
    # Hamming search against numpy: first smallest distance, accepted below 80
    a = kd[:40].copy()
    b = kd[20:400].copy()
    b[5, 0] ^= np.uint64(0xFF)    # 8 bits away from its original = a[25]
    bi, bd = o_match(a, b)
    pc = np.array([[sum(bin(int(x ^ y)).count("1") for x, y in zip(ra, rb)) for rb in b] for ra in a])
    exp_i = pc.argmin(1)
    exp_d = pc.min(1)
    assert np.array_equal(bd, np.minimum(exp_d, 128))
    assert np.array_equal(bi, np.where(exp_d < 80, exp_i, -1))
    assert list(bi[20:40]) == list(range(0, 20)) and bd[25] == 8 and np.all(bd[20:25] == 0)


def _loop_scene(rng, n=120, outliers=12, yaw_deg=8.0):
    """two camera poses looking at one cloud; returns the inputs of findConnection and the true relative pose"""
    qic = rot_zyx(0.01, -0.02, 0.015) @ np.array([[0, 0, 1.0], [-1, 0, 0], [0, -1, 0]])   # camera z forward = body x
    tic = np.array([0.05, -0.02, 0.03])
    R_cur, T_cur = rot_zyx(0.3, 0.02, -0.01), np.array([1.0, 2.0, 0.5])
    R_old, T_old = rot_zyx(0.3 - np.radians(yaw_deg), 0.015, -0.02), np.array([1.25, 1.9, 0.55])
    # points in front of both cameras
    pc = np.c_[rng.uniform(-1.5, 1.5, n), rng.uniform(-1.0, 1.0, n), rng.uniform(2.5, 6.0, n)]
    Rwc, Twc = R_cur @ qic, T_cur + R_cur @ tic
    world = (Rwc @ pc.T).T + Twc
    Rwo, Two = R_old @ qic, T_old + R_old @ tic
    po = (Rwo.T @ (world - Two).T).T
    old_norm_true = po[:, :2] / po[:, 2:3]
    m_old = 400
    old_norm = np.zeros((m_old, 2), np.float32)
    slots = rng.permutation(m_old)[:n]
    old_norm[slots] = old_norm_true + rng.normal(0, 0.0008, (n, 2))
    match = slots.astype(np.int32)
    bad = rng.permutation(n)[:outliers]
    old_norm[slots[bad]] += rng.uniform(0.2, 0.5, (outliers, 2)).astype(np.float32)
    ids = np.arange(1000, 1000 + n, dtype=np.float64)
    return dict(p3=world.astype(np.float32), pn=(pc[:, :2] / pc[:, 2:3]).astype(np.float32), ids=ids, match=match, old_norm=old_norm, T=T_cur, R=R_cur,
                qic=qic, tic=tic, R_old=R_old, T_old=T_old, bad=set(bad.tolist()))


def test_find_connection_recovers_the_true_relative_pose_and_rejects_outliers():
    s = _loop_scene(np.random.default_rng(11))
    ok, info, mp, pT, pR = o_find_connection(s["p3"], s["pn"], s["ids"], s["match"], s["old_norm"], s["T"], s["R"], s["qic"], s["tic"])
    assert ok
    rel_t = s["R_old"].T @ (s["T"] - s["T_old"])
    assert np.abs(info[:3] - rel_t).max() < 0.02 and np.abs(pT - s["T_old"]).max() < 0.02
    assert abs(info[7] - 8.0) < 0.3                                              # relative yaw (degrees)
    kept = set(mp[:, 2].astype(int) - 1000)
    assert not (kept & s["bad"]) and len(kept) >= 100                              # gross outliers gone, inliers kept
    assert np.all(np.diff(mp[:, 2]) > 0)                                          # ascending feature id (window order)
    # gates: too few matches, and a relative yaw beyond 30 degrees
    few = s["match"].copy(); few[20:] = -1
    assert not o_find_connection(s["p3"], s["pn"], s["ids"], few, s["old_norm"], s["T"], s["R"], s["qic"], s["tic"])[0]
    s2 = _loop_scene(np.random.default_rng(12), yaw_deg=35.0)
    ok2, info2, mp2, _, _ = o_find_connection(s2["p3"], s2["pn"], s2["ids"], s2["match"], s2["old_norm"], s2["T"], s2["R"], s2["qic"], s2["tic"])
    assert not ok2 and len(mp2) == 0


def _drift_graph(n=240, loop_back=True):
    """a closed circuit: true poses, VIO poses with accumulated yaw / position drift, a loop edge from the last node to node 0 measured truly"""
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    t_true = np.c_[4 * np.cos(ang), 4 * np.sin(ang), 0.2 * np.sin(2 * ang)]
    R_true = np.array([rot_zyx(a + np.pi / 2, 0.02 * np.sin(a), 0.03 * np.cos(a)) for a in ang])
    # VIO: rotate / shift increments by a growing yaw error and a position error
    t_vio, R_vio = [t_true[0]], [R_true[0]]
    for i in range(1, n):
        dR = R_true[i - 1].T @ R_true[i]
        dt = R_true[i - 1].T @ (t_true[i] - t_true[i - 1])
        dRe = rot_zyx(np.radians(0.02), 0, 0) @ dR               # 0.02 degree yaw drift per step
        t_vio.append(t_vio[-1] + R_vio[-1] @ (dt * 1.002 + np.array([0.0008, 0, 0])))
        R_vio.append(R_vio[-1] @ dRe)
    t_vio, R_vio = np.array(t_vio), np.array(R_vio)
    seq = np.ones(n, np.int32)
    loop_to = -np.ones(n, np.int32)
    info = np.zeros((n, 8))
    if loop_back:
        i, c = n - 1, 0
        yaw = lambda R: np.degrees(np.arctan2(R[1, 0], R[0, 0]))
        rel_t = R_true[c].T @ (t_true[i] - t_true[c])
        info[i, :3] = rel_t
        info[i, 7] = ((yaw(R_true[i]) - yaw(R_true[c]) + 180) % 360) - 180
        loop_to[i] = c
    return t_true, R_true, t_vio, R_vio, seq, loop_to, info


def test_optimize4dof_closes_the_loop_towards_the_truth():
    t_true, R_true, t_vio, R_vio, seq, loop_to, info = _drift_graph()
    to, Ro, drift = o_optimize4dof(t_vio, R_vio, seq, loop_to, info)
    e0 = np.linalg.norm(t_vio - t_true, axis=1)
    e1 = np.linalg.norm(to - t_true, axis=1)
    assert e0[-1] > 0.2                                   # the drift is real
    # (a chain of n keyframes with four sequential edges each resists a loop edge of equal weight with stiffness 30 / n: the optimum leaves
    # about (30 / n) / (30 / n + w) of the gap open, w <= 1 under the Huber loss -- a quarter at n = 240)
    assert e1[-1] < 0.4 * e0[-1] and e1.mean() < 1.2 * e0.mean(), (e0[-1], e1[-1], e0.mean(), e1.mean())
    assert np.allclose(to[0], t_vio[0]) and np.allclose(Ro[0], R_vio[0], atol=1e-12)   # the earliest looped keyframe is constant
    # pitch and roll are not touched (4-DoF)
    pr = lambda R: (np.arctan2(-R[2, 0], np.hypot(R[0, 0], R[1, 0])), np.arctan2(R[2, 1], R[2, 2]))
    assert max(abs(pr(Ro[k])[0] - pr(R_vio[k])[0]) + abs(pr(Ro[k])[1] - pr(R_vio[k])[1]) for k in range(len(to))) < 1e-9
    # drift output = pose correction of the newest keyframe: r_drift * vio + t_drift lands on its optimised position
    yd = np.radians(drift[0])
    Rd = np.array([[np.cos(yd), -np.sin(yd), 0], [np.sin(yd), np.cos(yd), 0], [0, 0, 1]])
    assert np.abs(Rd @ t_vio[-1] + drift[1:] - to[-1]).max() < 1e-9
    # without a loop edge nothing moves
    _, _, t_vio2, R_vio2, seq2, lt2, info2 = _drift_graph(loop_back=False)
    to2, _, d2 = o_optimize4dof(t_vio2, R_vio2, seq2, lt2, info2)
    assert np.abs(to2 - t_vio2).max() < 1e-6 and abs(d2[0]) < 1e-6


def o_optimize6dof(t, R, seq, loop_to, loop_info):
    L = olib()
    t, R = np.ascontiguousarray(t, np.float64).reshape(-1, 3), np.ascontiguousarray(R, np.float64).reshape(-1, 9)
    n = len(t)
    sq, lt, li = np.ascontiguousarray(seq, np.int32), np.ascontiguousarray(loop_to, np.int32), np.ascontiguousarray(loop_info, np.float64).reshape(n, 8)
    to, Ro, dr = np.zeros((n, 3)), np.zeros((n, 9)), np.zeros(12)
    L.ovio_pg_optimize6dof(n, t.ctypes.data, R.ctypes.data, sq.ctypes.data, lt.ctypes.data, li.ctypes.data, to.ctypes.data, Ro.ctypes.data, dr.ctypes.data)
    return to, Ro.reshape(n, 3, 3), dr


def _R2q(R):
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    return np.array([w, (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)])


def _drift_graph6(n=120):
    """the circuit of _drift_graph with drift in all six degrees of freedom (the `imu: 0` case: nothing anchors pitch and roll) and two loop
    edges measured truly as (relative t, relative q)"""
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    t_true = np.c_[4 * np.cos(ang), 4 * np.sin(ang), 0.2 * np.sin(2 * ang)]
    R_true = np.array([rot_zyx(a + np.pi / 2, 0.05 * np.sin(a), 0.04 * np.cos(a)) for a in ang])
    t_vio, R_vio = [t_true[0]], [R_true[0]]
    for i in range(1, n):
        dR = R_true[i - 1].T @ R_true[i]
        dt = R_true[i - 1].T @ (t_true[i] - t_true[i - 1])
        t_vio.append(t_vio[-1] + R_vio[-1] @ (dt * 1.003 + np.array([0.001, 0, 0.0005])))
        R_vio.append(R_vio[-1] @ (rot_zyx(np.radians(0.03), np.radians(0.02), np.radians(-0.015)) @ dR))
    t_vio, R_vio = np.array(t_vio), np.array(R_vio)
    seq, loop_to, info = np.ones(n, np.int32), -np.ones(n, np.int32), np.zeros((n, 8))
    for (i, c) in ((n - 1, 0), (n - 3, 2)):
        info[i, :3] = R_true[c].T @ (t_true[i] - t_true[c])
        info[i, 3:7] = _R2q(R_true[c].T @ R_true[i])
        loop_to[i] = c
    return t_true, R_true, t_vio, R_vio, seq, loop_to, info


def test_optimize6dof_closes_the_loop_in_all_six_degrees_of_freedom():
    """PoseGraph::optimize6DoF (pose_graph.cpp:583-740): RelativeRTError edges, quaternion parameterisation"""
    t_true, R_true, t_vio, R_vio, seq, loop_to, info = _drift_graph6()
    to, Ro, dr = o_optimize6dof(t_vio, R_vio, seq, loop_to, info)
    e0, e1 = np.linalg.norm(t_vio - t_true, axis=1), np.linalg.norm(to - t_true, axis=1)
    ang = lambda A, B: np.degrees(np.arccos(np.clip((np.trace(A.T @ B) - 1) / 2, -1, 1)))
    a0, a1 = ang(R_vio[-1], R_true[-1]), ang(Ro[-1], R_true[-1])
    assert e0[-1] > 0.2 and a0 > 3.0                                       # the drift is real, rotation included
    # position and ATTITUDE move towards the truth (4-DoF keeps pitch / roll) -- by a quarter in five iterations: the sequential edges weigh a
    # rotation 200 / rad (q_var 0.01) while the loop edges sit deep in the Huber regime (weight sqrt(0.1 / |r|) = 0.13 at 3.5 degrees)
    assert e1[-1] < 0.85 * e0[-1] and a1 < 0.95 * a0, (e0[-1], e1[-1], a0, a1)
    assert np.allclose(to[0], t_vio[0]) and np.allclose(Ro[0], R_vio[0], atol=1e-12)
    assert np.abs(Ro @ Ro.transpose(0, 2, 1) - np.eye(3)).max() < 1e-12    # the quaternions stay unit
    rd, td = dr[:9].reshape(3, 3), dr[9:]
    assert np.abs(rd @ t_vio[-1] + td - to[-1]).max() < 1e-9 and np.abs(rd @ R_vio[-1] - Ro[-1]).max() < 1e-9
    # one loop edge alone (two nodes of different sequences: no sequential edge) is solved exactly: the Jacobians are the residual's
    R0, t0, Rm, tm = rot_zyx(0.3, 0.05, -0.02), np.array([1.0, 2.0, 0.5]), rot_zyx(0.1, -0.03, 0.04), np.array([0.3, -0.2, 0.1])
    R1t, t1t = R0 @ Rm, t0 + R0 @ tm
    inf2 = np.zeros((2, 8)); inf2[1, :3] = tm; inf2[1, 3:7] = _R2q(Rm)
    tq, Rq, _ = o_optimize6dof(np.array([t0, t1t + [0.03, -0.02, 0.01]]), np.array([R0, R1t @ rot_zyx(0.02, 0.01, -0.015)]), np.array([1, 2], np.int32),
                               np.array([-1, 0], np.int32), inf2)
    assert np.abs(tq[1] - t1t).max() < 1e-12 and np.abs(Rq[1] - R1t).max() < 1e-12
    # a consistent graph (VIO = truth, loops measured truly) is a fixed point; sequence 0 stays constant
    to2, Ro2, _ = o_optimize6dof(t_true, R_true, seq, loop_to, info)
    assert np.abs(to2 - t_true).max() < 1e-9 and np.abs(Ro2 - R_true).max() < 1e-9
    seq0 = seq.copy(); seq0[:10] = 0
    to3, _, _ = o_optimize6dof(t_vio, R_vio, seq0, loop_to, info)
    assert np.abs(to3[:10] - t_vio[:10]).max() == 0
