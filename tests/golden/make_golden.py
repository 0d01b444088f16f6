"""Generates the committed fixtures under tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

The reference holds no golden vectors and cannot be built or imported here, so the fixtures are
  * definitional: expected values computed by the independent numpy restatements in tests/test_oracle_kat.py
    (FAST from the ring definition, pyrDown from the 5x5 kernel) -- these pin the oracle;
  * regression: outputs of the CPU oracle (oracle/liboracle.so) on seeded inputs -- these pin the HIP path on the GPU box
    without running anything but the C ABI, and detect drift of the oracle itself.
A fixture is data only: inputs + expected outputs (npz)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import test_oracle_kat as K  # noqa: E402
import vio_ct  # noqa: E402


def fast_fixture(orc):
    rng = np.random.default_rng(101)
    img = rng.integers(80, 120, (64, 80)).astype(np.uint8)
    for _ in range(60):
        x, y, w, h = rng.integers(0, 72), rng.integers(0, 56), rng.integers(2, 12), rng.integers(2, 12)
        img[y:y + h, x:x + w] = rng.integers(0, 255)
    kp = np.array(K.fast_detect_def(img), np.int32)
    rois = np.array([[0, 0, 80, 64], [7, 5, 40, 33], [40, 20, 40, 44], [3, 3, 9, 9]], np.int32)
    exp = [np.array(K.fast_detect_def(np.ascontiguousarray(img[ry:ry + rh, rx:rx + rw])), np.int32).reshape(-1, 3) for rx, ry, rw, rh in rois]
    np.savez_compressed(os.path.join(HERE, "fast_definition.npz"), img=img, rois=rois, **{f"kp{i}": e for i, e in enumerate(exp)})
    print("fast:", [len(e) for e in exp])


def pyr_fixture():
    rng = np.random.default_rng(102)
    img = rng.integers(0, 256, (37, 53)).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "pyrdown_definition.npz"), img=img, out=K.pyr_down_def(img))


def lk_fixture(orc):
    w, h = 160, 120
    a, b = K._texture(w, h), K._texture(w, h, 1.7, -0.9)
    gx, gy = np.meshgrid(np.arange(14, 150, 12), np.arange(14, 110, 12))
    prev = np.ascontiguousarray(np.c_[gx.ravel(), gy.ravel()].astype(np.float32) + np.float32(0.375))
    prev = np.vstack([prev, [[1.5, 2.5], [158.0, 118.0], [-40.0, 5.0]]]).astype(np.float32)
    out = {}
    for lvl in (1, 3):
        nxt = (prev + np.float32(0.5)).astype(np.float32)
        st = np.zeros(len(prev), np.uint8)
        orc.ovio_lk(a.ctypes.data, b.ctypes.data, w, h, lvl, len(prev), prev.ctypes.data, nxt.ctypes.data, st.ctypes.data, 1)
        out[f"next{lvl}"], out[f"status{lvl}"] = nxt, st
    np.savez_compressed(os.path.join(HERE, "lk_regression.npz"), prev_img=a, next_img=b, prev=prev, **out)
    print("lk tracked:", int(out["status1"].sum()), "/", len(prev))


def factor_fixture(P, orc):
    cfg = P.default_config(tr=0.0)
    rng = np.random.default_rng(103)
    dt, acc, gyr, ba, bg, pi, sbi, pj, sbj = K._imu_inputs(rng)
    h = K._preint(orc, cfg, dt, acc, gyr, acc[0], gyr[0], ba, bg)
    pre = np.zeros(461)
    orc.ovio_preint_get(h, pre.ctypes.data)
    r, J = K._imu_eval(orc, h, cfg.g_norm, pi, sbi, pj, sbj)
    orc.ovio_preint_destroy(h)
    proj = []
    for use_td in (0, 1):
        for _ in range(4):
            qi = K._rand_pose(rng, 0.5)
            qj = K.pose_plus(qi, np.r_[rng.normal(0, 0.1, 3), rng.normal(0, 0.03, 3)])
            ex = K.pose_plus(np.r_[np.array(cfg.tic[:]), 0.5, -0.5, 0.5, -0.5], np.r_[np.zeros(3), rng.normal(0, 0.02, 3)])
            oi = np.r_[rng.uniform(-0.4, 0.4, 2), 1.0, rng.uniform(0, 640), rng.uniform(0, 480), rng.normal(0, 0.1, 2), 0.001, 2.0]
            oj = np.r_[rng.uniform(-0.4, 0.4, 2), 1.0, rng.uniform(0, 640), rng.uniform(0, 480), rng.normal(0, 0.1, 2), -0.002, 2.0]
            inv_dep, td = 1.0 / rng.uniform(1.5, 6.0), 0.003
            rr, JJ = K._proj_eval(orc, cfg, qi, qj, ex, inv_dep, td, oi, oj, use_td)
            proj.append(np.r_[use_td, inv_dep, td, qi, qj, ex, oi, oj, rr, JJ])
    np.savez_compressed(os.path.join(HERE, "factors_regression.npz"), dt=dt, acc=acc, gyr=gyr, ba=ba, bg=bg, pose_i=pi, sb_i=sbi, pose_j=pj,
                        sb_j=sbj, preint=pre, imu_r=r, imu_J=J, proj=np.array(proj))


def pipeline_fixture(P):
    """40 frames of synthetic sequence 3 (canonical bench config): the oracle's published state per frame."""
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    n = 40
    o = vio_ct.run_oracle_sequence(cfg, sc, 3, n)
    frames = np.array([x[0] for x in o["traj"]], np.int32)
    Pw = np.array([x[1] for x in o["traj"]])
    Qw = np.array([x[2] for x in o["traj"]])
    Vw = np.array([x[3] for x in o["traj"]])
    st = np.array([[s["solver_flag"], s["frame_count"], s["marginalization_flag"], s["n_landmarks"]] for s in o["status"]], np.int32)
    ids, cnt, cur, un, vel = o["oracle"].tracks()
    np.savez_compressed(os.path.join(HERE, "pipeline_regression.npz"), seq=3, n_frames=n, frames=frames, P=Pw, Q=Qw, V=Vw, status=st,
                        gt=np.array(o["gt"]), track_ids=ids, track_cnt=cnt, track_cur=cur)
    print("pipeline: published", len(frames), "ATE", vio_ct.ate_rmse(Pw, np.array(o["gt"])))


def dynamic_init_fixture(P):
    """Moving-start sequence through the oracle's static_init: 0 branch (SfM + visual-inertial alignment): the window right after the
    initialisation and the published trajectory.  Vectors for the HIP side of SURVEY.md 8f rank 1 (not built yet) and drift detection."""
    cfg = P.canonical_config()
    cfg.dynamic_init = 1
    sc = vio_ct.synth_like(cfg)
    sc.t_static = 0.0
    n = 30
    o = vio_ct.run_oracle_sequence(cfg, sc, 3, n)
    frames = np.array([x[0] for x in o["traj"]], np.int32)
    Pw = np.array([x[1] for x in o["traj"]])
    Qw = np.array([x[2] for x in o["traj"]])
    Vw = np.array([x[3] for x in o["traj"]])
    st = np.array([[s["solver_flag"], s["frame_count"], s["marginalization_flag"], s["n_landmarks"]] for s in o["status"]], np.int32)
    np.savez_compressed(os.path.join(HERE, "dynamic_init_regression.npz"), seq=3, n_frames=n, t_static=0.0, frames=frames, P=Pw, Q=Qw, V=Vw,
                        status=st, gt=np.array(o["gt"]))
    print("dynamic init: first published frame", frames[0], "ATE", vio_ct.ate_rmse(Pw, np.array(o["gt"])))


if __name__ == "__main__":
    P, orc = vio_ct.pkg(), vio_ct.oracle()
    fast_fixture(orc)
    pyr_fixture()
    lk_fixture(orc)
    factor_fixture(P, orc)
    pipeline_fixture(P)
    dynamic_init_fixture(P)
