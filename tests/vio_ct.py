"""Test helpers: ctypes binding of the CPU oracle (oracle/liboracle.so), sequence drivers and trajectory metrics.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/ (it is the checker, never the
thing measured or shipped)."""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")


def pkg():
    P = importlib.import_module("vins-rgbd-fast_amd")
    if not os.path.exists(os.path.join(ROOT, "vins-rgbd-fast_amd", "libvio_hip.so")):
        P.build()  # fresh checkout without built artefacts: hipcc cross-compiles for gfx950 (no CPU fallback either way)
    return P


def build_oracle(target="liboracle.so"):
    r = subprocess.run(["make", "-C", ORACLE_DIR, target], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    return ORACLE_SO


_orc = {}


def _stale(lib_path):
    t = os.path.getmtime(lib_path)
    return any(os.path.getmtime(os.path.join(ORACLE_DIR, f)) > t + 1.0 for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".h")))


def oracle(path=None):
    """ctypes handle of the oracle library; path: another build of the SAME sources (oracle/Makefile `control`: liboracle_fma.so,
    liboracle_order.so), used only by the self-divergence control experiment (tests/oracle_control.py)"""
    path = os.path.abspath(path or ORACLE_SO)
    if path not in _orc:
        if os.path.dirname(path) == os.path.abspath(ORACLE_DIR) and (not os.path.exists(path) or _stale(path)):
            build_oracle(os.path.basename(path))   # missing, or older than a source file of oracle/ (a control build left over from before an edit)
        L = C.CDLL(path)
        L.ovio_pipeline_create.restype = C.c_void_p
        L.ovio_tracker_create.restype = C.c_void_p
        L.ovio_preint_create.restype = C.c_void_p
        L.ovio_gate_create.restype = C.c_void_p
        for name, args in {
            "ovio_pipeline_create": [C.c_void_p], "ovio_pipeline_destroy": [C.c_void_p], "ovio_pipeline_restart": [C.c_void_p],
            "ovio_set_relo_frame": [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p], "ovio_get_relo": [C.c_void_p, C.c_void_p],
            "ovio_pair_color_depth": [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
            "ovio_push_imu_n": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
            "ovio_feed": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double],
            "ovio_feed_mode": [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int],
            "ovio_track": [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p],
            "ovio_process_obs": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double],
            "ovio_predict_motion": [C.c_void_p, C.c_double, C.c_double, C.c_void_p],
            "ovio_latest_odometry": [C.c_void_p, C.c_void_p],
            "ovio_set_tracker_lag": [C.c_void_p, C.c_int],
            "ovio_set_fisheye_mask": [C.c_void_p, C.c_void_p],
            "ovio_get_landmarks_ex": [C.c_void_p, C.c_int, C.c_void_p],
            "ovio_gate_create": [C.c_int, C.c_int], "ovio_gate_destroy": [C.c_void_p], "ovio_gate_step": [C.c_void_p, C.c_double],
            "ovio_gate_empty_map": [C.c_void_p, C.c_double],
            "ovio_get_status": [C.c_void_p, C.c_void_p], "ovio_get_bound_stats": [C.c_void_p, C.c_void_p], "ovio_get_line_search_stats": [C.c_void_p, C.c_void_p], "ovio_get_window": [C.c_void_p, C.c_void_p],
            "ovio_get_extrinsic": [C.c_void_p, C.c_void_p], "ovio_get_landmarks": [C.c_void_p, C.c_int, C.c_void_p],
            "ovio_get_tracks": [C.c_void_p, C.c_int] + [C.c_void_p] * 5,
            "ovio_tracker_create": [C.c_void_p], "ovio_tracker_destroy": [C.c_void_p],
            "ovio_tracker_read": [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int],
            "ovio_tracker_get": [C.c_void_p, C.c_int] + [C.c_void_p] * 5,
            "ovio_tracker_grid": [C.c_void_p, C.c_void_p, C.c_void_p],
            "ovio_cam_lift": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p],
            "ovio_cam_project": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p],
            "ovio_pyr_down": [C.c_void_p, C.c_int, C.c_int, C.c_void_p],
            "ovio_clahe": [C.c_void_p, C.c_int, C.c_int, C.c_void_p],
            "ovio_fast_score": [C.c_void_p],
            "ovio_fast_roi": [C.c_void_p] + [C.c_int] * 7 + [C.c_void_p],
            "ovio_circle_hw": [C.c_int, C.c_void_p],
            "ovio_lk": [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int],
            "ovio_ransac": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
            "ovio_eval_projection": [C.c_void_p] * 4 + [C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p],
            "ovio_preint_create": [C.c_void_p] * 5, "ovio_preint_destroy": [C.c_void_p],
            "ovio_preint_push": [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p],
            "ovio_preint_repropagate": [C.c_void_p, C.c_void_p, C.c_void_p], "ovio_preint_get": [C.c_void_p, C.c_void_p],
            "ovio_eval_imu": [C.c_void_p, C.c_double] + [C.c_void_p] * 6,
            "ovio_sym_eig": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
            "ovio_get_prior": [C.c_void_p] * 5,
            "ovio_marg_finish": [C.c_int, C.c_int] + [C.c_void_p] * 4,
            "ovio_sincos_det": [C.c_int] + [C.c_void_p] * 3, "ovio_set_deviations": [C.c_int],
        }.items():
            getattr(L, name).argtypes = args
        assert L.ovio_config_size() == C.sizeof(pkg().Config), "oracle Config and vio_config layouts differ"
        _orc[path] = L
    return _orc[path]


class OraclePipeline:
    def __init__(self, cfg, lib=None):
        self.L = oracle(lib)
        self.cfg = cfg
        self.W = cfg.window_size
        self.h = C.c_void_p(self.L.ovio_pipeline_create(C.byref(cfg)))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ovio_pipeline_destroy(self.h)
            self.h = None

    def push_imu(self, t, acc, gyr):
        t = np.ascontiguousarray(t, np.float64); acc = np.ascontiguousarray(acc, np.float64); gyr = np.ascontiguousarray(gyr, np.float64)
        self.L.ovio_push_imu_n(self.h, len(t), t.ctypes.data, acc.ctypes.data, gyr.ctypes.data)

    def restart(self):
        """the stream-discontinuity branch of process_tracker (estimator_nodelet.cpp:243-262): estimator restarted, tracker kept"""
        self.L.ovio_pipeline_restart(self.h)

    def set_relo_frame(self, stamp, index, match_points, relo_t, relo_r):
        """Estimator::setReloFrame (estimator.cpp:1728-1747): match_points[n][3] = (x, y, feature id) ascending in id"""
        mp = np.ascontiguousarray(match_points, np.float64).reshape(-1, 3)
        t = np.ascontiguousarray(relo_t, np.float64); R = np.ascontiguousarray(relo_r, np.float64)
        self.L.ovio_set_relo_frame(self.h, float(stamp), int(index), len(mp), mp.ctypes.data, t.ctypes.data, R.ctypes.data)

    def relo(self):
        o = np.zeros(30)
        self.L.ovio_get_relo(self.h, o.ctypes.data)
        return dict(relative_t=o[0:3], relative_q=o[3:7], relative_yaw=o[7], drift_t=o[8:11], drift_r=o[11:20].reshape(3, 3), relo_pose=o[20:27],
                    pending=int(o[27]), local_index=int(o[28]), n_factors=int(o[29]))

    def feed(self, gray, depth, t, mode=2):
        return self.L.ovio_feed_mode(self.h, gray.ctypes.data, depth.ctypes.data, float(t), int(mode))

    def track(self, gray, t, mode=2, R=None, cap=2048):
        """the tracker half (process_tracker): returns (ids, obs[n][7]) of the packaged feature map (empty = nothing to process)"""
        ids, obs = np.zeros(cap, np.int32), np.zeros((cap, 7))
        Rp = None if R is None else np.ascontiguousarray(R, np.float64)
        n = self.L.ovio_track(self.h, gray.ctypes.data, float(t), int(mode), None if Rp is None else Rp.ctypes.data, cap, ids.ctypes.data,
                              obs.ctypes.data)
        return ids[:n].copy(), obs[:n].copy()

    def process_obs(self, ids, obs, depth, t):
        ids = np.ascontiguousarray(ids, np.int32); obs = np.ascontiguousarray(obs, np.float64)
        return self.L.ovio_process_obs(self.h, len(ids), ids.ctypes.data, obs.ctypes.data, depth.ctypes.data, float(t))

    def predict_motion(self, t0, t1):
        R = np.zeros(9)
        self.L.ovio_predict_motion(self.h, float(t0), float(t1), R.ctypes.data)
        return R.reshape(3, 3)

    def set_tracker_lag(self, lag):
        self.L.ovio_set_tracker_lag(self.h, int(lag))

    def set_fisheye_mask(self, mask):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.L.ovio_set_fisheye_mask(self.h, None if m is None else m.ctypes.data)

    def latest_odometry(self):
        o = np.zeros(11)
        self.L.ovio_latest_odometry(self.h, o.ctypes.data)
        return o

    def landmarks_ex(self, cap=4096):
        out = np.zeros((cap, 12))
        n = self.L.ovio_get_landmarks_ex(self.h, cap, out.ctypes.data)
        return out[:min(n, cap)]

    def bound_stats(self):
        """(candidate steps cut by the inverse-depth upper bound, bounded landmarks that entered solves) since construction"""
        o = np.zeros(2)
        self.L.ovio_get_bound_stats(self.h, o.ctypes.data)
        return int(o[0]), int(o[1])

    def line_search_stats(self):
        """(trial evaluations, shortened steps) of the Armijo line search of bounds-constrained solves since construction"""
        o = np.zeros(2)
        self.L.ovio_get_line_search_stats(self.h, o.ctypes.data)
        return int(o[0]), int(o[1])

    def status(self):
        s = np.zeros(16)
        self.L.ovio_get_status(self.h, s.ctypes.data)
        keys = ["solver_flag", "frame_count", "marginalization_flag", "td", "n_landmarks", "last_track_num", "reboot_count",
                "frames_processed", "iterations", "successful_steps", "initial_cost", "final_cost", "n_in_problem", "n_residuals",
                "n_var_landmarks", "has_prior"]
        return dict(zip(keys, s))

    def window(self):
        w = np.zeros((self.W + 1, 17))
        self.L.ovio_get_window(self.h, w.ctypes.data)
        return w

    def tracks(self, cap=2048):
        ids, cnt = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        cur, un, vel = np.zeros((cap, 2), np.float32), np.zeros((cap, 2), np.float32), np.zeros((cap, 2), np.float32)
        n = self.L.ovio_get_tracks(self.h, cap, ids.ctypes.data, cnt.ctypes.data, cur.ctypes.data, un.ctypes.data, vel.ctypes.data)
        return ids[:n], cnt[:n], cur[:n], un[:n], vel[:n]

    def landmarks(self, cap=4096):
        out = np.zeros((cap, 7))
        n = self.L.ovio_get_landmarks(self.h, cap, out.ctypes.data)
        return out[:min(n, cap)]

    def prior(self):
        n = 6 * self.W + 16
        J, r, x0, pres = np.zeros((n, n)), np.zeros(n), np.zeros(self.W * 7 + 17), np.zeros(self.W + 3, np.uint8)
        k = self.L.ovio_get_prior(self.h, J.ctypes.data, r.ctypes.data, x0.ctypes.data, pres.ctypes.data)
        return (J, r, x0, pres) if k else None


def oracle_pair_color_depth(color, depth):
    """oracle restatement of the nodelet's colour / depth pairing (estimator_nodelet.cpp:200-232): ([(i, j)], thrown colour, thrown depth)"""
    L = oracle()
    tc, td = np.ascontiguousarray(color, np.float64), np.ascontiguousarray(depth, np.float64)
    pairs, thrown = np.zeros(2 * max(1, min(len(tc), len(td))), np.int32), np.zeros(2, np.int32)
    n = L.ovio_pair_color_depth(len(tc), tc.ctypes.data, len(td), td.ctypes.data, pairs.ctypes.data, thrown.ctypes.data)
    return [(int(pairs[2 * k]), int(pairs[2 * k + 1])) for k in range(n)], int(thrown[0]), int(thrown[1])


class OracleGate:
    """oracle restatement of the nodelet's frequency control (estimator_nodelet.cpp:234-286)"""
    SKIP, TRACK, PUBLISH, FIRST, RESET = 0, 1, 2, 3, 4

    def __init__(self, freq, frontend_freq):
        self.L = oracle()
        self.h = C.c_void_p(self.L.ovio_gate_create(int(freq), int(frontend_freq)))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ovio_gate_destroy(self.h)
            self.h = None

    def step(self, t):
        return self.L.ovio_gate_step(self.h, float(t))

    def empty_map(self, t):
        self.L.ovio_gate_empty_map(self.h, float(t))


class OracleTracker:
    def __init__(self, cfg):
        self.L = oracle()
        self.h = C.c_void_p(self.L.ovio_tracker_create(C.byref(cfg)))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ovio_tracker_destroy(self.h)
            self.h = None

    def read(self, gray, t, R=None, publish=True):
        R = np.eye(3) if R is None else np.ascontiguousarray(R, np.float64)
        self.L.ovio_tracker_read(self.h, gray.ctypes.data, float(t), R.ctypes.data, 1 if publish else 0)

    def tracks(self, cap=2048):
        ids, cnt = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        cur, un, vel = np.zeros((cap, 2), np.float32), np.zeros((cap, 2), np.float32), np.zeros((cap, 2), np.float32)
        n = self.L.ovio_tracker_get(self.h, cap, ids.ctypes.data, cnt.ctypes.data, cur.ctypes.data, un.ctypes.data, vel.ctypes.data)
        return ids[:n], cnt[:n], cur[:n], un[:n], vel[:n]


def synth_like(cfg, **kw):
    """vio_synth_config consistent with a vio_config (same intrinsics / extrinsics / gravity)."""
    P = pkg()
    sc = P.default_synth(**kw)
    sc.width, sc.height = cfg.width, cfg.height
    for k in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "g_norm"):
        setattr(sc, k, getattr(cfg, k))
    for i in range(9):
        sc.ric[i] = cfg.ric[i]
    for i in range(3):
        sc.tic[i] = cfg.tic[i]
    return sc


def frame_times(sc, n):
    return np.arange(n) / sc.cam_rate


def imu_until(t_imu, k0, t_frame, imu_rate):
    """index one past the last IMU sample to push before feeding the frame at t_frame (one sample beyond the stamp)."""
    k = k0
    lim = t_frame + 1.5 / imu_rate
    while k < len(t_imu) and t_imu[k] < lim:
        k += 1
    return k


def yaw_align(est, gt):
    """least-squares yaw + translation alignment of est onto gt (gravity-aligned 4-DoF), returns aligned est."""
    ec, gc = est - est.mean(0), gt - gt.mean(0)
    num = (ec[:, 0] * gc[:, 1] - ec[:, 1] * gc[:, 0]).sum()
    den = (ec[:, 0] * gc[:, 0] + ec[:, 1] * gc[:, 1]).sum()
    th = np.arctan2(num, den)
    c, s = np.cos(th), np.sin(th)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    return (R @ ec.T).T + gt.mean(0)


def ate_rmse(est, gt):
    al = yaw_align(np.asarray(est), np.asarray(gt))
    return float(np.sqrt(((al - gt) ** 2).sum(1).mean()))


def run_oracle_sequence(cfg, sc, seq, n_frames, frames=None, modes=None, hook=None, tracker_lag=0, lib=None, fisheye_mask=None):
    """Drive the oracle over n_frames of sequence seq. frames: optional list of (gray, depth) to reuse; modes: optional per-frame
    frame mode (0 skip / 1 track / 2 publish); hook(f, oracle): called after every frame.
    Returns dict(traj=[(frame, P(3), Q(4), V(3))], gt=..., status=[...], frames=[...])."""
    P = pkg()
    syn = P.Synth(sc)
    o = OraclePipeline(cfg, lib)
    if tracker_lag:
        o.set_tracker_lag(tracker_lag)
    if fisheye_mask is not None:
        o.set_fisheye_mask(fisheye_mask)
    nimu = int(n_frames / sc.cam_rate * sc.imu_rate) + 64
    ti, ai, gi = syn.imu(seq, nimu)
    k = 0
    out = dict(traj=[], gt=[], status=[], frames=[], processed=[])
    for f, tf in enumerate(frame_times(sc, n_frames)):
        k2 = imu_until(ti, k, tf, sc.imu_rate)
        if k2 > k:
            o.push_imu(ti[k:k2], ai[k:k2], gi[k:k2])
        k = k2
        if frames is not None:
            g, d = frames[f]
        else:
            g, d = syn.render_host(seq, tf)
        out["frames"].append((g, d))
        r = o.feed(g, d, tf, 2 if modes is None else int(modes[f]))
        st = o.status()
        out["status"].append(st)
        out["processed"].append(r)
        if hook is not None:
            hook(f, o)
        if st["solver_flag"] == 1 and r == 1:
            w = o.window()
            out["traj"].append((f, w[cfg.window_size, :3].copy(), w[cfg.window_size, 3:7].copy(), w[cfg.window_size, 7:10].copy()))
            out["gt"].append(syn.pose(seq, tf)[0])
    out["oracle"] = o
    return out


def run_hip_batch(P, cfg, sc, seqs, n_frames, frames, modes=None, hook=None, imu_batch=False, tracker_lag=0, fisheye_mask=None):
    """Drive a VioBatch over host frames (frames[i][f] = (gray, depth) of sequence seqs[i]) with IMU pushed frame by frame.
    modes: optional [n_frames] frame modes applied to every sequence; hook(f, batch) after every frame.
    Returns (batch, traj, stat): per sequence [(frame, P, Q, V)] and [vio_status per frame]."""
    syn = P.Synth(sc)
    S = len(seqs)
    b = P.VioBatch(cfg, S)
    if tracker_lag:
        b.set_tracker_lag(tracker_lag)
    if fisheye_mask is not None:
        b.set_fisheye_mask(fisheye_mask)
    nimu = int(n_frames / sc.cam_rate * sc.imu_rate) + 64
    imu = [syn.imu(s, nimu) for s in seqs]
    k = [0] * S
    traj = [[] for _ in seqs]
    stat = [[] for _ in seqs]
    for f, tf in enumerate(frame_times(sc, n_frames)):
        k2 = [imu_until(imu[i][0], k[i], tf, sc.imu_rate) for i in range(S)]
        if imu_batch:
            stride = max(max(k2[i] - k[i] for i in range(S)), 1)
            tt, aa, gg = np.zeros((S, stride)), np.zeros((S, stride, 3)), np.zeros((S, stride, 3))
            for i in range(S):
                m = k2[i] - k[i]
                tt[i, :m] = imu[i][0][k[i]:k2[i]]; aa[i, :m] = imu[i][1][k[i]:k2[i]]; gg[i, :m] = imu[i][2][k[i]:k2[i]]
            b.push_imu_batch(tt, aa, gg, n=[k2[i] - k[i] for i in range(S)])
        else:
            for i in range(S):
                if k2[i] > k[i]:
                    b.push_imu(i, imu[i][0][k[i]:k2[i]], imu[i][1][k[i]:k2[i]], imu[i][2][k[i]:k2[i]])
        k = k2
        gray = np.stack([frames[i][f][0] for i in range(S)])
        depth = np.stack([frames[i][f][1] for i in range(S)])
        b.feed(gray, depth, [tf] * S, modes=None if modes is None else [int(modes[f])] * S)
        for i in range(S):
            st = b.status(i)
            stat[i].append(st)
            if st.solver_flag == 1 and st.processed:
                w = b.window(i)
                traj[i].append((f, w[cfg.window_size, :3].copy(), w[cfg.window_size, 3:7].copy(), w[cfg.window_size, 7:10].copy()))
        if hook is not None:
            hook(f, b)
    return b, traj, stat


def gate_modes(gate, times):
    """frame modes (0 skip / 1 track / 2 publish) a frame gate assigns to the stamps `times`; FIRST counts as publish (the
    pipelines recognise the first image themselves), a RESET is not expected in a regular stream."""
    out = []
    for t in times:
        d = gate.step(float(t))
        assert d != 4, "unexpected stream discontinuity"
        out.append(2 if d == 3 else d)
    return out
