"""Loop-closure slice of pose_graph (include/vio_posegraph.h): ctypes over the C ABI, shaped like the reference's KeyFrame.

    kf = KeyFrame(cfg, pattern, stamp, index, vio_T, vio_R, gray, point_3d, point_2d_uv, point_2d_norm, point_id)   # keyframe.cpp:14-43
    ok = kf.findConnection(old_kf, qic, tic)        # keyframe.cpp:252-528; on success kf.match_points / kf.loop_info are set
    batch.set_relo_frame(seq, kf.time_stamp, kf.index, kf.match_points, old_kf.T_w_i, old_kf.R_w_i)                   # Estimator::setReloFrame
    t, R, drift = optimize4DoF(t, R, sequence, loop_to, loop_info)                                                    # pose_graph.cpp:410-581

    voc = Vocabulary.load("brief_k10L6.bin")        # PoseGraph::loadVocabulary (the blob is missing from the reference tree: the caller's file)
    loop_index = voc.detectLoop(kf.brief_descriptors, kf.index)     # PoseGraph::detectLoop, pose_graph.cpp:308-393 (-1: none)

Descriptor extraction, matching and the vocabulary walk are HIP kernels; there is no CPU fallback."""
import ctypes as C
import importlib
import os
import re

import numpy as np

MIN_LOOP_NUM = 25   # pose_graph/src/utility/parameters.h


def _lib():
    P = importlib.import_module("vins-rgbd-fast_amd")
    L = P.lib()
    if not getattr(L, "_pg_bound", False):
        L.vio_pg_describe.argtypes = [C.POINTER(P.Config), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p]
        L.vio_pg_match.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.vio_pg_find_connection.argtypes = [C.c_int] + [C.c_void_p] * 8 + [C.c_int] + [C.c_void_p] * 5
        L.vio_pg_optimize4dof.argtypes = [C.c_int] + [C.c_void_p] * 8
        L.vio_pg_optimize6dof.argtypes = [C.c_int] + [C.c_void_p] * 8
        L.vio_pg_stage_blur.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.vio_pg_voc_load.argtypes = [C.c_char_p]
        L.vio_pg_voc_load.restype = C.c_void_p
        L.vio_pg_voc_create.argtypes = [C.c_int] * 5 + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 2
        L.vio_pg_voc_create.restype = C.c_void_p
        L.vio_pg_voc_destroy.argtypes = [C.c_void_p]
        L.vio_pg_voc_destroy.restype = None
        L.vio_pg_voc_info.argtypes = [C.c_void_p, C.c_void_p]
        L.vio_pg_voc_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.vio_pg_voc_bow.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.vio_pg_db_add.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.vio_pg_db_query.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.vio_pg_detect_loop.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4
        L._pg_bound = True
    return P, L


def _chk(P, L, rc, what):
    if rc < 0:
        raise P.VioError("%s failed (%d): %s" % (what, rc, L.vio_last_error().decode()))
    return rc


def load_brief_pattern(path):
    """BRIEF_PATTERN_FILE (support_files/brief_pattern.yml: OpenCV FileStorage sequences x1, y1, x2, y2 of 256 integers each), or an .npz
    holding the same four arrays: returns int32[1024] = x1 | y1 | x2 | y2 as the C ABI takes it (BriefExtractor::BriefExtractor,
    keyframe.cpp:582-597)."""
    if path.endswith(".npz"):
        z = np.load(path)
        out = np.concatenate([np.asarray(z[k], np.int32).reshape(-1) for k in ("x1", "y1", "x2", "y2")])
    else:
        txt = open(path).read()
        seqs = {}
        for name in ("x1", "y1", "x2", "y2"):
            m = re.search(r"^%s:\s*\n((?:\s*-\s*-?\d+\s*\n)+)" % name, txt, re.M)
            if m:
                seqs[name] = [int(v) for v in re.findall(r"-\s*(-?\d+)", m.group(1))]
            else:   # flow style: x1: [ 1, 2, ... ]
                m = re.search(r"^%s:\s*\[([^\]]*)\]" % name, txt, re.M)
                if not m:
                    raise ValueError("%s: no sequence %s" % (path, name))
                seqs[name] = [int(v) for v in m.group(1).replace("\n", " ").split(",") if v.strip()]
        out = np.concatenate([np.asarray(seqs[k], np.int32) for k in ("x1", "y1", "x2", "y2")])
    if out.shape != (1024,):
        raise ValueError("%s: expected 4 x 256 pattern entries, got %d" % (path, out.size))
    return np.ascontiguousarray(out)


def describe(cfg, gray, window_uv, pattern, fast_threshold=20, cap=8192):
    """computeWindowBRIEFPoint + computeBRIEFPoint: (window descriptors [n][4] u64, keypoints [m][2] f32, their descriptors, their normalised
    coordinates)"""
    P, L = _lib()
    gray = np.ascontiguousarray(gray, np.uint8)
    assert gray.shape == (cfg.height, cfg.width)
    uv = np.ascontiguousarray(window_uv, np.float32).reshape(-1, 2)
    n = len(uv)
    wd = np.zeros((max(n, 1), 4), np.uint64)
    kxy, kd, kn = np.zeros((cap, 2), np.float32), np.zeros((cap, 4), np.uint64), np.zeros((cap, 2), np.float32)
    pat = np.ascontiguousarray(pattern, np.int32)
    m = _chk(P, L, L.vio_pg_describe(C.byref(cfg), gray.ctypes.data, n, uv.ctypes.data, pat.ctypes.data, int(fast_threshold), wd.ctypes.data, cap,
                                     kxy.ctypes.data, kd.ctypes.data, kn.ctypes.data), "vio_pg_describe")
    if m > cap:
        # cv::FAST has no cap (keyframe.cpp:94-103): the compaction is row-major, a truncated list would lose the bottom of the image -- run
        # again with room for all of them
        return describe(cfg, gray, window_uv, pattern, fast_threshold, cap=int(m))
    return wd[:n], kxy[:m], kd[:m], kn[:m]


def match(window_desc, old_desc):
    """searchByBRIEFDes: (best index or -1 per window descriptor, best Hamming distance)"""
    P, L = _lib()
    a, b = np.ascontiguousarray(window_desc, np.uint64).reshape(-1, 4), np.ascontiguousarray(old_desc, np.uint64).reshape(-1, 4)
    bi, bd = np.zeros(max(len(a), 1), np.int32), np.zeros(max(len(a), 1), np.int32)
    _chk(P, L, L.vio_pg_match(a.ctypes.data, len(a), b.ctypes.data, len(b), bi.ctypes.data, bd.ctypes.data), "vio_pg_match")
    return bi[:len(a)], bd[:len(a)]


def find_connection(pt3d, pt_id, match_idx, old_norm, vio_T, vio_R, qic, tic, min_loop_num=MIN_LOOP_NUM):
    """findConnection after the descriptor search: (has_loop, loop_info[8], match_points[k][3], PnP_T_old, PnP_R_old)"""
    P, L = _lib()
    p3 = np.ascontiguousarray(pt3d, np.float32).reshape(-1, 3)
    n = len(p3)
    ids, mi = np.ascontiguousarray(pt_id, np.float64).reshape(-1), np.ascontiguousarray(match_idx, np.int32).reshape(-1)
    on = np.ascontiguousarray(old_norm, np.float32).reshape(-1, 2)
    T, R = np.ascontiguousarray(vio_T, np.float64), np.ascontiguousarray(vio_R, np.float64)
    q, t = np.ascontiguousarray(qic, np.float64), np.ascontiguousarray(tic, np.float64)
    info, mp, nm, pT, pR = np.zeros(8), np.zeros((max(n, 1), 3)), np.zeros(1, np.int32), np.zeros(3), np.zeros((3, 3))
    rc = _chk(P, L, L.vio_pg_find_connection(n, p3.ctypes.data, ids.ctypes.data, mi.ctypes.data, on.ctypes.data, T.ctypes.data, R.ctypes.data, q.ctypes.data,
                                             t.ctypes.data, int(min_loop_num), info.ctypes.data, mp.ctypes.data, nm.ctypes.data, pT.ctypes.data, pR.ctypes.data),
              "vio_pg_find_connection")
    return rc == 1, info, mp[:int(nm[0])].copy(), pT, pR


def optimize4DoF(t, R, sequence, loop_to, loop_info):
    """PoseGraph::optimize4DoF over the nodes earliest-looped .. current: (t_out[n][3], R_out[n][3][3], (yaw_drift_deg, t_drift))"""
    P, L = _lib()
    t, R = np.ascontiguousarray(t, np.float64).reshape(-1, 3), np.ascontiguousarray(R, np.float64).reshape(-1, 9)
    n = len(t)
    sq, lt = np.ascontiguousarray(sequence, np.int32).reshape(n), np.ascontiguousarray(loop_to, np.int32).reshape(n)
    li = np.ascontiguousarray(loop_info, np.float64).reshape(n, 8)
    to, Ro, dr = np.zeros((n, 3)), np.zeros((n, 9)), np.zeros(4)
    _chk(P, L, L.vio_pg_optimize4dof(n, t.ctypes.data, R.ctypes.data, sq.ctypes.data, lt.ctypes.data, li.ctypes.data, to.ctypes.data, Ro.ctypes.data,
                                     dr.ctypes.data), "vio_pg_optimize4dof")
    return to, Ro.reshape(n, 3, 3), (dr[0], dr[1:].copy())


def optimize6DoF(t, R, sequence, loop_to, loop_info):
    """PoseGraph::optimize6DoF (`imu: 0`): (t_out[n][3], R_out[n][3][3], (r_drift[3][3], t_drift[3]))"""
    P, L = _lib()
    t, R = np.ascontiguousarray(t, np.float64).reshape(-1, 3), np.ascontiguousarray(R, np.float64).reshape(-1, 9)
    n = len(t)
    sq, lt = np.ascontiguousarray(sequence, np.int32).reshape(n), np.ascontiguousarray(loop_to, np.int32).reshape(n)
    li = np.ascontiguousarray(loop_info, np.float64).reshape(n, 8)
    to, Ro, dr = np.zeros((n, 3)), np.zeros((n, 9)), np.zeros(12)
    _chk(P, L, L.vio_pg_optimize6dof(n, t.ctypes.data, R.ctypes.data, sq.ctypes.data, lt.ctypes.data, li.ctypes.data, to.ctypes.data, Ro.ctypes.data,
                                     dr.ctypes.data), "vio_pg_optimize6dof")
    return to, Ro.reshape(n, 3, 3), (dr[:9].reshape(3, 3).copy(), dr[9:].copy())


class KeyFrame:
    """KeyFrame (pose_graph/src/keyframe/keyframe.h): the online constructor computes the window descriptors and the FAST keypoints with their
    descriptors (keyframe.cpp:14-43), findConnection verifies a loop candidate chosen by the caller."""

    def __init__(self, cfg, pattern, time_stamp, index, vio_T_w_i, vio_R_w_i, image, point_3d, point_2d_uv, point_2d_norm, point_id, sequence=1):
        self.time_stamp, self.index, self.sequence = float(time_stamp), int(index), int(sequence)
        self.vio_T_w_i = np.array(vio_T_w_i, np.float64); self.vio_R_w_i = np.array(vio_R_w_i, np.float64).reshape(3, 3)
        self.T_w_i, self.R_w_i = self.vio_T_w_i.copy(), self.vio_R_w_i.copy()
        self.origin_vio_T, self.origin_vio_R = self.vio_T_w_i.copy(), self.vio_R_w_i.copy()
        self.point_3d = np.array(point_3d, np.float32).reshape(-1, 3)
        self.point_2d_uv = np.array(point_2d_uv, np.float32).reshape(-1, 2)
        self.point_2d_norm = np.array(point_2d_norm, np.float32).reshape(-1, 2)
        self.point_id = np.array(point_id, np.float64).reshape(-1)
        self.has_loop, self.loop_index, self.loop_info = False, -1, np.zeros(8)
        self.match_points = np.zeros((0, 3))
        self.window_brief_descriptors, kxy, self.brief_descriptors, kn = describe(cfg, image, self.point_2d_uv, pattern)
        self.keypoints, self.keypoints_norm = kxy, kn

    @classmethod
    def from_saved(cls, time_stamp, index, vio_T_w_i, vio_R_w_i, T_w_i, R_w_i, loop_index, loop_info, keypoints, keypoints_norm, brief_descriptors):
        """the "load previous keyframe" constructor (keyframe.cpp:46-78): the VIO pose is REPLACED by the loop-closed one, sequence 0, no window
        points (such a keyframe can be the old side of a loop only)"""
        kf = cls.__new__(cls)
        kf.time_stamp, kf.index, kf.sequence = float(time_stamp), int(index), 0
        kf.T_w_i, kf.R_w_i = np.array(T_w_i, np.float64), np.array(R_w_i, np.float64).reshape(3, 3)
        kf.vio_T_w_i, kf.vio_R_w_i = kf.T_w_i.copy(), kf.R_w_i.copy()
        kf.origin_vio_T, kf.origin_vio_R = kf.T_w_i.copy(), kf.R_w_i.copy()
        kf.point_3d, kf.point_2d_uv, kf.point_2d_norm, kf.point_id = np.zeros((0, 3), np.float32), np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32), np.zeros(0)
        kf.has_loop, kf.loop_index, kf.loop_info = int(loop_index) != -1, int(loop_index), np.array(loop_info, np.float64).reshape(8)
        kf.match_points = np.zeros((0, 3))
        kf.window_brief_descriptors = np.zeros((0, 4), np.uint64)
        kf.keypoints, kf.keypoints_norm = np.array(keypoints, np.float32).reshape(-1, 2), np.array(keypoints_norm, np.float32).reshape(-1, 2)
        kf.brief_descriptors = np.ascontiguousarray(brief_descriptors, np.uint64).reshape(-1, 4)
        return kf

    def findConnection(self, old_kf, qic, tic):
        idx, _ = match(self.window_brief_descriptors, old_kf.brief_descriptors)
        ok, info, mp, self.PnP_T_old, self.PnP_R_old = find_connection(self.point_3d, self.point_id, idx, old_kf.keypoints_norm, self.origin_vio_T,
                                                                       self.origin_vio_R, qic, tic)
        if ok:
            self.has_loop, self.loop_index, self.loop_info, self.match_points = True, old_kf.index, info, mp
        return ok


def _ypr2R(yaw_deg):
    y = np.deg2rad(yaw_deg)
    return np.array([[np.cos(y), -np.sin(y), 0.0], [np.sin(y), np.cos(y), 0.0], [0.0, 0.0, 1.0]])


def _yaw_deg(R):   # Utility::R2ypr(R).x()
    return float(np.rad2deg(np.arctan2(R[1, 0], R[0, 0])))


def _R2q_wxyz(R):   # Eigen Quaterniond(Matrix3d)
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        w = np.sqrt(t + 1.0) * 0.5
        f = 0.25 / w
        return np.array([w, (R[2, 1] - R[1, 2]) * f, (R[0, 2] - R[2, 0]) * f, (R[1, 0] - R[0, 1]) * f])
    i = int(np.argmax([R[0, 0], R[1, 1], R[2, 2]]))
    j, k = (i + 1) % 3, (i + 2) % 3
    q = np.zeros(4)
    r = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    q[1 + i] = 0.5 * r
    r = 0.5 / r
    q[0] = (R[k, j] - R[j, k]) * r
    q[1 + j] = (R[j, i] + R[i, j]) * r
    q[1 + k] = (R[k, i] + R[i, k]) * r
    return q


def _q2R_wxyz(q):   # Quaterniond::toRotationMatrix (no normalisation)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _bits_to_text(d):
    """boost::dynamic_bitset operator<<: the 256 bits of a descriptor, most significant (bit 255) first"""
    return "".join(format(int(d[w]), "064b") for w in (3, 2, 1, 0))


def _text_to_bits(txt):
    return np.array([int(txt[64 * (3 - w):64 * (4 - w)], 2) for w in range(4)], np.uint64)


class PoseGraph:
    """PoseGraph (pose_graph/src/pose_graph/pose_graph.{h,cpp}) for `imu: 1` without its threads, ROS messages and visualisation:
    addKeyFrame (:49-200) = the shift into the map frame, detectLoop, findConnection, the drift-corrected pose; optimize() = one pass of the
    optimize4DoF thread's loop body (:410-581), run by the caller instead of every two seconds.  The keyframes are this module's KeyFrame."""

    def __init__(self, vocabulary, qic, tic):
        self.voc, self.qic, self.tic = vocabulary, np.array(qic, np.float64).reshape(3, 3), np.array(tic, np.float64).reshape(3)
        self.keyframelist, self.optimize_buf = [], []
        self.global_index, self.sequence_cnt, self.sequence_loop, self.earliest_loop_index = 0, 0, [False], -1
        self.t_drift, self.r_drift, self.yaw_drift = np.zeros(3), np.eye(3), 0.0
        self.w_t_vio, self.w_r_vio = np.zeros(3), np.eye(3)

    def getKeyFrame(self, index):
        for kf in self.keyframelist:
            if kf.index == index:
                return kf
        return None

    def addKeyFrame(self, cur_kf, flag_detect_loop=True):
        """returns the index of the verified loop partner or -1"""
        if self.sequence_cnt != cur_kf.sequence:            # a new sequence starts in its own frame (:55-65)
            self.sequence_cnt += 1
            self.sequence_loop.append(False)
            self.w_t_vio, self.w_r_vio = np.zeros(3), np.eye(3)
            self.t_drift, self.r_drift = np.zeros(3), np.eye(3)
        cur_kf.vio_T_w_i = self.w_r_vio @ cur_kf.vio_T_w_i + self.w_t_vio
        cur_kf.vio_R_w_i = self.w_r_vio @ cur_kf.vio_R_w_i
        cur_kf.index = self.global_index
        self.global_index += 1
        loop_index = -1
        if flag_detect_loop:
            loop_index = self.voc.detectLoop(cur_kf.brief_descriptors, cur_kf.index)
        else:
            self.voc.add(cur_kf.brief_descriptors)             # addKeyFrameIntoVoc
        verified = -1
        if loop_index != -1:
            old_kf = self.getKeyFrame(loop_index)
            if old_kf is not None and cur_kf.findConnection(old_kf, self.qic, self.tic):
                verified = loop_index
                if self.earliest_loop_index > loop_index or self.earliest_loop_index == -1:
                    self.earliest_loop_index = loop_index
                if old_kf.sequence != cur_kf.sequence and not self.sequence_loop[cur_kf.sequence]:   # (:120-139) align the new sequence with the map
                    w_P_old, w_R_old = old_kf.vio_T_w_i, old_kf.vio_R_w_i     # old_kf->getVioPose (pose_graph.cpp:122), not the drift-corrected pose
                    rel_t, rel_q = cur_kf.loop_info[:3], cur_kf.loop_info[3:7]
                    w, x, y, z = rel_q
                    rel_R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
                    w_P_cur, w_R_cur = w_R_old @ rel_t + w_P_old, w_R_old @ rel_R
                    shift_r = _ypr2R(_yaw_deg(w_R_cur) - _yaw_deg(cur_kf.vio_R_w_i))
                    shift_t = w_P_cur - w_R_cur @ cur_kf.vio_R_w_i.T @ cur_kf.vio_T_w_i
                    self.w_r_vio, self.w_t_vio = shift_r, shift_t
                    cur_kf.vio_T_w_i = shift_r @ cur_kf.vio_T_w_i + shift_t
                    cur_kf.vio_R_w_i = shift_r @ cur_kf.vio_R_w_i
                    for kf in self.keyframelist:
                        if kf.sequence == cur_kf.sequence:
                            kf.vio_T_w_i, kf.vio_R_w_i = shift_r @ kf.vio_T_w_i + shift_t, shift_r @ kf.vio_R_w_i
                    self.sequence_loop[cur_kf.sequence] = True
                self.optimize_buf.append(cur_kf.index)
        cur_kf.T_w_i = self.r_drift @ cur_kf.vio_T_w_i + self.t_drift   # (:148-153)
        cur_kf.R_w_i = self.r_drift @ cur_kf.vio_R_w_i
        self.keyframelist.append(cur_kf)
        return verified

    def savePoseGraph(self, directory):
        """PoseGraph::savePoseGraph (pose_graph.cpp:849-927): pose_graph.txt (one line per keyframe: index, stamp, VIO t, loop-closed t, VIO q,
        loop-closed q (w x y z), loop_index, loop_info[8], number of keypoints; %f = six decimals like the reference), the two stamped
        trajectory files, and per keyframe <index>_briefdes.dat (one descriptor per line as 256 characters, bit 255 first) and
        <index>_keypoints.txt (pixel and normalised coordinates)"""
        os.makedirs(directory, exist_ok=True)
        with open(os.path.join(directory, "pose_graph.txt"), "w") as f, open(os.path.join(directory, "stamped_traj_estimate_mono_pg.txt"), "w") as fp, \
                open(os.path.join(directory, "stamped_traj_estimate_mono_vio.txt"), "w") as fv:
            for kf in self.keyframelist:
                vq, pq = _R2q_wxyz(kf.vio_R_w_i), _R2q_wxyz(kf.R_w_i)
                vals = [kf.time_stamp, *kf.vio_T_w_i, *kf.T_w_i, *vq, *pq]
                f.write(" %d " % kf.index + " ".join("%f" % v for v in vals) + " %d " % kf.loop_index + " ".join("%f" % v for v in kf.loop_info) +
                        " %d\n" % len(kf.keypoints))
                fp.write(" ".join("%f" % v for v in [kf.time_stamp, *kf.T_w_i, pq[1], pq[2], pq[3], pq[0]]) + "\n")
                fv.write(" ".join("%f" % v for v in [kf.time_stamp, *kf.vio_T_w_i, vq[1], vq[2], vq[3], vq[0]]) + "\n")
                with open(os.path.join(directory, "%d_briefdes.dat" % kf.index), "wb") as fb, open(os.path.join(directory, "%d_keypoints.txt" % kf.index), "w") as fk:
                    for d, xy, nn in zip(kf.brief_descriptors, kf.keypoints, kf.keypoints_norm):
                        fb.write((_bits_to_text(d) + "\n").encode())
                        fk.write("%f %f %f %f\n" % (xy[0], xy[1], nn[0], nn[1]))

    def loadPoseGraph(self, directory):
        """PoseGraph::loadPoseGraph (:929-1043) + loadKeyFrame(kf, 0) (:226-300): the saved keyframes become sequence 0 (held constant by the
        optimisation) with their loop-closed poses, and enter the vocabulary database; returns the number of keyframes read"""
        path = os.path.join(directory, "pose_graph.txt")
        if not os.path.exists(path):
            return 0
        n = 0
        for line in open(path):
            v = line.split()
            if len(v) != 26:
                continue
            index, stamp = int(v[0]), float(v[1])
            vio_T, pg_T = np.array(v[2:5], np.float64), np.array(v[5:8], np.float64)
            vio_q, pg_q = np.array(v[8:12], np.float64), np.array(v[12:16], np.float64)
            loop_index, loop_info, nkp = int(v[16]), np.array(v[17:25], np.float64), int(v[25])
            if loop_index != -1 and (self.earliest_loop_index > loop_index or self.earliest_loop_index == -1):
                self.earliest_loop_index = loop_index
            desc = [_text_to_bits(t) for t in open(os.path.join(directory, "%d_briefdes.dat" % index)).read().split()[:nkp]]
            kp = np.loadtxt(os.path.join(directory, "%d_keypoints.txt" % index), ndmin=2)[:nkp] if nkp else np.zeros((0, 4))
            kf = KeyFrame.from_saved(stamp, index, vio_T, _q2R_wxyz(vio_q), pg_T, _q2R_wxyz(pg_q), loop_index, loop_info, kp[:, 0:2], kp[:, 2:4],
                                     np.array(desc, np.uint64).reshape(-1, 4))
            kf.index = self.global_index                     # loadKeyFrame
            self.global_index += 1
            self.voc.add(kf.brief_descriptors)               # flag_detect_loop = 0: addKeyFrameIntoVoc
            self.keyframelist.append(kf)
            n += 1
        return n

    def optimize(self):
        """one pass of optimize4DoF's loop body over the keyframes earliest_loop_index .. newest queued one; False when nothing is queued"""
        if not self.optimize_buf:
            return False
        cur_index, first = self.optimize_buf[-1], self.earliest_loop_index
        self.optimize_buf = []
        nodes = [kf for kf in self.keyframelist if first <= kf.index <= cur_index]
        local = {kf.index: i for i, kf in enumerate(nodes)}
        loop_to = [local[kf.loop_index] if kf.has_loop else -1 for kf in nodes]
        t, R, (yaw, _) = optimize4DoF([kf.vio_T_w_i for kf in nodes], [kf.vio_R_w_i for kf in nodes], [kf.sequence for kf in nodes], loop_to,
                                      [kf.loop_info for kf in nodes])
        for kf, ti, Ri in zip(nodes, t, R):
            kf.T_w_i, kf.R_w_i = ti.copy(), Ri.copy()
        cur = nodes[-1]
        self.yaw_drift = _yaw_deg(cur.R_w_i) - _yaw_deg(cur.vio_R_w_i)          # (:547-553)
        self.r_drift = _ypr2R(self.yaw_drift)
        self.t_drift = cur.T_w_i - self.r_drift @ cur.vio_T_w_i
        for kf in self.keyframelist:
            if kf.index > cur_index:
                kf.T_w_i, kf.R_w_i = self.r_drift @ kf.vio_T_w_i + self.t_drift, self.r_drift @ kf.vio_R_w_i
        return True


def write_vocabulary(path, k, L, scoring, weighting, node_id, parent_id, weight, desc, word_node, word_id):
    """VINSLoop::Vocabulary::serialize (ThirdParty/VocabularyBinary.cpp): the file format PoseGraph::loadVocabulary reads"""
    node_id, parent_id = np.asarray(node_id, np.int32), np.asarray(parent_id, np.int32)
    nodes = np.zeros(len(node_id), np.dtype([("nodeId", "<i4"), ("parentId", "<i4"), ("weight", "<f8"), ("descriptor", "<u8", (4,))]))
    nodes["nodeId"], nodes["parentId"], nodes["weight"], nodes["descriptor"] = node_id, parent_id, weight, np.asarray(desc, np.uint64).reshape(-1, 4)
    words = np.zeros(len(word_node), np.dtype([("nodeId", "<i4"), ("wordId", "<i4")]))
    words["nodeId"], words["wordId"] = word_node, word_id
    with open(path, "wb") as f:
        f.write(np.asarray([k, L, scoring, weighting, len(nodes), len(words)], "<i4").tobytes())
        f.write(nodes.tobytes())
        f.write(words.tobytes())


class Vocabulary:
    """BriefVocabulary + BriefDatabase of PoseGraph (pose_graph.h:83-84): loadVocabulary, db.add, db.query, detectLoop."""

    def __init__(self, handle):
        self.P, self.L = _lib()
        if not handle:
            raise self.P.VioError("vocabulary: %s" % self.L.vio_last_error().decode())
        self.h = handle

    @classmethod
    def load(cls, path):
        P, L = _lib()
        return cls(L.vio_pg_voc_load(os.fsencode(path)))

    @classmethod
    def from_arrays(cls, k, L_, scoring, weighting, node_id, parent_id, weight, desc, word_node, word_id):
        P, L = _lib()
        a = [np.ascontiguousarray(node_id, np.int32), np.ascontiguousarray(parent_id, np.int32), np.ascontiguousarray(weight, np.float64),
             np.ascontiguousarray(desc, np.uint64), np.ascontiguousarray(word_node, np.int32), np.ascontiguousarray(word_id, np.int32)]
        return cls(L.vio_pg_voc_create(k, L_, scoring, weighting, len(a[0]), a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data,
                                       len(a[4]), a[4].ctypes.data, a[5].ctypes.data))

    def close(self):
        if self.h:
            self.L.vio_pg_voc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        o = np.zeros(7, np.int32)
        _chk(self.P, self.L, self.L.vio_pg_voc_info(self.h, o.ctypes.data), "vio_pg_voc_info")
        return dict(zip(("k", "L", "scoring", "weighting", "nodes", "words", "entries"), (int(x) for x in o)))

    @staticmethod
    def _desc(desc):
        d = np.ascontiguousarray(desc, np.uint64).reshape(-1, 4)
        return d, len(d)

    def transform(self, desc):
        """(word id, weight) of every descriptor: TemplatedVocabulary::transform(feature, ...)"""
        d, n = self._desc(desc)
        w, wt = np.zeros(n, np.int32), np.zeros(n)
        _chk(self.P, self.L, self.L.vio_pg_voc_transform(self.h, d.ctypes.data, n, w.ctypes.data, wt.ctypes.data), "vio_pg_voc_transform")
        return w, wt

    def bow(self, desc):
        """the L1-normalised bag-of-words vector: (ascending word ids, values)"""
        d, n = self._desc(desc)
        w, v = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1))
        m = _chk(self.P, self.L, self.L.vio_pg_voc_bow(self.h, d.ctypes.data, n, len(w), w.ctypes.data, v.ctypes.data), "vio_pg_voc_bow")
        return w[:m], v[:m]

    def add(self, desc):
        d, n = self._desc(desc)
        return _chk(self.P, self.L, self.L.vio_pg_db_add(self.h, d.ctypes.data, n), "vio_pg_db_add")

    def query(self, desc, max_results=4, max_id=-1):
        d, n = self._desc(desc)
        cap = max_results if max_results > 0 else max(self.info()["entries"], 1)
        ids, sc = np.zeros(cap, np.int32), np.zeros(cap)
        m = _chk(self.P, self.L, self.L.vio_pg_db_query(self.h, d.ctypes.data, n, max_results, max_id, ids.ctypes.data, sc.ctypes.data), "vio_pg_db_query")
        return ids[:m], sc[:m]

    def detectLoop(self, desc, frame_index, with_results=False):
        """PoseGraph::detectLoop: query, add, gates; the loop candidate's index or -1"""
        d, n = self._desc(desc)
        loop, ids, sc, m = C.c_int32(-1), np.zeros(4, np.int32), np.zeros(4), C.c_int32(0)
        _chk(self.P, self.L, self.L.vio_pg_detect_loop(self.h, d.ctypes.data, n, int(frame_index), C.addressof(loop), ids.ctypes.data, sc.ctypes.data,
                                                       C.addressof(m)), "vio_pg_detect_loop")
        return (loop.value, ids[:m.value], sc[:m.value]) if with_results else loop.value
