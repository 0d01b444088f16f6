"""Configuration and dataset I/O without ROS.

  * ``config_from_yaml``  the keys ``readParameters`` reads (vins_estimator/src/utility/parameters.cpp:81-243) from the
    reference's OpenCV-flavoured YAML (``%YAML:1.0`` header, ``!!opencv-matrix`` blocks) into a ``vio_config``;
  * ``OdometryCsvWriter``  the result file of ``pubOdometry`` (utility/visualization.cpp:214-225), byte for byte;
  * ``RgbdImuDirectory``   a rosbag-free recording: ``rgb.txt`` / ``depth.txt`` (TUM RGB-D association files: ``stamp path``)
    plus ``imu.txt`` (``stamp ax ay az gx gy gz``), 8-bit colour or grey PNG and 16-bit depth PNG in millimetres;
  * ``FrameGate``          the stream checks + frequency control at the top of ``process_tracker``
    (estimator_nodelet.cpp:94-95, 234-286: first image, discontinuity restart, ``frontend_freq`` skip, ``freq`` publish rate);
  * ``replay``             feeds such a recording through the C ABI exactly like the nodelet does for one camera
    (IMU up to the frame stamp + td, frame gate, then the colour + depth pair) and writes the CSV;
  * ``ate_rmse``           absolute trajectory error after yaw + translation alignment (gravity-aligned 4-DoF).

Host-side plumbing only: every frame still ends in ``libvio_hip.so``."""
import os
import re

import numpy as np


# --------------------------------------------------------------------------------------------------------------- YAML
def _scalar(tok):
    tok = tok.strip()
    if len(tok) >= 2 and tok[0] == tok[-1] and tok[0] in "\"'":
        return tok[1:-1]
    try:
        return int(tok)
    except ValueError:
        pass
    try:
        return float(tok)
    except ValueError:
        return tok


def _strip_comment(line):
    out, q = [], None
    for ch in line:
        if q:
            if ch == q:
                q = None
        elif ch in "\"'":
            q = ch
        elif ch == "#":
            break
        out.append(ch)
    return "".join(out).rstrip()


def parse_opencv_yaml(text):
    """Minimal reader for the cv::FileStorage YAML subset the reference's configs use: ``key: scalar``, one level of nested
    maps by indentation, flow sequences ``[a, b, ...]`` (possibly spanning lines) and ``!!opencv-matrix`` maps, which become
    numpy arrays of shape (rows, cols)."""
    lines = []
    for raw in text.splitlines():
        if raw.startswith("%YAML") or raw.strip() == "---":
            continue
        s = _strip_comment(raw)
        if s.strip():
            lines.append(s)
    # join flow sequences that span several lines
    joined, buf = [], None
    for s in lines:
        if buf is not None:
            buf += " " + s.strip()
            if buf.count("[") == buf.count("]"):
                joined.append(buf)
                buf = None
            continue
        if s.count("[") > s.count("]"):
            buf = s
        else:
            joined.append(s)
    if buf is not None:
        raise ValueError("unterminated flow sequence in YAML")
    root, stack = {}, [(-1, {})]
    stack[0] = (-1, root)
    pending_matrix = {}
    for s in joined:
        indent = len(s) - len(s.lstrip(" "))
        m = re.match(r"^\s*([A-Za-z_][\w\-]*)\s*:\s*(.*)$", s)
        if not m:
            raise ValueError("cannot parse YAML line: %r" % s)
        key, val = m.group(1), m.group(2).strip()
        while stack and indent <= stack[-1][0]:
            stack.pop()
        parent = stack[-1][1]
        if val == "" or val.startswith("!!"):
            node = {}
            if val.startswith("!!opencv-matrix"):
                pending_matrix[id(node)] = (parent, key)
            parent[key] = node
            stack.append((indent, node))
        elif val.startswith("["):
            parent[key] = [_scalar(x) for x in val.strip("[]").split(",") if x.strip()]
        else:
            parent[key] = _scalar(val)
    for nid, (parent, key) in pending_matrix.items():
        node = parent[key]
        parent[key] = np.array(node["data"], dtype=np.float64).reshape(int(node["rows"]), int(node["cols"]))
    return root


def config_from_yaml(path_or_text, P=None, strict=True):
    """vio_config from a reference configuration file (parameters.cpp:81-243).  Settings that select code paths outside the
    built hot path raise ValueError when strict (``estimate_extrinsic: 2``, a camera model other than PINHOLE); with strict=False they are returned in the second element as a list of notes."""
    if P is None:
        import importlib
        P = importlib.import_module("vins-rgbd-fast_amd")
    text = open(path_or_text).read() if os.path.exists(str(path_or_text)) else str(path_or_text)
    y = parse_opencv_yaml(text)
    c = P.default_config()
    notes = []

    def need(cond, msg):
        if cond:
            if strict:
                raise ValueError(msg)
            notes.append(msg)

    g = y.get
    c.width, c.height = int(g("image_width", c.width)), int(g("image_height", c.height))
    c.max_cnt, c.min_dist = int(g("max_cnt", c.max_cnt)), int(g("min_dist", c.min_dist))
    c.grid_rows, c.grid_cols = int(g("num_grid_rows", c.grid_rows)), int(g("num_grid_cols", c.grid_cols))
    c.f_threshold = float(g("F_threshold", c.f_threshold))
    c.depth_min, c.depth_max = float(g("depth_min_dist", c.depth_min)), float(g("depth_max_dist", c.depth_max))
    if "fix_depth" in y:
        c.fix_depth = int(y["fix_depth"])
    c.max_iterations = int(g("max_num_iterations", c.max_iterations))
    c.min_parallax_px = float(g("keyframe_parallax", c.min_parallax_px))
    for k in ("acc_n", "acc_w", "gyr_n", "gyr_w", "g_norm"):
        if k in y:
            setattr(c, k, float(y[k]))
    pp, dp = y.get("projection_parameters", {}), y.get("distortion_parameters", {})
    for k in ("fx", "fy", "cx", "cy"):
        if k in pp:
            setattr(c, k, float(pp[k]))
    for k in ("k1", "k2", "p1", "p2"):
        if k in dp:
            setattr(c, k, float(dp[k]))
    need(str(g("model_type", "PINHOLE")).upper() != "PINHOLE", "only the PINHOLE camera model is on the hot path")
    c.estimate_extrinsic = int(g("estimate_extrinsic", 0))
    need(c.estimate_extrinsic == 2, "estimate_extrinsic: 2 (online extrinsic initialisation) is out of scope")
    if "extrinsicRotation" in y:
        R = np.asarray(y["extrinsicRotation"], np.float64).reshape(3, 3)
        for i in range(9):
            c.ric[i] = float(R.ravel()[i])
    if "extrinsicTranslation" in y:
        T = np.asarray(y["extrinsicTranslation"], np.float64).ravel()
        for i in range(3):
            c.tic[i] = float(T[i])
    c.td = float(g("td", 0.0))
    c.estimate_td = int(g("estimate_td", 0))
    c.tr = float(g("rolling_shutter_tr", 0.0)) if int(g("rolling_shutter", 0)) else 0.0
    c.use_imu = 1 if int(g("imu", 1)) else 0
    if not c.use_imu:
        c.lk_max_level = 3   # calcOpticalFlowPyrLK(..., Size(21, 21), 3) without IMU prediction (feature_tracker.cpp:307-311)
        c.estimate_td = 0    # no td block without the IMU (estimator.cpp:1204)
    c.dynamic_init = 0 if int(g("static_init", 1)) else 1   # parameters.cpp:167: STATIC_INIT; 0 = SfM + visual-inertial alignment
    # FISHEYE (parameters.cpp:111-114): the mask file is <package>/config/fisheye_mask.jpg; decoding it is the caller's (no image codec here):
    # hand the decoded ROW x COL u8 image to VioBatch.set_fisheye_mask
    fisheye_mask = "config/fisheye_mask.jpg" if int(g("fisheye", 0)) == 1 else None
    if fisheye_mask is not None:
        notes.append("fisheye: 1 -- the mask image (%s, relative to the vins_estimator package) is NOT applied by config_from_yaml: decode it and call "
                     "VioBatch.set_fisheye_mask, otherwise the tracker runs unmasked and diverges from the reference" % fisheye_mask)
        import warnings
        warnings.warn(notes[-1], stacklevel=2)
    c.equalize = 1 if int(g("equalize", 0)) else 0   # parameters.cpp:110: CLAHE before tracking
    extra = dict(freq=int(g("freq", 0)), frontend_freq=int(g("frontend_freq", 0)), output_path=g("output_path", ""),
                 max_solver_time=float(g("max_solver_time", 0.0)), fisheye_mask=fisheye_mask, notes=notes)
    return c, extra


# ---------------------------------------------------------------------------------------------------------------- CSV
def format_odometry_row(stamp, P, Q_wxyz, V):
    """One line of VINS_RESULT_PATH (visualization.cpp:214-225): stamp in ns with precision 0, then P, Q(w,x,y,z), V with
    precision 5 (ios::fixed), comma separated with a trailing comma."""
    vals = list(P) + list(Q_wxyz) + list(V)
    return "%.0f," % (float(stamp) * 1e9) + ",".join("%.5f" % float(v) for v in vals) + ",\n"


class OdometryCsvWriter:
    def __init__(self, path, append=True):
        self.f = open(path, "a" if append else "w")

    def write(self, stamp, P, Q_wxyz, V):
        self.f.write(format_odometry_row(stamp, P, Q_wxyz, V))

    def write_rows(self, rows11):
        """rows of vio_get_odometry / vio_get_odometry_history: stamp, P(3), Q(w,x,y,z), V(3)"""
        for r in np.asarray(rows11, np.float64).reshape(-1, 11):
            self.write(r[0], r[1:4], r[4:8], r[8:11])

    def close(self):
        self.f.close()


def read_odometry_csv(path):
    rows = []
    for line in open(path):
        tok = [t for t in line.strip().split(",") if t != ""]
        if len(tok) >= 11:
            rows.append([float(t) for t in tok[:11]])
    a = np.array(rows, np.float64).reshape(-1, 11)
    a[:, 0] *= 1e-9
    return a


# ------------------------------------------------------------------------------------------------------------ dataset
def rgb_to_gray(img):
    """cv::cvtColor(RGB2GRAY) on 8-bit data: fixed point (R*4899 + G*9617 + B*1868 + 8192) >> 14 -- what cv_bridge's
    toCvCopy(.., MONO8) does to the colour frame in the nodelet (estimator_nodelet.cpp:277-290)."""
    a = np.asarray(img)
    if a.ndim == 2:
        return np.ascontiguousarray(a.astype(np.uint8))
    a = a[..., :3].astype(np.int32)
    return np.ascontiguousarray(((a[..., 0] * 4899 + a[..., 1] * 9617 + a[..., 2] * 1868 + 8192) >> 14).astype(np.uint8))


def _read_assoc(path):
    out = []
    for line in open(path):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        tok = line.replace(",", " ").split()
        out.append((float(tok[0]), tok[1]))
    return out


class ColorDepthSync:
    """The colour / depth pairing at the top of EstimatorNodelet::process_tracker (estimator_nodelet.cpp:200-232): two FIFO queues
    (img_buf, depth_buf); while both are non-empty compare the front stamps -- colour more than 3 ms OLDER than depth: pop colour
    ("throw color"); more than 3 ms NEWER: pop depth ("throw depth"); otherwise pop both as a pair.  `pop()` returns the next pair
    ``(colour_item, depth_item)`` or None when one queue ran dry (the nodelet waits on its condition variable there)."""
    TOLERANCE = 0.003

    def __init__(self):
        from collections import deque
        self.img_buf, self.depth_buf = deque(), deque()
        self.thrown_color = self.thrown_depth = 0

    def push_color(self, stamp, item):     # img_callback (estimator_nodelet.cpp:128-140)
        self.img_buf.append((float(stamp), item))

    def push_depth(self, stamp, item):     # depth_callback (:142-154)
        self.depth_buf.append((float(stamp), item))

    def pop(self):
        while self.img_buf and self.depth_buf:
            time_color, time_depth = self.img_buf[0][0], self.depth_buf[0][0]
            if time_color < time_depth - self.TOLERANCE:
                self.img_buf.popleft()
                self.thrown_color += 1
            elif time_color > time_depth + self.TOLERANCE:
                self.depth_buf.popleft()
                self.thrown_depth += 1
            else:
                return self.img_buf.popleft(), self.depth_buf.popleft()
        return None


def pair_color_depth(color_stamps, depth_stamps):
    """Index pairs (i, j) the nodelet's two-queue rule forms from two stamp lists in arrival order (offline use of ColorDepthSync)."""
    sync = ColorDepthSync()
    for i, t in enumerate(color_stamps):
        sync.push_color(t, i)
    for j, t in enumerate(depth_stamps):
        sync.push_depth(t, j)
    out = []
    while True:
        p = sync.pop()
        if p is None:
            return out
        out.append((p[0][1], p[1][1]))


class RgbdImuDirectory:
    """A rosbag-free recording: rgb.txt + depth.txt (``stamp relative/path.png``) and imu.txt (``stamp ax ay az gx gy gz``).
    Colour and depth frames are paired by the nodelet's own rule (ColorDepthSync: +-3 ms, two queues, estimator_nodelet.cpp:206-232);
    a frame's stamp is the colour stamp (:234 onwards uses time_color).  ``pairing="nearest"`` (not upstream) matches every colour
    frame with the nearest depth frame within ``max_dt`` instead -- for datasets whose streams are not hardware-synchronised."""

    def __init__(self, root, max_dt=0.02, *, pairing="nodelet"):
        # (max_dt keeps the position it had before the pairing rule became selectable; pairing is keyword-only)
        self.root = root
        rgb, dep = _read_assoc(os.path.join(root, "rgb.txt")), _read_assoc(os.path.join(root, "depth.txt"))
        self.pairs = []
        self.thrown_color = self.thrown_depth = 0
        if pairing == "nodelet":
            for i, j in pair_color_depth([t for t, _ in rgb], [t for t, _ in dep]):
                self.pairs.append((rgb[i][0], rgb[i][1], dep[j][1]))
            self.thrown_color, self.thrown_depth = len(rgb) - len(self.pairs), len(dep) - len(self.pairs)
            if rgb and dep and len(self.pairs) < 0.9 * min(len(rgb), len(dep)):
                # the +-3 ms rule assumes hardware-synchronised streams (RealSense); TUM-style recordings are not
                import warnings
                warnings.warn("RgbdImuDirectory(%r): the nodelet's +-3 ms colour / depth pairing kept %d of %d colour and %d depth frames "
                              "(%d / %d thrown); pass pairing='nearest' for streams that are not hardware-synchronised"
                              % (root, len(self.pairs), len(rgb), len(dep), self.thrown_color, self.thrown_depth), RuntimeWarning, stacklevel=2)
        elif pairing == "nearest":
            dt = np.array([t for t, _ in dep])
            for t, f in rgb:
                if len(dt) == 0:
                    break
                k = int(np.argmin(np.abs(dt - t)))
                if abs(dt[k] - t) <= max_dt:
                    self.pairs.append((t, f, dep[k][1]))
        else:
            raise ValueError("pairing must be 'nodelet' or 'nearest'")
        imu = np.loadtxt(os.path.join(root, "imu.txt"), comments="#", ndmin=2)
        if imu.shape[1] < 7:
            raise ValueError("imu.txt needs 7 columns: stamp ax ay az gx gy gz")
        self.imu_t, self.imu_acc, self.imu_gyr = imu[:, 0].copy(), np.ascontiguousarray(imu[:, 1:4]), np.ascontiguousarray(imu[:, 4:7])

    def __len__(self):
        return len(self.pairs)

    def frame(self, k):
        from PIL import Image
        t, fc, fd = self.pairs[k]
        gray = rgb_to_gray(np.array(Image.open(os.path.join(self.root, fc))))
        depth = np.array(Image.open(os.path.join(self.root, fd)))
        if depth.dtype != np.uint16:
            depth = depth.astype(np.uint16)
        return t, gray, np.ascontiguousarray(depth)


def write_recording(root, stamps, grays, depths, imu_t, imu_acc, imu_gyr, depth_stamps=None):
    """Inverse of RgbdImuDirectory (used by the tests and handy for exporting synthetic sequences).  depth_stamps: stamps of the depth
    frames when they differ from the colour stamps (same count)."""
    from PIL import Image
    os.makedirs(os.path.join(root, "rgb"), exist_ok=True)
    os.makedirs(os.path.join(root, "depth"), exist_ok=True)
    with open(os.path.join(root, "rgb.txt"), "w") as fr, open(os.path.join(root, "depth.txt"), "w") as fd:
        fr.write("# stamp filename\n")
        fd.write("# stamp filename\n")
        for k, (t, g, d) in enumerate(zip(stamps, grays, depths)):
            name = "%.6f.png" % t
            Image.fromarray(np.asarray(g, np.uint8)).save(os.path.join(root, "rgb", name))
            Image.fromarray(np.asarray(d, np.uint16)).save(os.path.join(root, "depth", name))
            fr.write("%.9f rgb/%s\n" % (t, name))
            fd.write("%.9f depth/%s\n" % (t if depth_stamps is None else depth_stamps[k], name))
    np.savetxt(os.path.join(root, "imu.txt"), np.c_[imu_t, imu_acc, imu_gyr], fmt="%.17g", header="stamp ax ay az gx gy gz")


class FrameGate:
    """Stream checks + frequency control of EstimatorNodelet::process_tracker (estimator_nodelet.cpp:94-95, 234-286) as a
    stand-alone state machine.  ``step(t)`` returns what the nodelet does with the frame stamped ``t``: FIRST (only sets the
    time base), RESET (stream discontinuity: restart the estimator, :243-262), SKIP ("Skip this frame", before
    readImage), TRACK (readImage with PUB_THIS_FRAME false) or PUBLISH.  The first three values equal the VIO_FRAME_* modes."""
    SKIP, TRACK, PUBLISH, FIRST, RESET = 0, 1, 2, 3, 4

    def __init__(self, freq, frontend_freq):
        self.freq = 100 if int(freq) == 0 else int(freq)     # parameters.cpp:133-134
        self.frontend_freq = int(frontend_freq)
        self.first_image_flag = True
        self.first_image_time = self.last_image_time = 0.0
        self.pub_count, self.input_count = 1, 0               # estimator_nodelet.cpp:94-95

    @staticmethod
    def _round(x):
        return float(np.floor(abs(x) + 0.5)) * (1.0 if x >= 0 else -1.0)   # C round(): half away from zero

    def step(self, t):
        t = float(t)
        if self.first_image_flag:
            self.first_image_flag = False
            self.first_image_time = self.last_image_time = t
            return self.FIRST
        if t - self.last_image_time > 1.0 or t < self.last_image_time:
            self.first_image_flag = True
            self.last_image_time = 0.0
            self.pub_count = 1
            return self.RESET
        span = t - self.first_image_time
        if self._round(np.float64(self.input_count) / span if span != 0 else np.inf) > self.frontend_freq:
            return self.SKIP
        self.input_count += 1
        pub = False
        rate = np.float64(self.pub_count) / span if span != 0 else np.inf
        if self._round(rate) <= self.freq:
            pub = True
            if abs(rate - self.freq) < 0.01 * self.freq:
                self.first_image_time = t
                self.pub_count = 0
                self.input_count = 0
        self.last_image_time = t
        if pub:
            self.pub_count += 1
        return self.PUBLISH if pub else self.TRACK

    def empty_map(self, t):
        """a PUBLISH frame whose feature map came out empty restarts the rate window (estimator_nodelet.cpp:386-392)"""
        self.first_image_time = float(t)
        self.pub_count = self.input_count = 0


def replay(batch, rec, csv_path=None, seq=0, on_frame=None, freq=0, frontend_freq=0):
    """Feed a recording through a single-sequence slot of a VioBatch the way the nodelet does: push IMU through the frame stamp
    (one sample beyond, so that IMUAvailable holds), run the frame gate (``freq`` / ``frontend_freq`` of the configuration file;
    frontend_freq == 0 disables the gate: every frame is published), feed the pair with the gate's mode, append a CSV row whenever
    the estimator is NON_LINEAR.  A stream discontinuity restarts the estimator of the sequence (vio_reset_seq; the tracker keeps its state).  Returns the rows
    [stamp, P, Q(wxyz), V]."""
    rows, k = [], 0
    wr = OdometryCsvWriter(csv_path, append=False) if csv_path else None
    S = batch.S
    gate = FrameGate(freq, frontend_freq) if int(frontend_freq) > 0 else None
    init_pub = init_feature = False   # host mirror of estimator_nodelet.cpp:365-377, only to recognise an EMPTY published map
    for f in range(len(rec)):
        t, gray, depth = rec.frame(f)
        k2 = k
        while k2 < len(rec.imu_t) and rec.imu_t[k2] <= t + 1e-9:
            k2 += 1
        k2 = min(len(rec.imu_t), k2 + 1)
        if k2 > k:
            batch.push_imu(seq, rec.imu_t[k:k2], rec.imu_acc[k:k2], rec.imu_gyr[k:k2])
            k = k2
        mode = FrameGate.PUBLISH
        if gate is not None:
            d = gate.step(t)
            if d == FrameGate.RESET:
                # estimator_nodelet.cpp:243-262: feature_buf emptied, estimator.clearState() + setParameter(); the tracker (points, ids,
                # previous image) and init_pub / init_feature are NOT reset upstream, and are not here
                batch.reset_seq(seq)
                if on_frame:
                    on_frame(f, batch.status(seq))
                continue
            mode = FrameGate.PUBLISH if d == FrameGate.FIRST else d   # the first image is recognised on the device as well
        g = np.repeat(gray[None], S, 0) if S > 1 else gray[None]
        d_ = np.repeat(depth[None], S, 0) if S > 1 else depth[None]
        modes = np.full(S, FrameGate.SKIP, np.uint8)
        modes[seq] = mode
        batch.feed(g, d_, [t] * S, modes=modes)
        st = batch.status(seq)
        if gate is not None and mode == FrameGate.PUBLISH and d != FrameGate.FIRST:
            if not init_pub:
                init_pub = True
            elif not init_feature:
                init_feature = True
            elif not st.processed and st.code == 0:
                gate.empty_map(t)
        if st.solver_flag == 1 and st.processed:
            row = batch.odometry()[seq]
            rows.append(row.copy())
            if wr:
                wr.write(row[0], row[1:4], row[4:8], row[8:11])
        if on_frame:
            on_frame(f, st)
    if wr:
        wr.close()
    return np.array(rows).reshape(-1, 11)


# ---------------------------------------------------------------------------------------------------------------- ATE
def yaw_align(est, gt):
    est, gt = np.asarray(est, np.float64), np.asarray(gt, np.float64)
    ec, gc = est - est.mean(0), gt - gt.mean(0)
    th = np.arctan2((ec[:, 0] * gc[:, 1] - ec[:, 1] * gc[:, 0]).sum(), (ec[:, 0] * gc[:, 0] + ec[:, 1] * gc[:, 1]).sum())
    c, s = np.cos(th), np.sin(th)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    return (R @ ec.T).T + gt.mean(0)


def ate_rmse(est, gt):
    al = yaw_align(est, gt)
    return float(np.sqrt(((al - np.asarray(gt)) ** 2).sum(1).mean()))
