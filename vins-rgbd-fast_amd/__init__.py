"""MI355X-native VIO hot path — Python host side (ctypes over the C ABI in include/vio_abi.h).

The package directory name contains hyphens, so import it with
``importlib.import_module("vins-rgbd-fast_amd")`` (``__graft_entry__.load_package()`` does that).

Nothing here computes on the CPU: every call ends in ``libvio_hip.so`` and fails with ``VioError`` when the
extension is missing or no GPU is present.  PyTorch is optional and only used as a source of device pointers.

Mirrors of the reference interface (same names / argument meaning):
  FeatureTracker.readImage / updateID      vins_estimator/src/feature_tracker/feature_tracker.h:36-47
  Estimator.inputIMU / processImage / ...   vins_estimator/src/estimator/estimator.h:29-60
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvio_hip.so")

VIO_OK, VIO_NEED_IMU, VIO_REBOOTED = 0, 1, 2
VIO_EINVAL, VIO_EDEVICE, VIO_ECAPACITY = -1, -2, -3


class VioError(RuntimeError):
    pass


class Config(C.Structure):
    """vio_config (include/vio_abi.h); same field order as oracle ovio::Config."""
    _fields_ = [(n, C.c_int32) for n in (
        "width", "height", "max_cnt", "min_dist", "grid_rows", "grid_cols", "window_size", "max_landmarks", "fix_depth",
        "estimate_extrinsic", "estimate_td", "max_iterations", "ransac_max_iters", "lk_max_level", "dynamic_init", "use_imu", "reference_quirks",
        "marg_exact", "equalize")] + \
        [(n, C.c_double) for n in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2", "focal_length", "f_threshold", "depth_min",
                                   "depth_max", "acc_n", "acc_w", "gyr_n", "gyr_w", "g_norm")] + \
        [("ric", C.c_double * 9), ("tic", C.c_double * 3)] + \
        [(n, C.c_double) for n in ("td", "tr", "min_parallax_px", "init_depth")]


class SynthConfig(C.Structure):
    """vio_synth_config (include/vio_synth.h)."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32)] + \
        [(n, C.c_double) for n in ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2")] + \
        [("ric", C.c_double * 9), ("tic", C.c_double * 3)] + \
        [(n, C.c_double) for n in ("g_norm", "imu_rate", "cam_rate", "t_static", "acc_noise", "gyr_noise", "acc_bias_walk",
                                   "gyr_bias_walk")] + [("seed", C.c_uint64)]


class Status(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "code", "solver_flag", "frame_count", "marginalization_flag", "n_landmarks", "last_track_num", "n_tracks", "processed",
        "iterations", "successful_steps", "n_in_problem", "n_residuals", "n_var_landmarks", "has_prior", "reboot_count",
        "frames_processed")] + [(n, C.c_double) for n in ("initial_cost", "final_cost", "td")] + \
        [(n, C.c_int32) for n in ("overflow_flags", "overflow_frames", "iterations_total", "solves_total")]


FRAME_SKIP, FRAME_TRACK, FRAME_PUBLISH = 0, 1, 2


def build(verbose=False):
    """Compile libvio_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-j8", "-C", os.path.join(_HERE, "csrc")], capture_output=True, text=True)
    if r.returncode != 0:
        raise VioError("hipcc build failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout[-2000:])
    return _LIB_PATH


_lib = None


def lib():
    """Load libvio_hip.so (no CPU fallback: raises if the extension is missing)."""
    global _lib
    if _lib is None:
        path = _LIB_PATH
        if os.environ.get("VIO_HIP_LIB") == "timers":
            # the profiling tools' build with the in-kernel phase timers compiled in (csrc/Makefile target `timers`); never the default
            path = os.path.join(_HERE, "libvio_hip_timers.so")
            if not os.path.exists(path):
                r = subprocess.run(["make", "-j8", "-C", os.path.join(_HERE, "csrc"), "timers"], capture_output=True, text=True)
                if r.returncode != 0:
                    raise VioError("hipcc build (timers) failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
        if not os.path.exists(path):
            raise VioError("libvio_hip.so is not built: run __graft_entry__.build() (the product path has no CPU fallback)")
        L = C.CDLL(path)
        L.vio_create.restype = C.c_void_p
        L.vio_create.argtypes = [C.POINTER(Config), C.c_int, C.c_int]
        L.vio_create_on_device.restype = C.c_void_p
        L.vio_create_on_device.argtypes = [C.POINTER(Config), C.c_int, C.c_int, C.c_int]
        L.vio_get_device.argtypes = [C.c_void_p]
        L.vio_host_buffers_done.argtypes = [C.c_void_p, C.c_int]
        L.vio_destroy.argtypes = [C.c_void_p]
        L.vio_last_error.restype = C.c_char_p
        L.vio_reset.argtypes = [C.c_void_p]
        L.vio_push_imu.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vio_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.vio_track.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.vio_process.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.vio_sync.argtypes = [C.c_void_p]
        L.vio_reset_seq.argtypes = [C.c_void_p, C.c_int]
        L.vio_reset_tracker_seq.argtypes = [C.c_void_p, C.c_int]
        L.vio_set_relo_frame.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vio_get_relo.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.vio_push_imu_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vio_feed_modes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.vio_track_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.vio_predict_motion.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p]
        L.vio_process_obs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        L.vio_process_obs_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.vio_get_packaged.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.vio_get_capacity.argtypes = [C.c_void_p, C.c_void_p]
        L.vio_get_landmarks_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.vio_get_odometry_history.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.vio_get_stream.restype = C.c_void_p
        L.vio_get_stream.argtypes = [C.c_void_p]
        L.vio_get_status.argtypes = [C.c_void_p, C.c_int, C.POINTER(Status)]
        L.vio_get_window.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.vio_get_odometry.argtypes = [C.c_void_p, C.c_void_p]
        L.vio_get_extrinsic.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.vio_get_latest_odometry.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.vio_set_tracker_lag.argtypes = [C.c_void_p, C.c_int]
        L.vio_set_fisheye_mask.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.vio_get_status_all.argtypes = [C.c_void_p, C.c_void_p]
        L.vio_get_tracks.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
        L.vio_get_landmarks.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.vio_get_prior.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4
        L.vio_get_timings.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.vio_profile_begin.argtypes = [C.c_void_p, C.c_int]
        L.vio_profile_end.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.vio_synth_pose.argtypes = [C.POINTER(SynthConfig), C.c_uint64, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vio_synth_imu.argtypes = [C.POINTER(SynthConfig), C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vio_synth_render_host.argtypes = [C.POINTER(SynthConfig), C.c_uint64, C.c_double, C.c_void_p, C.c_void_p]
        L.vio_synth_render_device.argtypes = [C.POINTER(SynthConfig), C.c_int, C.c_uint64, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vio_stage_pyr_down.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.vio_stage_clahe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.vio_stage_fast_roi.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.c_void_p]
        L.vio_stage_lk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vio_stage_ransac.argtypes = [C.POINTER(Config), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vio_stage_imu_factor.argtypes = [C.POINTER(Config), C.c_int] + [C.c_void_p] * 14
        L.vio_stage_projection.argtypes = [C.POINTER(Config)] + [C.c_void_p] * 3 + [C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                                                                  C.c_int, C.c_void_p, C.c_void_p]
        L.vio_device_alloc.restype = C.c_void_p
        L.vio_device_alloc.argtypes = [C.c_size_t]
        L.vio_device_free.argtypes = [C.c_void_p]
        L.vio_device_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.vio_device_download.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.vio_stage_projection_residual.argtypes = L.vio_stage_projection.argtypes
        L.vio_stage_pnp.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.vio_stage_imu_block.argtypes = [C.POINTER(Config), C.c_int] + [C.c_void_p] * 12
        L.vio_stage_chol.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5
        _lib = L
    return _lib


def default_config(**kw):
    c = Config()
    lib().vio_config_default(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def canonical_config(**kw):
    """BASELINE config 2/3: 640x480, 150 features, 5x6 grid, W = 10; landmarks are optimisation variables
    (fix_depth 0, depth range as in config/realsense/vio_campus.yaml which is the upstream 150-feature setting)."""
    d = dict(fix_depth=0, depth_max=10.0)
    d.update(kw)
    return default_config(**d)


def default_synth(**kw):
    c = SynthConfig()
    lib().vio_synth_config_default(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    if hasattr(a, "data_ptr"):  # torch tensor (device or host)
        return a.data_ptr()
    return a


class PinnedArray:
    """numpy array over page-locked host memory (vio_host_alloc): hand `.a` to feed / track / process with on_device=False"""

    def __init__(self, shape, dtype):
        self.L = lib()
        self.L.vio_host_alloc.restype = C.c_void_p
        self.L.vio_host_alloc.argtypes = [C.c_size_t]
        self.L.vio_host_free.argtypes = [C.c_void_p]
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = self.L.vio_host_alloc(n)
        if not self.ptr:
            raise VioError("vio_host_alloc(%d) failed" % n)
        self.a = np.frombuffer((C.c_uint8 * n).from_address(self.ptr), dtype=dtype).reshape(shape)

    def free(self):
        if self.ptr:
            self.a = None
            self.L.vio_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBuffer:
    """A plain HBM allocation (vio_device_alloc) for the on_device = True paths when the caller has no HIP binding of its own
    (bench.py uses torch tensors instead).  `ptr` is the device address; slices are addressed with `at(byte_offset)`."""

    def __init__(self, nbytes):
        self.L = lib()
        self.nbytes = int(nbytes)
        self.ptr = self.L.vio_device_alloc(self.nbytes)
        if not self.ptr:
            raise VioError("vio_device_alloc(%d) failed: %s" % (self.nbytes, self.L.vio_last_error().decode()))

    def at(self, byte_offset):
        return self.ptr + int(byte_offset)

    def download(self, byte_offset, shape, dtype):
        out = np.empty(shape, dtype)
        if self.L.vio_device_download(out.ctypes.data, self.ptr + int(byte_offset), out.nbytes) != 0:
            raise VioError("vio_device_download failed")
        return out

    def upload(self, byte_offset, arr):
        arr = np.ascontiguousarray(arr)
        if self.L.vio_device_upload(self.ptr + int(byte_offset), arr.ctypes.data, arr.nbytes) != 0:
            raise VioError("vio_device_upload failed")

    def free(self):
        if self.ptr:
            self.L.vio_device_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Synth:
    """Synthetic RGB-D + IMU workload."""

    def __init__(self, cfg=None):
        self.cfg = cfg or default_synth()
        self.L = lib()

    def pose(self, seq, t):
        p, R, v = np.zeros(3), np.zeros(9), np.zeros(3)
        self.L.vio_synth_pose(C.byref(self.cfg), seq, t, p.ctypes.data, R.ctypes.data, v.ctypes.data)
        return p, R.reshape(3, 3), v

    def imu(self, seq, n):
        t, a, g = np.zeros(n), np.zeros((n, 3)), np.zeros((n, 3))
        self.L.vio_synth_imu(C.byref(self.cfg), seq, n, t.ctypes.data, a.ctypes.data, g.ctypes.data)
        return t, a, g

    def render_host(self, seq, t):
        g = np.zeros((self.cfg.height, self.cfg.width), np.uint8)
        d = np.zeros((self.cfg.height, self.cfg.width), np.uint16)
        self.L.vio_synth_render_host(C.byref(self.cfg), seq, t, g.ctypes.data, d.ctypes.data)
        return g, d

    def render_device(self, n_seq, seq0, t, d_gray, d_depth, stream=None):
        rc = self.L.vio_synth_render_device(C.byref(self.cfg), n_seq, seq0, t, _ptr(d_gray), _ptr(d_depth), stream)
        if rc != 0:
            raise VioError("vio_synth_render_device failed (%d)" % rc)


class VioBatch:
    """A batch of S independent sequences resident in HBM (vio_batch)."""

    def __init__(self, cfg=None, n_seq=1, imu_capacity=8192, device=None):
        """device: HIP device index the handle lives on (None = the calling thread's current device); the handle keeps it and every call
        binds to it (vio_create_on_device), so handles on different GPUs can be driven from one thread or from one thread each."""
        self.L = lib()
        self.cfg = cfg or default_config()
        self.S = n_seq
        self.W = self.cfg.window_size
        self.h = self.L.vio_create_on_device(C.byref(self.cfg), n_seq, imu_capacity, -1 if device is None else int(device))
        if not self.h:
            raise VioError("vio_create failed: %s" % self.L.vio_last_error().decode())
        self.h = C.c_void_p(self.h)

    @property
    def device(self):
        return int(self.L.vio_get_device(self.h))

    def host_buffers_done(self, calls_ago=0):
        """vio_host_buffers_done: have the page-locked image buffers of the feed issued `calls_ago` feeds ago been uploaded?"""
        return self._chk(self.L.vio_host_buffers_done(self.h, int(calls_ago)), "vio_host_buffers_done") == 1

    def close(self):
        if self.h:
            self.L.vio_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc < 0:
            raise VioError("%s failed (%d): %s" % (what, rc, self.L.vio_last_error().decode()))
        return rc

    def push_imu(self, seq, t, acc, gyr):
        t = np.ascontiguousarray(t, np.float64).reshape(-1)
        acc = np.ascontiguousarray(acc, np.float64).reshape(-1, 3)
        gyr = np.ascontiguousarray(gyr, np.float64).reshape(-1, 3)
        self._chk(self.L.vio_push_imu(self.h, seq, len(t), t.ctypes.data, acc.ctypes.data, gyr.ctypes.data), "vio_push_imu")

    def push_imu_batch(self, t, acc, gyr, n=None):
        """vio_push_imu_batch: t [S][stride], acc / gyr [S][stride][3]; n = optional per-sequence counts."""
        t = np.ascontiguousarray(t, np.float64).reshape(self.S, -1)
        stride = t.shape[1]
        acc = np.ascontiguousarray(acc, np.float64).reshape(self.S, stride, 3)
        gyr = np.ascontiguousarray(gyr, np.float64).reshape(self.S, stride, 3)
        nn = None if n is None else np.ascontiguousarray(n, np.int32).reshape(self.S)
        self._chk(self.L.vio_push_imu_batch(self.h, None if nn is None else nn.ctypes.data, stride, t.ctypes.data, acc.ctypes.data,
                                            gyr.ctypes.data), "vio_push_imu_batch")

    def feed(self, gray, depth, stamps, on_device=False, modes=None):
        stamps = np.ascontiguousarray(stamps, np.float64).reshape(-1)
        assert len(stamps) == self.S
        m = None if modes is None else np.ascontiguousarray(modes, np.uint8).reshape(self.S)
        self._chk(self.L.vio_feed_modes(self.h, _ptr(gray), _ptr(depth), stamps.ctypes.data, None if m is None else m.ctypes.data,
                                        1 if on_device else 0), "vio_feed")
        self._keep = (gray, depth, stamps, m)   # host buffers stay alive until the next feed has returned (vio_abi.h)

    def track(self, gray, stamps, publish=True, on_device=False, modes=None, R_rel=None):
        """vio_track / vio_track_ex: modes = per-sequence FRAME_* (default: publish for all), R_rel = caller-supplied relative
        rotations [S][3][3] (rows of NaN = predict on the device)."""
        stamps = np.ascontiguousarray(stamps, np.float64).reshape(-1)
        if modes is None and R_rel is None:
            self._keep = (gray, stamps)
            self._chk(self.L.vio_track(self.h, _ptr(gray), stamps.ctypes.data, 1 if publish else 0, 1 if on_device else 0), "vio_track")
            return
        m = np.full(self.S, FRAME_PUBLISH if publish else FRAME_TRACK, np.uint8) if modes is None else \
            np.ascontiguousarray(modes, np.uint8).reshape(self.S)
        R = None if R_rel is None else np.ascontiguousarray(R_rel, np.float64).reshape(self.S, 9)
        self._keep = (gray, stamps, m, R)
        self._chk(self.L.vio_track_ex(self.h, _ptr(gray), stamps.ctypes.data, m.ctypes.data, None if R is None else R.ctypes.data,
                                      1 if on_device else 0), "vio_track_ex")

    def predict_motion(self, seq, t0, t1):
        R = np.zeros(9)
        self._chk(self.L.vio_predict_motion(self.h, seq, float(t0), float(t1), R.ctypes.data), "vio_predict_motion")
        return R.reshape(3, 3)

    def process(self, depth, on_device=False):
        self._keep2 = depth
        self._chk(self.L.vio_process(self.h, _ptr(depth), 1 if on_device else 0), "vio_process")

    def process_obs(self, seq, ids, obs, depth, stamp):
        """Estimator::processImage(image, header) with a caller-supplied feature map (ids ascending, obs [n][7])."""
        ids = np.ascontiguousarray(ids, np.int32).reshape(-1)
        obs = np.ascontiguousarray(obs, np.float64).reshape(-1, 7)
        depth = np.ascontiguousarray(depth, np.uint16)
        self._keep3 = (ids, obs, depth)
        self._chk(self.L.vio_process_obs(self.h, seq, len(ids), ids.ctypes.data, obs.ctypes.data, depth.ctypes.data, float(stamp)),
                  "vio_process_obs")

    def process_obs_batch(self, n_obs, ids, obs, depth, stamps, on_device=False):
        n_obs = np.ascontiguousarray(n_obs, np.int32).reshape(self.S)
        ids = np.ascontiguousarray(ids, np.int32).reshape(self.S, -1)
        cap = ids.shape[1]
        obs = np.ascontiguousarray(obs, np.float64).reshape(self.S, cap, 7)
        stamps = np.ascontiguousarray(stamps, np.float64).reshape(self.S)
        self._keep3 = (n_obs, ids, obs, depth, stamps)
        self._chk(self.L.vio_process_obs_batch(self.h, n_obs.ctypes.data, ids.ctypes.data, obs.ctypes.data, cap, _ptr(depth),
                                               stamps.ctypes.data, 1 if on_device else 0), "vio_process_obs_batch")

    def packaged(self, seq=0, cap=2048):
        """The feature map packaged by the last track / feed (what the nodelet pushes to feature_buf): ids, obs [n][7]."""
        ids, obs = np.zeros(cap, np.int32), np.zeros((cap, 7))
        n = self._chk(self.L.vio_get_packaged(self.h, seq, cap, ids.ctypes.data, obs.ctypes.data), "vio_get_packaged")
        return ids[:n].copy(), obs[:n].copy()

    def solver_kind(self):
        """0 persistent fallback kernel, 1 phased solver (LDS-resident Schur complement), 2 phased solver (HBM-resident, large windows)"""
        self.L.vio_get_solver_kind.argtypes = [C.c_void_p]
        return self.L.vio_get_solver_kind(self.h)

    def bound_stats(self, seq=0):
        """(inverse depths cut by the upper bound, bounded landmarks that entered solves, line-search trial evaluations, shortened steps) of
        sequence seq since creation (estimator.cpp:1282-1297; Ceres' projected Armijo line search on the bounds-constrained program)"""
        o = np.zeros(4, np.int64)
        self.L.vio_get_bound_stats.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self._chk(self.L.vio_get_bound_stats(self.h, int(seq), o.ctypes.data), "vio_get_bound_stats")
        return tuple(int(x) for x in o)

    def marg_certificate(self, seq=0):
        """marg_exact = 2: (marginalisations whose certificate failed since creation, last one certified?)"""
        o = np.zeros(2, np.int32)
        self.L.vio_get_marg_certificate.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self._chk(self.L.vio_get_marg_certificate(self.h, int(seq), o.ctypes.data), "vio_get_marg_certificate")
        return int(o[0]), bool(o[1])

    def capacity(self):
        c = np.zeros(3, np.int32)
        self._chk(self.L.vio_get_capacity(self.h, c.ctypes.data), "vio_get_capacity")
        return dict(tracks=int(c[0]), landmarks=int(c[1]), imu=int(c[2]))

    def reset_seq(self, seq):
        """Estimator::clearState() + setParameter() for one sequence (the nodelet's stream-discontinuity restart: the tracker is kept)"""
        self._chk(self.L.vio_reset_seq(self.h, seq), "vio_reset_seq")

    def reset_tracker_seq(self, seq):
        self._chk(self.L.vio_reset_tracker_seq(self.h, seq), "vio_reset_tracker_seq")

    def sync(self):
        self._chk(self.L.vio_sync(self.h), "vio_sync")

    def reset(self):
        self._chk(self.L.vio_reset(self.h), "vio_reset")

    def stream(self):
        return self.L.vio_get_stream(self.h)

    def status(self, seq=0):
        s = Status()
        self._chk(self.L.vio_get_status(self.h, seq, C.byref(s)), "vio_get_status")
        return s

    def status_all(self):
        """vio_status of every sequence (two device reads for the whole batch)"""
        arr = (Status * self.S)()
        self._chk(self.L.vio_get_status_all(self.h, arr), "vio_get_status_all")
        return list(arr)

    def window(self, seq=0):
        w = np.zeros((self.W + 1, 17))
        self._chk(self.L.vio_get_window(self.h, seq, w.ctypes.data), "vio_get_window")
        return w

    def odometry(self):
        o = np.zeros((self.S, 11))
        self._chk(self.L.vio_get_odometry(self.h, o.ctypes.data), "vio_get_odometry")
        return o

    def odometry_history(self, seq=0, cap=2048):
        o = np.zeros((cap, 11))
        n = self._chk(self.L.vio_get_odometry_history(self.h, seq, cap, o.ctypes.data), "vio_get_odometry_history")
        return o[:min(n, cap)]

    def set_tracker_lag(self, lag):
        """0: the tracker of frame f+1 waits for the optimisation of frame f; 1: it overlaps it, reading latest_Bg / td as of frame f-1"""
        self._chk(self.L.vio_set_tracker_lag(self.h, int(lag)), "vio_set_tracker_lag")

    def set_fisheye_mask(self, mask):
        """FISHEYE: the feature mask starts from `mask` (H x W u8, 255 = usable) instead of all 255 (feature_tracker.cpp:175-176); None = off"""
        if mask is None:
            self._chk(self.L.vio_set_fisheye_mask(self.h, None, 0), "vio_set_fisheye_mask")
            return
        on_dev = hasattr(mask, "data_ptr") and getattr(mask, "is_cuda", False)
        m = mask if on_dev else np.ascontiguousarray(mask, np.uint8)
        if not on_dev and m.shape != (self.cfg.height, self.cfg.width):
            raise ValueError("fisheye mask must be height x width")
        self._chk(self.L.vio_set_fisheye_mask(self.h, _ptr(m), 1 if on_dev else 0), "vio_set_fisheye_mask")

    def latest_odometry(self, seq=0):
        """IMU-rate pose (pubLatestOdometry): t, P(3), Q(wxyz), V(3) of the newest window state propagated through the IMU pushed since"""
        o = np.zeros(11)
        self._chk(self.L.vio_get_latest_odometry(self.h, seq, o.ctypes.data), "vio_get_latest_odometry")
        return o

    def set_relo_frame(self, seq, stamp, index, match_points, relo_t, relo_r):
        """Estimator::setReloFrame (estimator.cpp:1728-1747): match_points[n][3] = (x, y, feature id) ascending in id"""
        mp = np.ascontiguousarray(match_points, np.float64).reshape(-1, 3)
        t = np.ascontiguousarray(relo_t, np.float64).reshape(3)
        R = np.ascontiguousarray(relo_r, np.float64).reshape(9)
        self._chk(self.L.vio_set_relo_frame(self.h, seq, float(stamp), int(index), len(mp), mp.ctypes.data, t.ctypes.data, R.ctypes.data), "vio_set_relo_frame")

    def relo(self, seq=0):
        o = np.zeros(30)
        self._chk(self.L.vio_get_relo(self.h, seq, o.ctypes.data), "vio_get_relo")
        return dict(relative_t=o[0:3], relative_q=o[3:7], relative_yaw=o[7], drift_t=o[8:11], drift_r=o[11:20].reshape(3, 3), relo_pose=o[20:27],
                    pending=int(o[27]), local_index=int(o[28]), n_factors=int(o[29]))

    def extrinsic(self, seq=0):
        e = np.zeros(13)
        self._chk(self.L.vio_get_extrinsic(self.h, seq, e.ctypes.data), "vio_get_extrinsic")
        return e

    def tracks(self, seq=0, cap=2048):
        ids, cnt = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        cur, un, vel = np.zeros((cap, 2), np.float32), np.zeros((cap, 2), np.float32), np.zeros((cap, 2), np.float32)
        n = self._chk(self.L.vio_get_tracks(self.h, seq, cap, ids.ctypes.data, cnt.ctypes.data, cur.ctypes.data, un.ctypes.data,
                                            vel.ctypes.data), "vio_get_tracks")
        return ids[:n], cnt[:n], cur[:n], un[:n], vel[:n]

    def landmarks(self, seq=0, cap=4096):
        out = np.zeros((cap, 7))
        n = self._chk(self.L.vio_get_landmarks(self.h, seq, cap, out.ctypes.data), "vio_get_landmarks")
        return out[:min(n, cap)]

    def landmarks_ex(self, seq=0, cap=4096):
        out = np.zeros((cap, 12))
        n = self._chk(self.L.vio_get_landmarks_ex(self.h, seq, cap, out.ctypes.data), "vio_get_landmarks_ex")
        return out[:min(n, cap)]

    def prior(self, seq=0):
        n = 6 * self.W + 16
        J, r, x0, pres = np.zeros((n, n)), np.zeros(n), np.zeros(self.W * 7 + 17), np.zeros(self.W + 3, np.uint8)
        k = self._chk(self.L.vio_get_prior(self.h, seq, J.ctypes.data, r.ctypes.data, x0.ctypes.data, pres.ctypes.data), "vio_get_prior")
        return (J, r, x0, pres) if k else None

    KERNELS = ("fe_begin", "fe_pyrdown", "fe_predict", "fe_lk", "fe_select", "fe_fast", "fe_add", "be_ingest", "be_solve", "be_marg")

    def profile_begin(self, max_steps):
        self._chk(self.L.vio_profile_begin(self.h, max_steps), "vio_profile_begin")

    def profile_end(self):
        t = np.zeros(len(self.KERNELS))
        n = self._chk(self.L.vio_profile_end(self.h, len(t), t.ctypes.data), "vio_profile_end")
        return n, dict(zip(self.KERNELS, t.tolist()))

    def timings(self):
        t = np.zeros(8)
        n = self._chk(self.L.vio_get_timings(self.h, 8, t.ctypes.data), "vio_get_timings")
        return t[:n]


# ------------------------------------------------------------------------------------------------ reference-shaped mirrors
class FeatureTracker:
    """FeatureTracker (feature_tracker.h:31-97) backed by one HBM-resident sequence."""

    def __init__(self, batch, seq=0):
        self.b, self.seq = batch, seq
        self.ids = self.track_cnt = self.cur_pts = self.cur_un_pts = self.pts_velocity = None

    def readImage(self, img, cur_time, relative_R=None, publish=True):
        """readImage(const cv::Mat&, double, const Matrix3d& relative_R = Identity) (feature_tracker.h:36-37); publish is the
        global PUB_THIS_FRAME.  relative_R=None predicts it on the device from the pushed IMU (Estimator::predictMotion)."""
        assert self.b.S == 1, "the per-sequence mirror drives single-sequence batches"
        R = None if relative_R is None else np.asarray(relative_R, np.float64).reshape(1, 9)
        self.b.track(np.ascontiguousarray(img, np.uint8), [cur_time], publish=publish, R_rel=R)
        self.ids, self.track_cnt, self.cur_pts, self.cur_un_pts, self.pts_velocity = self.b.tracks(self.seq)

    def updateID(self, i):
        """ids are assigned on the device inside readImage; kept for call-site compatibility (estimator_nodelet.cpp:324-330)."""
        return i < len(self.ids)

    def image_map(self):
        """The feature map of estimator_nodelet.cpp:336-363 for the last published frame: {feature_id: (x, y, 1, u, v, vx, vy)}."""
        ids, obs = self.b.packaged(self.seq)
        return {int(i): o for i, o in zip(ids, obs)}


class Estimator:
    """Estimator (estimator.h:25-201) call surface backed by one HBM-resident sequence."""

    def __init__(self, cfg=None, imu_capacity=8192):
        self.batch = VioBatch(cfg, 1, imu_capacity)
        self.featureTracker = FeatureTracker(self.batch)

    def inputIMU(self, t, linearAcceleration, angularVelocity):
        self.batch.push_imu(0, [t], [linearAcceleration], [angularVelocity])

    def predictMotion(self, t0, t1):
        """Matrix3d predictMotion(double t0, double t1) (estimator.h:56, estimator.cpp:1790-1860)."""
        return self.batch.predict_motion(0, t0, t1)

    def processImage(self, image, header, depth_mm):
        """processImage(const map<int, Matrix<double,7,1>>& image, const Header& header) (estimator.h:46) preceded by
        f_manager.inputDepth(depth) (estimator_nodelet.cpp:537-539).  image: {feature_id: 7-vector}; header: stamp in seconds."""
        ids = sorted(image)
        obs = np.array([image[i] for i in ids], np.float64).reshape(-1, 7)
        self.batch.process_obs(0, ids, obs, depth_mm, header)
        return self.batch.status(0).code

    def processLastTracked(self, depth_mm):
        """Short cut without the host round trip: inputDepth + processImage on the map the last readImage packaged on the device."""
        self.batch.process(np.ascontiguousarray(depth_mm, np.uint16))
        return self.batch.status(0).code

    def clearState(self):
        """Estimator::clearState() (estimator.cpp:43-116): the estimator side only, the FeatureTracker keeps its state"""
        self.batch.reset_seq(0)

    def setParameter(self):
        """estimator.cpp:15-41: parameters are bound at construction (vio_reset_seq re-applies them)"""

    @property
    def solver_flag(self):
        return self.batch.status(0).solver_flag

    def window(self):
        return self.batch.window(0)
