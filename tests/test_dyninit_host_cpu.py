"""The host-side building blocks of the PRODUCT's dynamic initialisation (vins-rgbd-fast_amd/csrc/dyninit_host.cpp, exported as
vio_stage_host_*; no GPU involved) against the oracle's restatement of the same reference routines (oracle/initial.cpp) on identical
inputs: cv::solvePnP ITERATIVE, cv::solvePnPRansac(EPNP) with OpenCV's RNG stream, relativePose + GlobalSFM::construct.  The two
implementations share no code (own dense algebra, own EPnP / Levenberg-Marquardt / bundle adjustment)."""
import ctypes as C

import numpy as np
import pytest

import test_oracle_init_cpu as oi


@pytest.fixture(scope="module")
def sfm(orc):
    orc.ovio_solve_pnp_iterative.argtypes = [C.c_int] + [C.c_void_p] * 4
    orc.ovio_solve_pnp_ransac_epnp.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    orc.ovio_sfm_window.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 8
    return orc


@pytest.fixture(scope="module")
def host(P):
    L = P.lib()
    L.vio_stage_host_pnp.argtypes = [C.c_int] + [C.c_void_p] * 4
    L.vio_stage_host_pnp_ransac.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    L.vio_stage_host_sfm_window.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 8
    L.vio_stage_host_alignment.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    return L


@pytest.mark.parametrize("seed,n", [(1, 40), (2, 12), (3, 200)])
def test_solve_pnp_iterative_matches_the_oracle(host, sfm, seed, n):
    rng, X, m, Rt, tt = oi.scene(n, seed)
    m = m + rng.normal(0, 0.5 / 460, m.shape)                      # half a pixel of noise: the optimum is not the true pose
    R0 = np.ascontiguousarray(oi.rot([0.3, -1, 0.2], 0.12))
    t0 = np.array([0.2, 0.0, 0.1])
    Ro, to = R0.copy(), t0.copy()
    Rh, th = R0.copy(), t0.copy()
    assert sfm.ovio_solve_pnp_iterative(n, X.ctypes.data, m.ctypes.data, Ro.ctypes.data, to.ctypes.data) == 1
    assert host.vio_stage_host_pnp(n, X.ctypes.data, m.ctypes.data, Rh.ctypes.data, th.ctypes.data) == 1
    # both iterate CvLevMarq to its epsilon (FLT_EPSILON on the parameter change): agreement to ~1e-7, far inside the noise
    assert np.abs(Rh - Ro).max() < 5e-7 and np.abs(th - to).max() < 5e-7, (float(np.abs(Rh - Ro).max()), float(np.abs(th - to).max()))
    assert np.abs(Rh @ Rh.T - np.eye(3)).max() < 1e-12
    assert host.vio_stage_host_pnp(3, X.ctypes.data, m.ctypes.data, Rh.ctypes.data, th.ctypes.data) == 0


@pytest.mark.parametrize("seed", [5, 6, 7])
def test_epnp_ransac_matches_the_oracle(host, sfm, seed):
    """Same RNG stream (cv::RNG(-1) restated on both sides) -> same 5-point subsets -> same model, refined on the same inliers."""
    rng, X, m, Rt, tt = oi.scene(60, seed)
    bad = rng.choice(60, 15, replace=False)
    m2 = m.copy()
    m2[bad] += rng.uniform(0.05, 0.3, (15, 2)) * rng.choice([-1, 1], (15, 2))
    Ro, to, inl = np.zeros((3, 3)), np.zeros(3), np.zeros(60, np.uint8)
    Rh, th = np.zeros((3, 3)), np.zeros(3)
    assert sfm.ovio_solve_pnp_ransac_epnp(60, X.ctypes.data, m2.ctypes.data, 100, 1 / 460, 0.99, Ro.ctypes.data, to.ctypes.data, inl.ctypes.data) == 1
    assert host.vio_stage_host_pnp_ransac(60, X.ctypes.data, m2.ctypes.data, 100, 1 / 460, 0.99, Rh.ctypes.data, th.ctypes.data) == 1
    assert np.abs(Rh - Ro).max() < 1e-7 and np.abs(th - to).max() < 1e-7
    assert np.abs(Rh - Rt).max() < 1e-6 and np.abs(th - tt).max() < 1e-6
    assert host.vio_stage_host_pnp_ransac(4, X.ctypes.data, m.ctypes.data, 100, 1 / 460, 0.99, Rh.ctypes.data, th.ctypes.data) == 0


@pytest.mark.parametrize("noise_px,depth_noise,seed", [(0.0, 0.0, 2), (0.3, 0.005, 2), (0.3, 0.005, 9)])
def test_sfm_window_matches_the_oracle(host, sfm, noise_px, depth_noise, seed):
    W = 10
    start, nobs, obs, Rwc, pwc, X = oi.window_tracks(W, 160, 0.12, noise_px, depth_noise, seed)
    rc_o, l_o, q_o, T_o, pts_o, st_o = oi.run_sfm(sfm, W, start, nobs, obs)
    nf = len(start)
    l = C.c_int(-1)
    q, T, pts, st = np.zeros((W + 1, 4)), np.zeros((W + 1, 3)), np.zeros((nf, 4)), np.zeros(4)
    rc = host.vio_stage_host_sfm_window(W, nf, start.ctypes.data, nobs.ctypes.data, obs.ctypes.data, C.byref(l), q.ctypes.data, T.ctypes.data,
                                        pts.ctypes.data, st.ctypes.data)
    assert rc == rc_o == 0 and l.value == l_o
    assert np.array_equal(pts[:, 0] > 0, pts_o[:, 0] > 0)                     # the same tracks were triangulated
    assert int(st[0]) == int(st_o[0])                                         # the bundle adjustment took the same number of iterations
    sign = np.sign((q * q_o).sum(1))[:, None]                                 # q and -q are the same rotation
    # the bundle adjustment stops on Ceres' function tolerance on both sides: agreement to ~1e-7 in pose and structure
    assert np.abs(q * sign - q_o).max() < 1e-6, float(np.abs(q * sign - q_o).max())
    assert np.abs(T - T_o).max() < 1e-6, float(np.abs(T - T_o).max())
    ok = pts[:, 0] > 0
    assert np.abs(pts[ok, 1:] - pts_o[ok, 1:]).max() < 1e-5


def test_sfm_window_failure_codes_match_the_oracle(host, sfm):
    W = 10
    for args in ((160, 0.004, 3, 0.0005), (18, 0.12, 4, 0.03)):
        start, nobs, obs, *_ = oi.window_tracks(W, args[0], args[1], 0.0, 0.0, args[2], rot_rate=args[3])
        rc_o, *_ = oi.run_sfm(sfm, W, start, nobs, obs)
        nf = len(start)
        l = C.c_int(-1)
        q, T, pts, st = np.zeros((W + 1, 4)), np.zeros((W + 1, 3)), np.zeros((nf, 4)), np.zeros(4)
        rc = host.vio_stage_host_sfm_window(W, nf, start.ctypes.data, nobs.ctypes.data, obs.ctypes.data, C.byref(l), q.ctypes.data, T.ctypes.data,
                                            pts.ctypes.data, st.ctypes.data)
        assert rc == rc_o == 1


@pytest.mark.parametrize("n,dt,noise", [(11, 0.1, 0.0), (25, 0.05, 0.0), (21, 0.1, 2e-3)])
def test_visual_inertial_alignment_matches_the_oracle(host, orc, n, dt, noise):
    """LinearAlignmentWithDepth + RefineGravityWithDepth: gravity and per-frame velocities, product host code vs oracle (own LDLT each)."""
    tic = np.array([0.05, -0.02, 0.1])
    R_cw = oi.rot([1, 2, -1], 0.9)
    frames, truth = oi.make_frames(n, dt, tic, R_cw, np.array([0.4, -1.0, 2.0]), noise=noise, seed=3)
    orc.ovio_linear_alignment_with_depth.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    go, xo = np.zeros(3), np.zeros(3 * n + 3)
    gh, xh = np.zeros(3), np.zeros(3 * n + 3)
    tic = np.ascontiguousarray(tic)
    ok_o = orc.ovio_linear_alignment_with_depth(n, frames.ctypes.data, tic.ctypes.data, oi.G, go.ctypes.data, xo.ctypes.data)
    ok_h = host.vio_stage_host_alignment(n, frames.ctypes.data, tic.ctypes.data, oi.G, gh.ctypes.data, xh.ctypes.data)
    assert ok_o == ok_h == 1
    assert np.abs(gh - go).max() < 1e-9 and np.abs(xh[:3 * n + 2] - xo[:3 * n + 2]).max() < 1e-8
    assert abs(np.linalg.norm(gh) - oi.G) < 1e-12
