"""The C++ mirror of FeatureTracker / Estimator (include/vio_adapter.hpp) built with g++ and run against the ctypes path."""
import os
import subprocess

import numpy as np
import pytest

import vio_ct

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_adapter_matches_ctypes_path(P, tmp_path):
    exe = str(tmp_path / "adapter_demo")
    pk = os.path.join(ROOT, "vins-rgbd-fast_amd")
    r = subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "adapter_demo.cpp"),
                        "-L" + pk, "-lvio_hip", "-Wl,-rpath," + pk, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    seq, n = 2, 22
    out = subprocess.run([exe, str(seq), str(n)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    rows = np.array([[float(x) for x in line.split()] for line in out.stdout.strip().splitlines()])
    assert len(rows) >= 6
    # the same sequence through vio_feed from Python
    cfg = P.canonical_config()
    sc = P.default_synth()
    syn = P.Synth(sc)
    b = P.VioBatch(cfg, 1)
    ti, ai, gi = syn.imu(seq, int(n / sc.cam_rate * sc.imu_rate) + 64)
    k, ref = 0, []
    for f in range(n):
        tf = f / sc.cam_rate
        k2 = vio_ct.imu_until(ti, k, tf, sc.imu_rate)
        if k2 > k:
            b.push_imu(0, ti[k:k2], ai[k:k2], gi[k:k2])
        k = k2
        g, d = syn.render_host(seq, tf)
        b.feed(g[None], d[None], [tf])
        st = b.status(0)
        if st.solver_flag == 1 and st.processed:
            ref.append(np.r_[tf, b.window(0)[cfg.window_size, :3], len(b.tracks(0)[0])])
    ref = np.array(ref)
    assert rows.shape == ref.shape
    assert np.abs(rows[:, :4] - ref[:, :4]).max() < 2e-9 and np.array_equal(rows[:, 4], ref[:, 4])  # printed with 9 decimals
