"""Round-3 GPU tests: the north-star ATE criterion with statistical power (128 sequences x 300 frames, strict 1 %), the literal
marginalisation (vio_config.marg_exact) as a causality instrument for the long-run divergence, and the HIP path driven by two ranks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import parity_long
import vio_ct

pytestmark = pytest.mark.gpu


GOLDEN_ATE = {0: os.path.join(vio_ct.ROOT, "tests", "golden", "oracle_ate_300.npz"),
              1: os.path.join(vio_ct.ROOT, "tests", "golden", "oracle_ate_300_lag1.npz")}


@pytest.mark.parametrize("lag", [0, 1])
def test_300_frames_ate_within_one_percent_strict(P, lag):
    """The north-star criterion -- ATE of the HIP path within 1 % of the reference algorithm's on identical input -- with the statistical
    power it needs.  Measured (profiles/round3_parity_300_s128.json): after ~50 frames HIP and oracle are two realisations of a chaotic
    estimator (126 of 128 sequences separate by more than 1 um within 300 frames whichever marginalisation form the HIP side uses, the
    per-sequence ATE difference scatters with sigma = 1.65 mm around a 15.8 mm mean), so the mean over 128 sequences still carries a
    standard error of 0.93 % -- it cannot resolve 1 %.  Over 1024 sequences the standard error is 0.33 %: the oracle's side of that
    comparison (8 CPU hours, no GPU needed) is the committed fixture tests/golden/oracle_ate_300.npz (generator next to it), the HIP
    side runs here, on identical pixels (device renderer == host renderer, asserted).  Asserted STRICTLY:
        |mean ATE_hip - mean ATE_oracle| <= 1 % of mean ATE_oracle,   and that the sample can resolve it (standard error < 0.5 %).
    lag = 1 is the mode bench.py measures (vio_set_tracker_lag(1): the tracker of frame f + 1 is predicted from the window as it was
    after frame f - 1); its fixture oracle_ate_300_lag1.npz is the oracle run with the same one-frame-late prediction
    (tests/oracle_control.py, 1024 sequences x 300 frames), so the criterion covers the measured configuration too."""
    fx = np.load(GOLDEN_ATE[lag])
    assert int(fx["tracker_lag"]) == lag if "tracker_lag" in fx.files else lag == 0
    seq0, n_frames, ate_o = int(fx["seq0"]), int(fx["frames"]), fx["ate"]
    N = len(ate_o)
    assert N >= 512 and int(fx["reboots"].sum()) == 0
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    syn = P.Synth(sc)
    ate_h, maxdist = np.zeros(N), []
    S = 128
    for b0 in range(0, N, S):
        n = min(S, N - b0)
        hist, stats, t_feed = parity_long.run_hip(P, cfg, sc, seq0 + b0, n, n_frames, lag=lag, check_render=(b0 == 0))
        assert all(st.reboot_count == 0 and st.solver_flag == 1 for st in stats)
        for i in range(n):
            h = hist[i]
            assert len(h) == int(fx["n_rows"][b0 + i]), (b0 + i, len(h), int(fx["n_rows"][b0 + i]))
            gt = np.array([syn.pose(seq0 + b0 + i, float(t))[0] for t in h[:, 0]])
            ate_h[b0 + i] = vio_ct.ate_rmse(h[:, 1:4], gt)
            if b0 + i < len(fx["positions"]):
                f0 = int(fx["first_frame"][b0 + i])
                po = fx["positions"][b0 + i][f0:f0 + len(h)]
                maxdist.append(float(np.linalg.norm(po - h[:, 1:4], axis=1).max()))
    d = ate_h - ate_o
    rel = abs(ate_h.mean() - ate_o.mean()) / ate_o.mean()
    se_rel = d.std(ddof=1) / np.sqrt(N) / ate_o.mean()
    out_dir = os.path.join(vio_ct.ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        json.dump(dict(tracker_lag=lag, sequences=N, first_sequence=seq0, frames=n_frames, mean_ate_oracle_m=float(ate_o.mean()), mean_ate_hip_m=float(ate_h.mean()),
                       signed_rel_diff_of_means=float((ate_h.mean() - ate_o.mean()) / ate_o.mean()), standard_error_rel=float(se_rel),
                       std_of_pair_difference_m=float(d.std(ddof=1)), hip_better=int((d < 0).sum()), hip_worse=int((d > 0).sum()),
                       max_distance_first_128=dict(median=float(np.median(maxdist)), max=float(np.max(maxdist)), beyond_1um=int((np.array(maxdist) > 1e-6).sum())),
                       ate_hip_m=ate_h.tolist()), open(os.path.join(out_dir, "parity_300_s1024%s.json" % ("_lag1" if lag else "")), "w"), indent=1)
    assert se_rel < 0.005, se_rel
    assert rel <= 0.01, (rel, se_rel, float(ate_o.mean()), float(ate_h.mean()))
    assert ate_h.max() < 0.1


def test_literal_marginalisation_mode_tracks_the_oracle(P):
    """vio_config.marg_exact = 1 (MarginalizationInfo::marginalize followed literally, marginalization_factor.cpp:281-315: full m x m
    eigen-decomposition with the 1e-8 cut, prior rebuilt from the truncated factors) against the oracle over 70 frames x 8 sequences: no
    reboots, the same frames processed, positions within 1e-5 m over the first 30 processed frames of every sequence -- the parity
    instrument behind profiles/round3_parity_300_s128.json works as a marginalisation.  Beyond that horizon a sequence may separate from
    the oracle by a flipped decision (measured on this set: sequence 704 jumps to 1.7 mm after frame 39 in BOTH modes, five sequences stay
    below 3e-5 m to the end), so the tail is bounded statistically: at least six of the eight within 1e-4 m, none beyond 1 cm (the ATE of
    these sequences is ~15 mm).  What the instrument showed at 300 frames: with it 127 of 128 sequences still separate from the oracle,
    126 without it -- the long-run divergence is not caused by the fast form's deviations from the reference."""
    cfg = P.canonical_config(marg_exact=1)
    sc = vio_ct.synth_like(cfg)
    S, seq0, n_frames = 8, 700, 70
    hist, stats, _ = parity_long.run_hip(P, cfg, sc, seq0, S, n_frames, check_render=False)
    assert all(st.reboot_count == 0 and st.solver_flag == 1 for st in stats)
    orc = parity_long.run_oracle_pool(range(seq0, seq0 + S), n_frames, procs=S)
    head, tail = [], []
    for i in range(S):
        fr, po, gt, reb = orc[seq0 + i]
        assert reb == 0 and len(hist[i]) == len(po) >= 50
        d = np.abs(hist[i][:, 1:4] - po).max(1)
        head.append(float(d[:30].max()))
        tail.append(float(d.max()))
    assert max(head) < 1e-5, head
    assert sum(v < 1e-4 for v in tail) >= 6 and max(tail) < 1e-2, tail


def test_certified_literal_marginalisation_mode(P):
    """vio_config.marg_exact = 2 (round 5): MarginalizationInfo::marginalize with its first eigen-decomposition replaced by a certified inverse
    (the pseudo-inverse with the 1e-8 cut IS the inverse when no eigenvalue is near the cut, which every frame proves from a bound on
    |A_mm^-1|_F) and its second half -- the eigen-decomposition of the new prior with the cut that does drop directions -- followed literally
    (marginalization_factor.cpp:293-315, LDS-resident).  Against the oracle like the fully literal mode above (1e-5 m over the first 30
    processed frames), every marginalisation certified, and two orders of magnitude cheaper than marg_exact = 1 on this workload, whose
    marginalised block (m = 150 .. 215) does not fit LDS."""
    import ctypes as C
    cfg = P.canonical_config(marg_exact=2)
    sc = vio_ct.synth_like(cfg)
    S, seq0, n_frames = 8, 700, 70
    hist, stats, t_feed, b = parity_long.run_hip(P, cfg, sc, seq0, S, n_frames, check_render=False, keep=True)
    assert all(st.reboot_count == 0 and st.solver_flag == 1 for st in stats)
    L = P.lib()
    L.vio_debug_seq.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    dbg = np.zeros(16, np.int32)
    for i in range(S):
        L.vio_debug_seq(b.h, i, dbg.ctypes.data)
        assert dbg[12] == 0 and dbg[11] == 1, (i, dbg[11], dbg[12])        # no uncertified marginalisation in 70 frames; the last one certified
        assert b.marg_certificate(i) == (0, True)                          # ... and the same through the C ABI (vio_get_marg_certificate)
        assert dbg[0] == 0                                                  # the prior's eigen-decomposition ran LDS-resident (no Jacobi sweeps)
    b.close()
    orc = parity_long.run_oracle_pool(range(seq0, seq0 + S), n_frames, procs=S)
    head, tail = [], []
    for i in range(S):
        fr, po, gt, reb = orc[seq0 + i]
        assert reb == 0 and len(hist[i]) == len(po) >= 50
        d = np.abs(hist[i][:, 1:4] - po).max(1)
        head.append(float(d[:30].max()))
        tail.append(float(d.max()))
    assert max(head) < 1e-5, head
    assert sum(v < 1e-4 for v in tail) >= 6 and max(tail) < 1e-2, tail


def test_long_run_on_identical_frames_hip_equals_the_matched_oracle(P):
    """The experiment VERDICT r4 asked for (item 1), in small: 32 sequences x 200 frames, the HIP path on DEVICE-rendered frames against the
    oracle -- switched to the HIP path's formulations (OVIO_DEVIATIONS = 31) -- on HOST-rendered frames.  Since round 6 the two renderers are one
    (tests/test_gpu_render.py), so no frames have to be carried across (rounds 3 - 4 compared runs whose inputs differed in 1e-7 of the pixels,
    which is what made 126 of 128 sequences separate; round 5 fed the oracle the device's frames through /dev/shm).  At 128 x 300
    (test_census_128_sequences_against_the_matched_oracle): 110 of 128 sequences identical to 1 um over all 300 frames, median largest distance
    5e-10 m, 18 separated -- fewer than two round-off builds of the oracle itself (53 - 58 of 128, profiles/round4_oracle_self_divergence.json).
    Here: most sequences identical to 1 um, the median largest distance below 1e-8 m, the early frames at round-off."""
    rep = parity_long.run(P, S=32, seq0=700, n_frames=200, lag=0, modes=("fast",), oracle_devs=(31,), same_frames=False)
    sm = rep["modes"]["fast_vs_oracle_dev31"]["summary"]
    out_dir = os.path.join(vio_ct.ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        json.dump(rep, open(os.path.join(out_dir, "parity_200_s32_one_renderer.json"), "w"), indent=1)
    assert rep["modes"]["fast_vs_oracle_dev31"]["hip_reboots"] == 0
    assert sm["identical_to_1um"] >= 22, sm                     # (128 x 300: 110 of 128)
    assert sm["median_max_distance_m"] < 1e-8, sm               # (128 x 300: 5.3e-10)
    assert sm["early30_max_distance_m"]["median"] < 2e-9, sm    # (128 x 300: 1.2e-10)


def test_census_128_sequences_against_the_matched_oracle(P):
    """The 128 x 300 census as an assertion: the HIP path on device-rendered frames against the committed positions of the
    oracle with the HIP formulations (tests/golden/oracle_dev31_300.npz: sequences 700 .. 827, 300 frames, tracker lag 0, host-rendered frames,
    OVIO_DEVIATIONS = 31; generator: tests/oracle_control.py run --only devallc + census-fixture).  Round 5 measured 18 of 128 sequences more
    than 1 um apart at some frame (profiles/round5_parity_300_s128_same_frames.json); two round-off builds of the oracle itself: 53 - 58."""
    path = os.path.join(vio_ct.ROOT, "tests", "golden", "oracle_dev31_300.npz")
    fx = np.load(path)
    seq0, n_frames, S = int(fx["seq0"]), int(fx["frames"]), len(fx["first_frame"])
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    hist, stats, _ = parity_long.run_hip(P, cfg, sc, seq0, S, n_frames, lag=0)
    far, dmax = 0, []
    for i in range(S):
        f0, n = int(fx["first_frame"][i]), int(fx["n_rows"][i])
        assert stats[i].reboot_count == 0 and len(hist[i]) == n, (i, len(hist[i]), n)
        d = np.linalg.norm(hist[i][:, 1:4] - fx["positions"][i, f0:f0 + n], axis=1)
        dmax.append(float(d.max()))
        far += int(d.max() > 1e-6)
    out_dir = os.path.join(vio_ct.ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        json.dump(dict(sequences=S, frames=n_frames, beyond_1um=far, median_max_distance_m=float(np.median(dmax)), max_distance_m=dmax),
                  open(os.path.join(out_dir, "census_128x300.json"), "w"), indent=1)
    assert far <= 25, (far, float(np.median(dmax)))
    assert np.median(dmax) < 1e-8, float(np.median(dmax))


def _bench(args, env_extra, timeout=900):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(vio_ct.ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_two_ranks_drive_the_hip_path_on_one_gpu(tmp_path):
    """The N > 1 control flow of bench.py executed for real: two torch.distributed ranks (gloo rendezvous, both on cuda:0 -- this box has
    one GPU) each own 16 sequences (global ids 0..15 and 16..31), no data-path collective, one reduction of the job totals.  The job
    line must account for both shards, and rank 1's windows must be bit-identical to a single-rank run of the same global ids:
    results do not depend on which rank / how many ranks computed them."""
    common = ["--seqs", "16", "--steps", "8", "--warmup", "4", "--repeats", "1", "--cpu-seqs", "0", "--cpu-procs", "0", "--pcie-steps", "0",
              "--stream-steps", "0", "--aux", "0"]
    d2, d1 = str(tmp_path / "two"), str(tmp_path / "one")
    j2 = _bench(["--gpus", "2"] + common + ["--dump", d2], {"VIO_BENCH_DEVICE": "0", "VIO_BENCH_BACKEND": "gloo"})
    assert j2["n_gpus"] == 2 and j2["valid"] is True and j2["scaling"] == "weak"
    frames = j2["value"] * j2["ms_per_step"] * 1e-3 * j2["steps"]
    assert abs(frames - 2 * 16 * 8) < 1e-6 * frames, frames
    # the per-rank table: both ranks accounted for, each with its own shard, rate and validity (a slow or duplicated device would show here;
    # the distinct-device assertion is waived by VIO_BENCH_DEVICE, which this one-GPU box needs)
    pr = j2["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and all(r["valid"] for r in pr) and all(r["frames_per_s"] > 0 for r in pr)
    assert all(r["ate_rms_m"] is not None and r["ate_rms_m"] < 0.2 for r in pr)
    assert pr[0]["device"] == pr[1]["device"]                                      # (same physical device here, by construction)
    assert j2["value"] <= sum(r["frames_per_s"] for r in pr) * (1 + 1e-9)          # the job rate uses the MAX-over-ranks time
    j1 = _bench(["--gpus", "1"] + common + ["--seq-offset", "16", "--dump", d1], {})
    assert j1["n_gpus"] == 1 and j1["valid"] is True
    a, b = np.load(d2 + ".rank1.npz"), np.load(d1 + ".rank0.npz")
    assert int(a["seq0"]) == 16 and int(b["seq0"]) == 16
    assert np.array_equal(a["windows"], b["windows"]) and np.array_equal(a["odometry"], b["odometry"])
    r0 = np.load(d2 + ".rank0.npz")
    assert int(r0["seq0"]) == 0 and not np.array_equal(r0["windows"], a["windows"])


def test_config5_batch_of_64_on_the_phased_solver(P):
    """BASELINE configs[4] at its per-GPU batch: 64 sequences x 1280x720, 300 features, 7x8 grid, 20-keyframe window (W is a compile-time
    10 upstream, parameters.h:12; a run-time parameter here).  P = 322: the Schur complement (231 tiles) does not fit LDS, the phased
    solver keeps it in HBM / L2 and streams it through the Cholesky one block column at a time (ps_serial_big_kernel).  Four sequences
    against the oracle (identical decisions every frame, positions within 1e-5 m); every fourth sequence of the 64 bit-identical to its
    stand-alone (S = 1) run."""
    kw = dict(width=1280, height=720, max_cnt=300, window_size=20, grid_rows=7, grid_cols=8, max_landmarks=2048,
              fx=604.5821781259577 * 2, fy=604.2544712985845 * 1.5, cx=321.2638233484251 * 2, cy=239.70969315130674 * 1.5)
    cfg = P.canonical_config(**kw)
    sc = vio_ct.synth_like(cfg)
    S, seq0, n_frames = 64, 40, 44
    chk = [0, 21, 42, 63]
    seen = {i: [] for i in chk}
    def grab(f, b):
        for i in chk:
            st = b.status(i)
            seen[i].append((st.solver_flag, st.frame_count, st.n_landmarks, st.marginalization_flag, st.n_residuals, st.n_in_problem, st.n_var_landmarks,
                            st.iterations, st.processed, st.code))
    # (the oracle gets the frames the device rendered: at this resolution the host renderer differs from the device's in a few pixels)
    frames = {i: [] for i in chk}
    hist, stats, t_feed, b = parity_long.run_hip(P, cfg, sc, seq0, S, n_frames, chunk=11, check_render=False, per_frame=grab, keep=True, grab=frames)
    assert all(st.solver_flag == 1 and st.reboot_count == 0 and st.overflow_frames == 0 for st in stats)
    assert b.solver_kind() == 2                  # the phased solver with the HBM-resident Schur complement, not the round-1 fallback
    wins = [b.window(i).copy() for i in range(S)]
    b.close()
    orc = parity_long.run_oracle_pool([seq0 + i for i in chk], n_frames, cfg_kw=dict(kw, _status=True), procs=len(chk),
                                      frames={seq0 + i: frames[i] for i in chk})
    for i in chk:
        fr, po, gt, reb, ost, oproc = orc[seq0 + i]
        assert reb == 0 and len(po) >= 18
        for f in range(n_frames):
            sh = seen[i][f]
            assert tuple(int(x) for x in ost[f][:3]) == sh[:3], (i, f, ost[f], sh)
            if sh[0] == 1 and sh[8]:
                assert tuple(int(x) for x in ost[f][3:8]) == sh[3:8], (i, f, ost[f], sh)      # marginalisation flag, residuals, landmarks, iterations
        h = hist[i]
        assert len(h) == len(po) and np.abs(h[:, 1:4] - po).max() < 1e-5, (i, float(np.abs(h[:, 1:4] - po).max()))
        assert vio_ct.ate_rmse(h[:, 1:4], gt) < 0.03
    for i in range(0, S, 4):
        h1, st1, _, b1 = parity_long.run_hip(P, cfg, sc, seq0 + i, 1, n_frames, chunk=n_frames, check_render=False, keep=True)
        assert np.array_equal(b1.window(0), wins[i]) and np.array_equal(h1[0], hist[i]), i
        b1.close()


def test_relocalisation_factors_match_the_oracle_and_the_truth(P):
    """SURVEY.md 8f rank 4, first slice: vio_set_relo_frame (Estimator::setReloFrame, estimator.cpp:1728-1747), the relocalisation
    factors inside optimization() (:1307-1346) and the drift / relative-pose outputs (:1034-1056), HIP against the oracle on the same
    scenario as tests/test_oracle_relo_cpu.py (the copy of window pose i is pulled onto the pose of frame i - 2 by the projection
    factors alone), and against the ground truth of the scene."""
    import test_oracle_relo_cpu as R
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    seq, n_frames, f_set, i_local, back = 3, 40, 36, 6, 2
    ro, ctx_o, o = R.drive_with_relo(P, cfg, sc, seq, n_frames, f_set, i_local, back, **R.oracle_accessors(cfg))

    def feed_obs(b, f, tf, g, d, imu):
        b.push_imu(0, *imu)
        b.feed(g[None], d[None], [tf])
        ids, obs = b.packaged(0)
        return ids, obs, True
    acc = dict(make_pipe=lambda: P.VioBatch(cfg, 1), set_relo=lambda b, *a: b.set_relo_frame(0, *a), get_relo=lambda b: b.relo(0),
               window_of=lambda b: b.window(0), feed_obs=feed_obs)
    rh, ctx_h, b = R.drive_with_relo(P, cfg, sc, seq, n_frames, f_set, i_local, back, **acc)
    assert ctx_h["n_match"] == ctx_o["n_match"] and abs(ctx_h["stamp_i"] - ctx_o["stamp_i"]) < 1e-12
    assert rh["pending"] == 0 and rh["local_index"] == ro["local_index"] == i_local
    assert rh["n_factors"] == ro["n_factors"] >= 30
    st = b.status(0)
    assert st.overflow_flags == 0 and st.solver_flag == 1
    # against the oracle
    assert np.abs(rh["relative_t"] - ro["relative_t"]).max() < 1e-6, (rh["relative_t"], ro["relative_t"])
    assert abs(rh["relative_yaw"] - ro["relative_yaw"]) < 1e-5
    sgn = np.sign(np.dot(rh["relative_q"], ro["relative_q"]))
    assert np.abs(sgn * rh["relative_q"] - ro["relative_q"]).max() < 1e-7
    assert np.abs(rh["drift_t"] - ro["drift_t"]).max() < 1e-6 and np.abs(rh["drift_r"] - ro["drift_r"]).max() < 1e-8
    assert np.abs(rh["relo_pose"] - ro["relo_pose"]).max() < 1e-6
    assert np.abs(ctx_h["window_after"][:, :3] - ctx_o["window_after"][:, :3]).max() < 1e-5      # the window itself after the relocalisation solve
    # against the truth (same bounds as the oracle's own test)
    rel_t = ctx_h["R_gt_k"].T @ (ctx_h["p_gt_i"] - ctx_h["p_gt_k"])
    assert np.abs(rh["relative_t"] - rel_t).max() < 0.01
    # a frame without a request runs the ordinary problem again (the request is consumed by exactly one solve)
    assert b.relo(0)["pending"] == 0
    # a stamp that is not in the window is ignored, like upstream
    b.set_relo_frame(0, 123.456, 1, np.array([[0.0, 0.0, 5.0]]), np.zeros(3), np.eye(3))
    assert b.relo(0)["pending"] == 0


def test_relocalisation_while_the_extrinsic_is_optimised_holds_it_for_that_solve(P):
    """DEVIATION 15: relo_Pose borrows the six tangent columns of the extrinsic, so with estimate_extrinsic 1 (the extrinsic is a
    variable of every solve once openExEstimation has latched, estimator.cpp:1187-1202) the ONE solve that carries relocalisation factors
    holds the extrinsic constant; the request is honoured (no flag), the next solve refines the extrinsic again.  Checked three ways:
    against the oracle with the same rule switched on (reference_quirks bit 2: 1e-6, like the constant-extrinsic test), against the oracle as
    the reference has it (joint optimisation of extrinsic and relo_Pose: the deviation is below a millimetre), and against the truth."""
    import test_oracle_relo_cpu as R
    sc = vio_ct.synth_like(P.canonical_config())
    seq, n_frames, f_set, i_local, back = 3, 60, 56, 6, 2     # late enough for openExEstimation to have latched
    cfg = P.canonical_config(estimate_extrinsic=1)
    cfg_mirror = P.canonical_config(estimate_extrinsic=1, reference_quirks=4)

    def feed_obs(b, f, tf, g, d, imu):
        b.push_imu(0, *imu)
        b.feed(g[None], d[None], [tf])
        ids, obs = b.packaged(0)
        return ids, obs, True
    ex_hist = []

    def window_of(b):
        ex_hist.append(b.extrinsic(0).copy())
        return b.window(0)
    acc = dict(make_pipe=lambda: P.VioBatch(cfg, 1), set_relo=lambda b, *a: b.set_relo_frame(0, *a), get_relo=lambda b: b.relo(0),
               window_of=window_of, feed_obs=feed_obs)
    rh, ctx_h, b = R.drive_with_relo(P, cfg, sc, seq, n_frames, f_set, i_local, back, **acc)
    st = b.status(0)
    assert rh["pending"] == 0 and rh["local_index"] == i_local and rh["n_factors"] >= 30 and st.overflow_flags == 0 and st.solver_flag == 1
    assert np.abs(ex_hist[0][:12] - np.r_[list(cfg.tic), list(cfg.ric)]).max() > 1e-6         # the extrinsic HAS been refined before the request
    assert np.abs(ex_hist[0] - ex_hist[1]).max() < 1e-12       # ... and is held by the relocalisation solve (double2vector's q -> R round trip only)
    # the oracle with the same rule
    ro, ctx_o, o = R.drive_with_relo(P, cfg_mirror, sc, seq, n_frames, f_set, i_local, back, **R.oracle_accessors(cfg_mirror))
    assert rh["n_factors"] == ro["n_factors"]
    assert np.abs(rh["relative_t"] - ro["relative_t"]).max() < 1e-6 and abs(rh["relative_yaw"] - ro["relative_yaw"]) < 1e-5
    assert np.abs(rh["drift_t"] - ro["drift_t"]).max() < 1e-6 and np.abs(rh["relo_pose"] - ro["relo_pose"]).max() < 1e-6
    assert np.abs(ctx_h["window_after"][:, :3] - ctx_o["window_after"][:, :3]).max() < 1e-5
    # the oracle as the reference has it: extrinsic and relo_Pose optimised together
    rj, ctx_j, oj = R.drive_with_relo(P, cfg, sc, seq, n_frames, f_set, i_local, back, **R.oracle_accessors(cfg))
    assert rj["n_factors"] == rh["n_factors"]
    dev = float(np.abs(rh["relative_t"] - rj["relative_t"]).max())
    assert 0 < dev < 1e-3, dev
    assert abs(rh["relative_yaw"] - rj["relative_yaw"]) < 0.02
    # the truth (looser than with the true extrinsic: on this trajectory the online calibration has wandered by then, for the oracle alike)
    rel_t = ctx_h["R_gt_k"].T @ (ctx_h["p_gt_i"] - ctx_h["p_gt_k"])
    assert np.abs(rh["relative_t"] - rel_t).max() < 0.03 and np.abs(rj["relative_t"] - rel_t).max() < 0.03
    # the following solves (frames f_set + 2 ...) refine the extrinsic again
    assert np.abs(b.extrinsic(0) - ex_hist[1]).max() > 1e-7
    b.close()


def test_two_cholesky_retries_in_one_solve_follow_the_oracle(P, monkeypatch):
    """The mu *= 10 retry ladder of the trust-region loop (oracle/backend.cpp solve(), Ceres' handling of a failed linear solve): with the
    test hook VIO_TEST_CHOL_FAIL_SHIFT the first TWO factorisations of every solve are reported as failed on both sides.  Every retry
    re-forms the Schur complement at the larger mu and costs the phased solver one iteration slot, so the handle is created with
    VIO_EXTRA_SLOTS = 4 (final evaluation + three spare); iteration counts, accepted steps, costs and the window must follow the oracle
    frame by frame.  With the default two extra slots the same run is truncated and says so (overflow flag 32) -- the documented limit."""
    n, seqs = 26, [31, 32]
    cfg = P.canonical_config(reference_quirks=2 << 8)
    sc = vio_ct.synth_like(cfg)
    oruns = [vio_ct.run_oracle_sequence(cfg, sc, s, n) for s in seqs]
    frames = [o["frames"] for o in oruns]
    monkeypatch.setenv("VIO_EXTRA_SLOTS", "4")
    b, traj, stat = vio_ct.run_hip_batch(P, cfg, sc, seqs, n, frames)
    nl = 0
    for i in range(len(seqs)):
        for f in range(n):
            so, sh = oruns[i]["status"][f], stat[i][f]
            assert (int(so["solver_flag"]), int(so["frame_count"]), int(so["n_landmarks"]), int(so["marginalization_flag"])) == \
                   (sh.solver_flag, sh.frame_count, sh.n_landmarks, sh.marginalization_flag), (i, f)
            if sh.solver_flag == 1 and sh.processed:
                nl += 1
                assert (int(so["iterations"]), int(so["successful_steps"])) == (sh.iterations, sh.successful_steps), (i, f, so["iterations"], sh.iterations)
                assert abs(so["final_cost"] - sh.final_cost) <= 1e-6 * max(1.0, abs(so["final_cost"])), (i, f)
            assert sh.overflow_flags & 32 == 0, (i, f)
        wo, wh = oruns[i]["oracle"].window(), b.window(i)
        assert np.abs(wo[:, :3] - wh[:, :3]).max() < 1e-5
    assert nl >= 20
    # the retries really happened: the unforced run of the same sequences ends elsewhere (mu stays higher after the forced failures)
    cfg0 = P.canonical_config()
    b0, _, _ = vio_ct.run_hip_batch(P, cfg0, sc, seqs, n, frames)
    assert np.abs(b0.window(0)[:, :3] - b.window(0)[:, :3]).max() > 1e-9
    b.close(); b0.close()
    # default slots: one spare only -> the second forced retry exhausts them; the frame is flagged, never silently wrong
    monkeypatch.delenv("VIO_EXTRA_SLOTS")
    b2, _, stat2 = vio_ct.run_hip_batch(P, cfg, sc, seqs[:1], n, frames[:1])
    assert any(s.overflow_flags & 32 for s in stat2[0] if s.solver_flag == 1)
    b2.close()
