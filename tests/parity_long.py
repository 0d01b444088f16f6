"""Long-run parity harness (test infrastructure): the HIP path -- default marginalisation and vio_config.marg_exact -- against the
oracle on many sequences x 300 frames (SURVEY.md 8d sequence length), the oracle side in a process pool (one sequence per core).

    python tests/parity_long.py --seqs 128 --frames 300 --out gpurun_out/parity_300_s128.json

What it measures (VERDICT r2 "next round" item 1):
  * the north-star criterion with statistical power: |mean ATE_hip - mean ATE_oracle| / mean ATE_oracle over S sequences, with the
    standard error of the mean difference, for both marginalisation modes;
  * causality of the long-run divergence: how many sequences separate from the oracle by more than 1e-6 m (and when), with the
    default marginalisation (analytic landmark elimination + quadratic-form prior, DESIGN.md deviations 10 / 13) and with the literal
    one (marginalization_factor.cpp:281-315).
Used by tests/test_gpu_parity3.py; only tests/ may touch the oracle."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _oracle_worker(job):
    """one oracle process (no GPU): n_frames of sequence seq rendered on the host (or the frames handed in: job[4]); returns frames,
    positions, ground truth, reboots"""
    seq, n_frames, lag, cfg_kw = job[:4]
    given = job[4] if len(job) > 4 else None
    if isinstance(given, str):   # a directory of per-sequence frame files written by run_hip(dump_dir=...): the frames the DEVICE rendered
        G = np.load(os.path.join(given, "seq_%05d_gray.npy" % seq), mmap_mode="r")
        D = np.load(os.path.join(given, "seq_%05d_depth.npy" % seq), mmap_mode="r")
        given = [(np.ascontiguousarray(G[f]), np.ascontiguousarray(D[f])) for f in range(n_frames)]
    import vio_ct
    P = vio_ct.pkg()
    want_status = bool(cfg_kw.pop("_status", False)) if isinstance(cfg_kw, dict) else False
    cfg = P.canonical_config(**cfg_kw)
    sc = vio_ct.synth_like(cfg)
    o = vio_ct.run_oracle_sequence(cfg, sc, seq, n_frames, tracker_lag=lag, frames=given)
    fr = np.array([x[0] for x in o["traj"]], np.int32)
    po = np.array([x[1] for x in o["traj"]])
    gt = np.array(o["gt"])
    if want_status:   # per-frame decisions for the tests that compare them
        keys = ("solver_flag", "frame_count", "n_landmarks", "marginalization_flag", "n_residuals", "n_in_problem", "n_var_landmarks", "iterations")
        return seq, fr, po, gt, int(o["oracle"].status()["reboot_count"]), np.array([[st[k] for k in keys] for st in o["status"]]), np.array(o["processed"])
    return seq, fr, po, gt, int(o["oracle"].status()["reboot_count"])


def run_oracle_pool(seqs, n_frames, lag=0, cfg_kw=None, procs=None, frames=None):
    """frames: optional {seq: [(gray, depth)] * n_frames} (e.g. downloaded from the device) instead of the host renderer"""
    procs = procs or len(os.sched_getaffinity(0))
    jobs = [(int(s), int(n_frames), int(lag), dict(cfg_kw or {})) + ((frames if isinstance(frames, str) else frames[int(s)],) if frames else ()) for s in seqs]
    with mp.get_context("spawn").Pool(min(procs, len(jobs))) as pool:
        res = pool.map(_oracle_worker, jobs, chunksize=1)
    return {r[0]: r[1:] for r in res}


def run_hip(P, cfg, sc, seq0, S, n_frames, lag=0, chunk=50, check_render=True, per_frame=None, keep=False, grab=None, dump_dir=None):
    # dump_dir: every frame of every sequence as the device rendered it goes to <dump_dir>/seq_%05d_{gray,depth}.npy (memory-mapped arrays
    # [n_frames][H][W]) so that the oracle processes can be fed the IDENTICAL pixels (run_oracle_pool(frames=dump_dir)): the device and the host
    # renderer agree on almost every pixel but not on all of them (float sinf / expf of two math libraries), see DESIGN.md 3 "Round 5"
    # grab: {local sequence index: []} -- filled with the (gray, depth) frames of those sequences as the device rendered them
    """vio_feed over S device-rendered sequences, frames rendered chunk by chunk into one HBM buffer; returns per sequence the
    odometry history rows [stamp, P(3), Q(4), V(3)], the final status, and the wall time of the feed loop"""
    import vio_ct
    syn = P.Synth(sc)
    H, W = cfg.height, cfg.width
    hw = H * W
    g = P.DeviceBuffer(chunk * S * hw)
    d = P.DeviceBuffer(chunk * S * hw * 2)
    times = vio_ct.frame_times(sc, n_frames)
    nimu = int(n_frames / sc.cam_rate * sc.imu_rate) + 64
    b = P.VioBatch(cfg, S, imu_capacity=nimu + 64)
    if lag:
        b.set_tracker_lag(lag)
    imu = [syn.imu(seq0 + s, nimu) for s in range(S)]
    b.push_imu_batch(np.stack([x[0] for x in imu]), np.stack([x[1] for x in imu]), np.stack([x[2] for x in imu]))
    t_feed = 0.0
    for f0 in range(0, n_frames, chunk):
        n = min(chunk, n_frames - f0)
        b.sync()   # the previous chunk's frames are still being read
        for k in range(n):
            syn.render_device(S, seq0, float(times[f0 + k]), g.at(k * S * hw), d.at(k * S * hw * 2))
        if check_render and f0 == 0:
            # the oracle processes render on the host: the two renderers must produce identical pixels
            for (k, i) in ((0, 0), (n - 1, S - 1)):
                gh, dh = syn.render_host(seq0 + i, float(times[k]))
                assert np.array_equal(gh, g.download((k * S + i) * hw, (H, W), np.uint8))
                assert np.array_equal(dh, d.download((k * S + i) * hw * 2, (H, W), np.uint16))
        if dump_dir is not None:
            if f0 == 0:
                os.makedirs(dump_dir, exist_ok=True)
                dump = [(np.lib.format.open_memmap(os.path.join(dump_dir, "seq_%05d_gray.npy" % (seq0 + i)), mode="w+", dtype=np.uint8, shape=(n_frames, H, W)),
                         np.lib.format.open_memmap(os.path.join(dump_dir, "seq_%05d_depth.npy" % (seq0 + i)), mode="w+", dtype=np.uint16, shape=(n_frames, H, W)))
                        for i in range(S)]
            for k in range(n):
                gk = g.download(k * S * hw, (S, H, W), np.uint8)
                dk = d.download(k * S * hw * 2, (S, H, W), np.uint16)
                for i in range(S):
                    dump[i][0][f0 + k] = gk[i]
                    dump[i][1][f0 + k] = dk[i]
            if f0 + n >= n_frames:
                for a_, b_ in dump:
                    a_.flush(); b_.flush()
                dump = None
        if grab is not None:
            for k in range(n):
                for i in grab:
                    grab[i].append((g.download((k * S + i) * hw, (H, W), np.uint8), d.download((k * S + i) * hw * 2, (H, W), np.uint16)))
        c0 = time.perf_counter()
        for k in range(n):
            b.feed(g.at(k * S * hw), d.at(k * S * hw * 2), np.full(S, times[f0 + k]), on_device=True)
            if per_frame is not None:
                per_frame(f0 + k, b)
        b.sync()
        t_feed += time.perf_counter() - c0
    hist = [b.odometry_history(s) for s in range(S)]
    stats = b.status_all()
    g.free(); d.free()
    if keep:
        return hist, stats, t_feed, b
    b.close()
    return hist, stats, t_feed


def compare(hist, orc, seq0, times):
    """per-sequence rows against the oracle: ATEs, maximum distance, first frame the two separate by more than 1e-6 m"""
    import vio_ct
    rows = []
    for i, h in enumerate(hist):
        fr, po, gt, reb = orc[seq0 + i]
        n = min(len(h), len(po))
        same_frames = len(h) == len(po) and np.allclose(h[:, 0], times[fr], atol=1e-9)
        dist = np.linalg.norm(po[:n] - h[:n, 1:4], axis=1)
        sep = np.nonzero(dist > 1e-6)[0]
        rows.append(dict(sequence=seq0 + i, frames=int(n), same_frames=bool(same_frames), oracle_reboots=reb,
                         ate_oracle_m=vio_ct.ate_rmse(po[:n], gt[:n]), ate_hip_m=vio_ct.ate_rmse(h[:n, 1:4], gt[:n]),
                         max_distance_m=float(dist.max()), final_distance_m=float(dist[-1]), early30_max_distance_m=float(dist[:30].max()),
                         first_frame_beyond_1um=(int(fr[sep[0]]) if len(sep) else None)))
    return rows


def summarise(rows):
    ao = np.array([r["ate_oracle_m"] for r in rows]); ah = np.array([r["ate_hip_m"] for r in rows])
    md = np.array([r["max_distance_m"] for r in rows])
    e30 = np.array([r.get("early30_max_distance_m", np.nan) for r in rows])
    fs = [r["first_frame_beyond_1um"] for r in rows if r["first_frame_beyond_1um"] is not None]
    diff = ah - ao
    return dict(sequences=len(rows), early30_max_distance_m=dict(median=float(np.median(e30)), p90=float(np.percentile(e30, 90)), max=float(e30.max()),
                                                                 below_1e_11=int((e30 <= 1e-11).sum()), below_1e_9=int((e30 <= 1e-9).sum())),
                median_first_frame_beyond_1um=(float(np.median(fs)) if fs else None), mean_ate_oracle_m=float(ao.mean()), mean_ate_hip_m=float(ah.mean()),
                rel_diff_of_means=float(abs(ah.mean() - ao.mean()) / ao.mean()), signed_rel_diff_of_means=float((ah.mean() - ao.mean()) / ao.mean()),
                standard_error_of_mean_diff_m=float(diff.std(ddof=1) / np.sqrt(len(diff))) if len(diff) > 1 else None,
                standard_error_rel=float(diff.std(ddof=1) / np.sqrt(len(diff)) / ao.mean()) if len(diff) > 1 else None,
                max_rel_diff_one_sequence=float(np.max(np.abs(diff) / ao)), max_distance_m=float(md.max()), median_max_distance_m=float(np.median(md)),
                separated_beyond_1um=int((md > 1e-6).sum()), separated_beyond_100um=int((md > 1e-4).sum()), separated_beyond_1mm=int((md > 1e-3).sum()),
                identical_to_1um=int((md <= 1e-6).sum()))


def run(P, S=128, seq0=700, n_frames=300, lag=0, modes=("fast", "exact"), procs=None, cfg_kw=None, oracle_devs=(0,), same_frames=False):
    """oracle_devs: OVIO_DEVIATIONS masks of the oracle runs to compare with (0 = the reference's formulation; 15 = every equivalent
    formulation the HIP path uses switched on, oracle/oracle.h ODEV_*: the attribution experiment of round 5)"""
    import vio_ct
    cfg_kw = dict(cfg_kw or {})
    cfg0 = P.canonical_config(**cfg_kw)
    sc = vio_ct.synth_like(cfg0)
    times = vio_ct.frame_times(sc, n_frames)
    out = dict(config=dict(sequences=S, first_sequence=seq0, frames=n_frames, tracker_lag=lag, **cfg_kw), modes={})
    hip = {}
    dump_dir = None
    if same_frames:   # same_frames: the oracle consumes the frames the device rendered (pixel-identical input on both sides)
        import tempfile
        dump_dir = tempfile.mkdtemp(prefix="vio_frames_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    out["config"]["same_frames"] = bool(same_frames)
    for mode in modes:
        cfg = P.canonical_config(marg_exact={"exact": 1, "certified": 2}.get(mode, 0), **cfg_kw)   # fast / exact / certified = vio_config.marg_exact 0 / 1 / 2
        hist, stats, t_feed = run_hip(P, cfg, sc, seq0, S, n_frames, lag=lag, check_render=(mode == modes[0] and not same_frames),
                                      dump_dir=(dump_dir if mode == modes[0] else None))
        hip[mode] = (hist, stats, t_feed)
    if same_frames:
        # how different are the two renderers?  host-render a sample and count the pixels that differ from what the device produced
        syn = P.Synth(sc)
        tms = vio_ct.frame_times(sc, n_frames)
        ng = nd = npx = 0
        for i in range(0, S, max(1, S // 8)):
            G = np.load(os.path.join(dump_dir, "seq_%05d_gray.npy" % (seq0 + i)), mmap_mode="r")
            D = np.load(os.path.join(dump_dir, "seq_%05d_depth.npy" % (seq0 + i)), mmap_mode="r")
            for f in range(0, n_frames, max(1, n_frames // 12)):
                gh, dh = syn.render_host(seq0 + i, float(tms[f]))
                ng += int((gh != G[f]).sum()); nd += int((dh != D[f]).sum()); npx += gh.size
        out["renderer_difference_sample"] = dict(pixels=npx, gray_pixels_differing=ng, depth_pixels_differing=nd)
    c0 = time.perf_counter()
    for dv in oracle_devs:
        if dv:
            os.environ["OVIO_DEVIATIONS"] = str(int(dv))     # inherited by the spawned oracle processes
        else:
            os.environ.pop("OVIO_DEVIATIONS", None)
        try:
            orc = run_oracle_pool(range(seq0, seq0 + S), n_frames, lag=lag, cfg_kw=cfg_kw, procs=procs, frames=dump_dir)
        finally:
            os.environ.pop("OVIO_DEVIATIONS", None)
        for mode in modes:
            hist, stats, t_feed = hip[mode]
            rows = compare(hist, orc, seq0, times)
            out["modes"][mode if not dv else "%s_vs_oracle_dev%d" % (mode, dv)] = dict(
                summary=summarise(rows), hip_feed_wall_s=t_feed, hip_frames_per_s=S * n_frames / t_feed,
                hip_reboots=int(sum(st.reboot_count for st in stats)), oracle_deviations=int(dv), rows=rows)
    out["oracle_wall_s"] = time.perf_counter() - c0
    if dump_dir:
        import shutil
        shutil.rmtree(dump_dir, ignore_errors=True)
    if "fast" in hip and "exact" in hip:
        # the two HIP modes against each other: where deviations 10 / 13 alone move the trajectory
        d = [float(np.linalg.norm(a[:min(len(a), len(b_)), 1:4] - b_[:min(len(a), len(b_)), 1:4], axis=1).max()) for a, b_ in zip(hip["fast"][0], hip["exact"][0])]
        out["fast_vs_exact_max_distance_m"] = dict(max=float(np.max(d)), median=float(np.median(d)), beyond_1e_6=int((np.array(d) > 1e-6).sum()))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", type=int, default=128)
    ap.add_argument("--seq0", type=int, default=700)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--lag", type=int, default=0)
    ap.add_argument("--modes", default="fast,exact")
    ap.add_argument("--procs", type=int, default=0)
    ap.add_argument("--same-frames", type=int, default=0, help="1: the oracle is fed the frames the device rendered (identical pixels on both sides)")
    ap.add_argument("--oracle-devs", default="0", help="comma separated OVIO_DEVIATIONS masks of the oracle runs (0 = reference formulation, 15 = all HIP formulations)")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "gpurun_out", "parity_300_s128.json"))
    a = ap.parse_args()
    import vio_ct
    P = vio_ct.pkg()
    rep = run(P, a.seqs, a.seq0, a.frames, a.lag, tuple(a.modes.split(",")), a.procs or None, oracle_devs=tuple(int(x) for x in a.oracle_devs.split(",")), same_frames=bool(a.same_frames))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(rep, open(a.out, "w"), indent=1)
    for m, v in rep["modes"].items():
        print(m, json.dumps(v["summary"]))
    print("fast vs exact", rep.get("fast_vs_exact_max_distance_m"))
    print("renderer", rep.get("renderer_difference_sample"))


if __name__ == "__main__":
    main()
