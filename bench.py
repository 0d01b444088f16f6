#!/usr/bin/env python
"""bench.py — VIO frames/s of the MI355X-native hot path (BASELINE.json metric).

One "step" = one vio_feed over a batch of S independent synthetic 640x480 RGB-D + 200 Hz IMU sequences resident in HBM
(one camera frame per sequence: readImage with PUB_THIS_FRAME + processImage incl. optimization() and marginalisation).
Workload at N=1: BASELINE configs[2] "batch of 128 independent synthetic sequences on 1 MI355X" (configs[1], S=1, is the
latency case and a parity test); N>1 shards 128 sequences per GPU with no data-path collective (weak scaling).
Frames are rendered on the device outside the timed region; steady state only (after solver_flag == NON_LINEAR).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector == FP64 matrix peak (MI355X_MICROARCH.md has no f64 row)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec


def backend_flops(I, O, F, k_obs, W, S_imu):
    """SURVEY.md §8d algorithmic FP64 flops of one back-end frame (1 FMA = 2 flop)."""
    P = 15 * (W + 1) + 7
    n_p = 6 * W + 16
    m = 15 + F / max(W, 1)
    per_iter = O * 1000 + W * 25000 + O * 1600 + W * 27000 + F * 2 * (6 * k_obs + 7) ** 2 + P ** 3 / 3 + 2 * P ** 2
    return I * per_iter + 2 * n_p ** 3 + (10 * m ** 3 + 2 * (n_p * m ** 2 + n_p ** 2 * m) + 10 * n_p ** 3) + 2 * S_imu * 38000


def frontend_bytes(w, h, n, levels):
    """SURVEY.md §8d algorithmic bytes of one front-end frame."""
    return 2 * w * h + sum(w * h // 4 ** l for l in range(1, levels + 1)) + n * (levels + 1) * 2 * 23 * 23


def aux_rate(P, vio_ct, torch, cfg, sc, dev, S, n_pre, Wm, K):
    """frames/s of the same step at another batch size (auxiliary data point; all 256 CUs busy at S = 256)."""
    H, Wd = cfg.height, cfg.width
    F = n_pre + Wm + K
    syn = P.Synth(sc)
    gray = torch.empty((F, S, H, Wd), dtype=torch.uint8, device=dev)
    depth = torch.empty((F, S, H, Wd), dtype=torch.uint16, device=dev)
    times = vio_ct.frame_times(sc, F)
    for f in range(F):
        syn.render_device(S, 0, float(times[f]), gray[f], depth[f])
    nimu = int(F / sc.cam_rate * sc.imu_rate) + 64
    b = P.VioBatch(cfg, S, imu_capacity=nimu + 64)
    for s in range(S):
        b.push_imu(s, *syn.imu(s, nimu))
    for f in range(n_pre + Wm):
        b.feed(gray[f], depth[f], np.full(S, times[f]), on_device=True)
    b.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        f = n_pre + Wm + k
        b.feed(gray[f], depth[f], np.full(S, times[f]), on_device=True)
    b.sync()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ok = all(b.status(s).solver_flag == 1 for s in range(S))
    b.close()
    return dict(sequences_per_gpu=S, frames_per_s=S * K / el, ms_per_step=el / K * 1e3, valid=bool(ok))


def _cpu_worker(job):
    """One oracle process (spawned, no GPU): replays `n_frames` of sequence `seq` rendered on the host (identical pixels to the device
    renderer) and returns (steady-state seconds inside Pipeline::feed, frames)."""
    seq, n_frames = job
    import vio_ct
    P = vio_ct.pkg()
    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    syn = P.Synth(sc)
    o = vio_ct.OraclePipeline(cfg)
    nimu = int(n_frames / sc.cam_rate * sc.imu_rate) + 64
    o.push_imu(*syn.imu(seq, nimu))
    tcpu, nfr = 0.0, 0
    for f, tf in enumerate(vio_ct.frame_times(sc, n_frames)):
        g, d = syn.render_host(seq, float(tf))
        steady = o.status()["solver_flag"] == 1
        c0 = time.perf_counter()
        r = o.feed(g, d, float(tf))
        c1 = time.perf_counter()
        if steady and r == 1:
            tcpu += c1 - c0
            nfr += 1
    return tcpu, nfr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--seqs", type=int, default=128, help="sequences per GPU")
    ap.add_argument("--cpu-seqs", type=int, default=8, help="sequences replayed through the CPU oracle on rank 0 (0 = skip)")
    ap.add_argument("--cpu-procs", type=int, default=0, help="also time the oracle as one sequence per core on this many processes "
                    "(0 = skip; reported as cpu_baseline_multicore, the headline cpu_baseline stays the 1-core figure)")
    ap.add_argument("--aux", action="store_true", help="also measure S=256 sequences per GPU (reported as aux_s256, never as value)")
    ap.add_argument("--pcie-steps", type=int, default=6, help="extra steps fed from HOST buffers after the timed region (0 = skip)")
    ap.add_argument("--stream-steps", type=int, default=10, help="extra steps with the IMU pushed frame by frame (vio_push_imu per sequence per frame)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # VIO_BENCH_DEVICE / VIO_BENCH_BACKEND exist only to dry-run the N > 1 control flow on a 1-GPU box (both ranks on cuda:0 over gloo)
    if os.environ.get("VIO_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["VIO_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    backend = os.environ.get("VIO_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend)  # "nccl" = RCCL over xGMI; used only for the barrier and the final throughput gather

    import __graft_entry__ as ge
    P = ge.load_package()
    if not os.path.exists(os.path.join(ROOT, "vins-rgbd-fast_amd", "libvio_hip.so")):
        ge.build()
    import vio_ct
    shard = __import__("importlib").import_module("vins-rgbd-fast_amd.shard")

    cfg = P.canonical_config()
    sc = vio_ct.synth_like(cfg)
    S, K, Wm = args.seqs, args.steps, args.warmup
    H, Wd = cfg.height, cfg.width
    n_pre = 16  # first-image skip + init_pub + init_feature + (window_size + 1) frames -> NON_LINEAR, + margin
    Kp = max(args.pcie_steps, 0)
    Ks = max(args.stream_steps, 0)
    F = n_pre + Wm + K + Kp + Ks
    seq0 = shard.sequence_shard(rank, world, S)[0]
    syn = P.Synth(sc)
    dev = torch.device("cuda", local_rank)
    gray = torch.empty((F, S, H, Wd), dtype=torch.uint8, device=dev)
    depth = torch.empty((F, S, H, Wd), dtype=torch.uint16, device=dev)
    times = vio_ct.frame_times(sc, F)
    for f in range(F):
        syn.render_device(S, seq0, float(times[f]), gray[f], depth[f])
    nimu = int(F / sc.cam_rate * sc.imu_rate) + 64
    b = P.VioBatch(cfg, S, imu_capacity=nimu + 64)
    imu_all = [syn.imu(seq0 + s, nimu) for s in range(S)]
    t_stream0 = times[F - Ks] if Ks > 0 else 1e300   # IMU up to (and one sample past) the last non-streaming frame goes in up front
    imu_k = []
    for s in range(S):
        ti, ai, gi = imu_all[s]
        k = vio_ct.imu_until(ti, 0, times[F - Ks - 1], sc.imu_rate) if Ks > 0 else len(ti)
        b.push_imu(s, ti[:k], ai[:k], gi[:k])
        imu_k.append(k)

    def feed(f):
        b.feed(gray[f], depth[f], np.full(S, times[f]), on_device=True)

    for f in range(n_pre + Wm):
        feed(f)
    b.sync()
    fp0 = np.array([b.status(s).frames_processed for s in range(S)])
    nl = np.array([b.status(s).solver_flag for s in range(S)])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    b.profile_begin(K)
    t0 = time.perf_counter()
    for k in range(K):
        feed(n_pre + Wm + k)
    b.sync()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if world > 1:
        dist.barrier()
    nprof, kms = b.profile_end()
    elapsed_local = t1 - t0

    # ---- PCIe-inclusive rate (never `value`): the same call handed pageable HOST buffers, S x 0.92 MB uploaded per step
    pcie = None
    if Kp > 0:
        hg = [gray[n_pre + Wm + K + k].cpu().numpy() for k in range(Kp)]
        hd = [depth[n_pre + Wm + K + k].cpu().numpy() for k in range(Kp)]
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for k in range(Kp):
            b.feed(hg[k], hd[k], np.full(S, times[n_pre + Wm + K + k]), on_device=False)
        b.sync()
        c1 = time.perf_counter()
        pcie = dict(frames_per_s=S * Kp / (c1 - c0), ms_per_step=(c1 - c0) / Kp * 1e3, steps=Kp,
                    note="vio_feed(on_device=0): pageable numpy buffers, %.1f MB uploaded per step" % (S * H * Wd * 3 / 1e6))

    # ---- streaming leg (never `value`): IMU arrives between frames, 20 samples per sequence pushed through vio_push_imu before
    # each vio_feed, device-resident images -- the call pattern of the reference's callbacks
    stream = None
    if Ks > 0:
        f0 = F - Ks
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for k in range(Ks):
            f = f0 + k
            for s in range(S):
                ti, ai, gi = imu_all[s]
                k2 = vio_ct.imu_until(ti, imu_k[s], times[f], sc.imu_rate)
                if k2 > imu_k[s]:
                    b.push_imu(s, ti[imu_k[s]:k2], ai[imu_k[s]:k2], gi[imu_k[s]:k2])
                    imu_k[s] = k2
            feed(f)
        b.sync()
        c1 = time.perf_counter()
        stream = dict(frames_per_s=S * Ks / (c1 - c0), ms_per_step=(c1 - c0) / Ks * 1e3, steps=Ks,
                      note="IMU pushed per sequence per frame from Python (host time of %d ctypes calls per step included)" % S)

    # ---- validity + accuracy (outside the timed region)
    stats = [b.status(s) for s in range(S)]
    fp1 = np.array([st.frames_processed for st in stats])
    all_processed = bool(np.all(fp1 - fp0 == K + Kp + Ks) and np.all(nl == 1))
    ates = []
    hist = {}
    for s in range(S):
        h = b.odometry_history(s)
        hist[s] = h
        if len(h) < 5:
            continue
        gt = np.array([syn.pose(seq0 + s, float(t))[0] for t in h[:, 0]])
        ates.append(vio_ct.ate_rmse(h[:, 1:4], gt))
    sq_err = float(np.sum([a * a for a in ates]))
    total_frames, elapsed, sq_err_all, n_ate_all = shard.job_totals(S * K, elapsed_local, sq_err, len(ates), device=dev if backend == "nccl" else None)
    worst = int(np.argmax(ates)) if ates else -1
    iters = float(np.mean([st.iterations for st in stats]))
    nres = float(np.mean([st.n_residuals for st in stats]))
    nvar = float(np.mean([st.n_var_landmarks for st in stats]))
    ninp = float(np.mean([st.n_in_problem for st in stats]))
    ntrk = float(np.mean([st.n_tracks for st in stats]))
    reboots = int(np.sum([st.reboot_count for st in stats]))

    # ---- roofline of the dominant kernel (HIP events on the batch stream, averaged over the timed steps)
    dom = max(kms, key=lambda k: kms[k])
    k_obs = nres / max(ninp, 1.0) + 1.0
    be_flops_seq = backend_flops(max(iters, 1.0), nres, nvar, k_obs, cfg.window_size, sc.imu_rate / sc.cam_rate)
    fe_bytes_seq = frontend_bytes(Wd, H, cfg.max_cnt, cfg.lk_max_level)
    be_ms = sum(v for k, v in kms.items() if k.startswith("be_"))
    fe_ms = sum(v for k, v in kms.items() if k.startswith("fe_"))
    ksym = dom + "_kernel"
    if dom == "be_solve" and int(os.environ.get("VIO_BE_THREADS", "512")) <= 512:
        ksym = "be_solve_kernel_512"  # the 512-thread build of the solve kernel (256 VGPRs per lane) is the default
    if dom.startswith("be_"):
        ach = be_flops_seq * S / (kms[dom] * 1e-3) / 1e12
        roof = dict(bound="mfma", kernel=ksym, achieved=ach, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / FP64_PEAK_TFLOPS,
                    traffic=None, ms=kms[dom],
                    note="FP64 path; algorithmic back-end flops per launch (SURVEY.md 8d with measured I, O, F) / average launch duration; "
                         "peak = FP64 vector/matrix 78.6 TFLOP/s")
    else:
        ach = fe_bytes_seq * S / (kms[dom] * 1e-3) / 1e9
        roof = dict(bound="hbm", kernel=ksym, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=None,
                    ms=kms[dom], note="algorithmic front-end bytes per launch (SURVEY.md 8d) / average launch duration")
    roof["frontend_GBps"] = fe_bytes_seq * S / (fe_ms * 1e-3) / 1e9 if fe_ms > 0 else None
    roof["backend_TFLOPs"] = be_flops_seq * S / (be_ms * 1e-3) / 1e12 if be_ms > 0 else None

    # ---- CPU baseline: the oracle (port of the reference algorithm, 1 thread) on the same rendered frames, rank 0 only
    cpu = None
    parity = None
    if rank == 0 and args.cpu_seqs > 0:
        ncs = min(args.cpu_seqs, S)
        tcpu, nfr, rm, ate_pairs = 0.0, 0, [], []
        for s in range(ncs):
            o = vio_ct.OraclePipeline(cfg)
            ti, ai, gi = syn.imu(seq0 + s, nimu)
            o.push_imu(ti, ai, gi)
            traj = []
            for f in range(F):
                g = gray[f, s].cpu().numpy()
                d = depth[f, s].cpu().numpy()
                steady = o.status()["solver_flag"] == 1
                c0 = time.perf_counter()
                r = o.feed(g, d, float(times[f]))
                c1 = time.perf_counter()
                if steady and r == 1:
                    tcpu += c1 - c0
                    nfr += 1
                if o.status()["solver_flag"] == 1 and r == 1:
                    traj.append(o.window()[cfg.window_size, :3].copy())
            hh = hist[s]
            m = min(len(traj), len(hh))
            if m > 0:
                rm.append(float(np.sqrt(((np.array(traj[:m]) - hh[:m, 1:4]) ** 2).sum(1).mean())))
            if m >= 5:
                gt = np.array([syn.pose(seq0 + s, float(t))[0] for t in hh[:m, 0]])
                ate_pairs.append((vio_ct.ate_rmse(hh[:m, 1:4], gt), vio_ct.ate_rmse(np.array(traj[:m]), gt)))
        cpu = dict(value=nfr / tcpu if tcpu > 0 else None, unit="frames/s", cores=1, kind="port",
                   sample="%d sequences x %d steady-state frames of the same rendered workload through oracle/ (-O3, 1 thread; "
                          "the reference binary needs ROS/OpenCV/Ceres and cannot be built here)" % (ncs, nfr // max(ncs, 1)),
                   cpu_seconds=tcpu)
        ah = float(np.mean([a for a, _ in ate_pairs])) if ate_pairs else None
        ao = float(np.mean([b for _, b in ate_pairs])) if ate_pairs else None
        parity = dict(traj_rmse_hip_vs_oracle_m=float(np.max(rm)) if rm else None, sequences=ncs, per_sequence=rm,
                      ate_hip_m=ah, ate_oracle_m=ao, ate_rel_diff=(abs(ah - ao) / ao if ao else None),
                      note="north-star tolerance: ATE of the HIP path within 1 % of the reference algorithm (oracle) on identical input")

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "round1_pmc_traffic.json")
    if os.path.exists(tpath):  # written by profiles/collect.sh from separate rocprofv3 --pmc passes over this same command
        tj = json.load(open(tpath))
        ent = next((v for k, v in tj.get("kernels", {}).items() if k.startswith(dom)), None)  # be_solve -> be_solve_kernel_512
        if ent and tj.get("sequences_per_gpu") == S:
            traffic = ent["hbm_bytes_per_launch"]
            roof["traffic_source"] = "profiles/round1_pmc_traffic.json (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction)"
    roof["traffic"] = traffic
    out = {
        "metric": "VIO frames/sec (640x480, 150 feats, 10-KF window)",
        "value": total_frames / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": K,
        "warmup": Wm,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]: batch of %d independent synthetic 640x480 RGB-D + 200 Hz IMU sequences per MI355X, "
                               "150 max features, 5x6 grid, 10-keyframe window, landmarks free (fix_depth 0)" % S,
                   "sequences_per_gpu": S, "image": [Wd, H], "max_cnt": cfg.max_cnt, "window_size": cfg.window_size,
                   "parallelism": "independent sequences sharded per GPU, no data-path collective"},
        "valid": all_processed,
        "ate_m": {"mean": float(np.mean(ates)) if ates else None, "max": float(np.max(ates)) if ates else None, "sequences": len(ates),
                  "median": float(np.median(ates)) if ates else None, "worst_sequence": seq0 + worst,
                  "job_rms": shard.ate_from_sums(sq_err_all, n_ate_all)},
        "solver": {"mean_iterations": iters, "mean_residuals": nres, "mean_var_landmarks": nvar, "mean_tracks": ntrk, "reboots": reboots},
        "kernels_ms": kms,
        "frontend_ms": fe_ms,
        "backend_ms": be_ms,
        "roofline": roof,
        "cpu_baseline": cpu,
        "parity": parity,
        "pcie_inclusive": pcie,
        "imu_streaming": stream,
    }
    if rank == 0 and args.cpu_procs > 1:
        import multiprocessing as mp
        nproc = min(args.cpu_procs, os.cpu_count() or 1)
        with mp.get_context("spawn").Pool(nproc) as pool:
            res = pool.map(_cpu_worker, [(seq0 + 1000 + i, 46) for i in range(nproc)])
        out["cpu_baseline_multicore"] = dict(value=float(sum(n / t for t, n in res if t > 0)), unit="frames/s", cores=nproc, kind="port",
                                             sample="%d oracle processes, one sequence each, 46 frames; sum of the per-process steady-state rates" % nproc)
    if args.aux and rank == 0:
        out["aux_s256"] = aux_rate(P, vio_ct, torch, cfg, sc, dev, 256, n_pre, Wm, K)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
