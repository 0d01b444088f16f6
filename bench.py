#!/usr/bin/env python
"""bench.py — VIO frames/s of the MI355X-native hot path (BASELINE.json metric).

One "step" = one vio_feed over a batch of S independent synthetic 640x480 RGB-D + 200 Hz IMU sequences resident in HBM
(one camera frame per sequence: readImage with PUB_THIS_FRAME + processImage incl. optimization() and marginalisation).
Workload at N=1: BASELINE configs[2] "batch of 128 independent synthetic sequences on 1 MI355X" (configs[1], S=1, is the
latency case and a parity test); N>1 shards 128 sequences per GPU with no data-path collective (weak scaling).
Frames are rendered on the device outside the timed region; steady state only (after solver_flag == NON_LINEAR).

    python bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no torchrun environment re-executes itself under `python -m torch.distributed.run` with N ranks (one per
GPU) and fails loudly if fewer than N devices are visible; under torchrun (WORLD_SIZE set) N must equal WORLD_SIZE.
The timed region is K steps bracketed by barrier + synchronize; it is repeated `--repeats` times on fresh frames and `value` is the
MEDIAN repeat (all repeats are listed).  Prints ONE JSON line on rank 0 with `roofline` (dominant kernel), `roofline_kernels`
(the next heaviest kernels) and `cpu_baseline` objects.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

# Runtime knobs of the product, set before HIP initialises: two stream groups per handle (the phased solver's serial
# kernel of one half-batch overlaps the data-parallel kernels of the other) need more hardware queues than the runtime's default 4.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP64_PEAK_TFLOPS = 78.6   # MI355X FP64 vector == FP64 matrix peak (MI355X_MICROARCH.md has no f64 row)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec


def backend_flops(I, O, F, k_obs, W, S_imu):
    """SURVEY.md §8d algorithmic FP64 flops of one back-end frame (1 FMA = 2 flop), split by the kernel that does the work:
    solve  = I x [projection eval + IMU eval + local J^T J (2x20, 15x30) + landmark Schur + dense Cholesky + back-substitution]
    marg   = prior J^T J + marginalisation Schur; the survey's 10 n_p^3 eigen-decomposition term is listed separately because the
             hot path does not perform it (prior kept as a quadratic form, DESIGN.md deviation 13) -- it is NOT credited to any kernel
    ingest = pre-integration of the new frame's IMU samples."""
    P = 15 * (W + 1) + 7
    n_p = 6 * W + 16
    m = 15 + F / max(W, 1)
    per_iter = O * 1000 + W * 25000 + O * 1600 + W * 27000 + F * 2 * (6 * k_obs + 7) ** 2 + P ** 3 / 3 + 2 * P ** 2
    return dict(solve=I * per_iter, marg=2 * n_p ** 3 + 10 * m ** 3 + 2 * (n_p * m ** 2 + n_p ** 2 * m), eig_not_done=10 * n_p ** 3,
                ingest=2 * S_imu * 38000)


def frontend_bytes(w, h, n, levels):
    """SURVEY.md §8d algorithmic bytes of one front-end frame, split by kernel: pyramid (read frame + write levels), FAST (read
    frame), LK (prev + next 23x23 u8 patches per level per feature)."""
    return dict(pyrdown=w * h + sum(w * h // 4 ** l for l in range(1, levels + 1)), fast=w * h, lk=n * (levels + 1) * 2 * 23 * 23)


def config5(P):
    """BASELINE configs[4]: 1280x720, 300 features, 7x8 grid, 20-keyframe window (intrinsics of the canonical camera scaled x2 / x1.5)"""
    return P.canonical_config(width=1280, height=720, max_cnt=300, window_size=20, grid_rows=7, grid_cols=8, max_landmarks=2048,
                              fx=604.5821781259577 * 2, fy=604.2544712985845 * 1.5, cx=321.2638233484251 * 2, cy=239.70969315130674 * 1.5)


def aux_rate(P, vio_ct, torch, cfg, sc, dev, S, n_pre, Wm, K, lag=0, seq0=0):
    """frames/s of the same step at another batch size / tracker ordering / configuration (auxiliary data point, never `value`)."""
    H, Wd = cfg.height, cfg.width
    F = n_pre + Wm + K
    syn = P.Synth(sc)
    gray = torch.empty((F, S, H, Wd), dtype=torch.uint8, device=dev)
    depth = torch.empty((F, S, H, Wd), dtype=torch.uint16, device=dev)
    times = vio_ct.frame_times(sc, F)
    for f in range(F):
        syn.render_device(S, seq0, float(times[f]), gray[f], depth[f])
    nimu = int(F / sc.cam_rate * sc.imu_rate) + 64
    b = P.VioBatch(cfg, S, imu_capacity=nimu + 64)
    b.set_tracker_lag(lag)
    imu = [syn.imu(seq0 + s, nimu) for s in range(S)]
    b.push_imu_batch(np.stack([x[0] for x in imu]), np.stack([x[1] for x in imu]), np.stack([x[2] for x in imu]))
    for f in range(n_pre + Wm):
        b.feed(gray[f], depth[f], np.full(S, times[f]), on_device=True)
    b.sync()
    torch.cuda.synchronize()
    b.profile_begin(K)
    t0 = time.perf_counter()
    for k in range(K):
        f = n_pre + Wm + k
        b.feed(gray[f], depth[f], np.full(S, times[f]), on_device=True)
    b.sync()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    _, kms = b.profile_end()
    stats = b.status_all()
    ok = all(st.solver_flag == 1 for st in stats) and float(np.mean([st.iterations for st in stats])) >= 3.0   # (healthy solves of this workload take 4 - 8 iterations)
    sol = dict(mean_residuals=float(np.mean([st.n_residuals for st in stats])), mean_var_landmarks=float(np.mean([st.n_var_landmarks for st in stats])),
               mean_in_problem=float(np.mean([st.n_in_problem for st in stats])), mean_iterations_last_frame=float(np.mean([st.iterations for st in stats])),
               reboots=int(sum(st.reboot_count for st in stats)), overflow_frames=int(sum(st.overflow_frames for st in stats)))
    b.close()
    del gray, depth
    torch.cuda.empty_cache()
    return dict(sequences_per_gpu=S, tracker_lag=lag, frames_per_s=S * K / el, ms_per_step=el / K * 1e3, steps=K, valid=bool(ok),
                kernels_ms={k: round(v, 4) for k, v in kms.items()}, solver=sol)


_CPU_BARRIER = None


def _cpu_init(barrier):
    global _CPU_BARRIER
    _CPU_BARRIER = barrier


def _cpu_worker(job):
    """One oracle process (spawned; loads ONLY oracle/liboracle.so -- neither the HIP library nor a HIP runtime): the frames of its sequence
    were rendered by the parent into /dev/shm and are read into memory first, then every worker waits at a barrier so that all of them run
    their steady-state frames at the same time; returns (steady-state seconds inside Pipeline::feed, frames)."""
    path, cfg_bytes = job
    import importlib
    import threading
    import vio_ct
    P = importlib.import_module("vins-rgbd-fast_amd")   # the module only (ctypes struct definitions); lib() is never called here
    cfg = P.Config.from_buffer_copy(cfg_bytes)
    z = np.load(path)
    gray, depth, times, it, ia, ig = z["gray"], z["depth"], z["times"], z["imu_t"], z["imu_acc"], z["imu_gyr"]
    o = vio_ct.OraclePipeline(cfg)
    o.push_imu(it, ia, ig)
    first = True
    tcpu, nfr = 0.0, 0
    for f in range(len(times)):
        steady = o.status()["solver_flag"] == 1
        if steady and first:
            first = False
            if _CPU_BARRIER is not None:
                try:
                    _CPU_BARRIER.wait(timeout=300)      # (the initialisation frames are behind every worker: the timed frames of all workers overlap)
                except threading.BrokenBarrierError:
                    pass
        c0 = time.perf_counter()
        r = o.feed(np.ascontiguousarray(gray[f]), np.ascontiguousarray(depth[f]), float(times[f]))
        c1 = time.perf_counter()
        if steady and r == 1:
            tcpu += c1 - c0
            nfr += 1
    return tcpu, nfr


def effective_cores():
    """(cores this process can actually use, how that was determined): the scheduler affinity capped by the cgroup CPU quota -- the GPU
    boxes expose 256 hardware threads to a container whose cpu.max is 16 CPUs"""
    aff = len(os.sched_getaffinity(0))
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    if quota is None:
        return aff, "sched_getaffinity (%d); no cgroup CPU quota" % aff
    return max(1, min(aff, int(quota + 0.5))), "min(sched_getaffinity = %d, cgroup cpu.max = %.1f CPUs)" % (aff, quota)


def physical_cores():
    """distinct (package, core) pairs of the hardware threads this process may run on (Linux sysfs); None when not readable"""
    try:
        seen = set()
        for cpu in os.sched_getaffinity(0):
            base = "/sys/devices/system/cpu/cpu%d/topology/" % cpu
            seen.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        return len(seen)
    except OSError:
        return None


def device_identity(torch, index):
    """something that tells two physical devices apart: uuid when the runtime exposes it, PCI bus id otherwise"""
    pr = torch.cuda.get_device_properties(index)
    ident = dict(index=int(index), name=pr.name)
    for k in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id"):
        v = getattr(pr, k, None)
        if v is not None:
            ident[k] = str(v)
    return ident


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=10, help="the K-step timed region is run this many times (fresh frames); value = median. "
                    "The default gives 10 x 20 = 200 timed frames per sequence under the driver's --steps 20 (SURVEY.md 8d asks for >= 200)")
    ap.add_argument("--seqs", type=int, default=128, help="sequences per GPU")
    ap.add_argument("--cpu-seqs", type=int, default=8, help="sequences replayed through the CPU oracle on rank 0 (0 = skip)")
    ap.add_argument("--cpu-procs", type=int, default=-1, help="oracle processes for the all-core CPU baseline, one sequence each "
                    "(-1 = every core this process may run on, 0 = skip)")
    ap.add_argument("--aux", type=int, default=1, help="1 (default): also measure S = 256 and S = 512 sequences per GPU and the lag-0 ordering "
                    "at S (reported as aux_s256 / aux_s512 / lag0, never as value); 0 = skip")
    ap.add_argument("--seq-offset", type=int, default=0, help="global id of rank 0's first sequence (replays another rank's shard on one GPU)")
    ap.add_argument("--dump", default="", help="write the final windows / odometry rows of this rank's sequences to <dump>.rank<r>.npz")
    ap.add_argument("--tracker-lag", type=int, default=1, choices=[0, 1], help="vio_set_tracker_lag: 1 = the tracker of frame f+1 overlaps the "
                    "optimisation of frame f (the reference's two threads with the estimator one frame behind), 0 = it waits for it")
    ap.add_argument("--pcie-steps", type=int, default=20, help="extra steps fed from HOST buffers after the timed region (0 = skip)")
    ap.add_argument("--stream-steps", type=int, default=10, help="extra steps with the IMU pushed frame by frame")
    args = ap.parse_args()

    # ---- N ranks: spawn them ourselves when the caller did not (python bench.py --gpus N), never silently run fewer
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus and os.environ.get("VIO_BENCH_DEVICE") is None:
            raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible" % (args.gpus, have))
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d (launch one rank per GPU)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # VIO_BENCH_DEVICE / VIO_BENCH_BACKEND exist only to dry-run the N > 1 control flow on a 1-GPU box (both ranks on cuda:0 over gloo)
    if os.environ.get("VIO_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["VIO_BENCH_DEVICE"])
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no device (%d visible)" % (local_rank, torch.cuda.device_count()))
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
    S, K, Wm, R = args.seqs, args.steps, args.warmup, max(args.repeats, 1)
    if S >= 64 and S % 2 == 0:
        os.environ.setdefault("VIO_GROUP_SEQS", str(S // 2))   # read by vio_create
    H, Wd = cfg.height, cfg.width
    n_pre = 16  # first-image skip + init_pub + init_feature + (window_size + 1) frames -> NON_LINEAR, + margin
    Kp = max(args.pcie_steps, 0)
    Ks = max(args.stream_steps, 0)
    F = n_pre + Wm + R * K + 2 * Kp + 2 * Ks   # (the PCIe leg runs twice: pageable and page-locked buffers)
    seq0 = args.seq_offset + shard.sequence_shard(rank, world, S)[0]
    syn = P.Synth(sc)
    dev = torch.device("cuda", local_rank)
    gray = torch.empty((F, S, H, Wd), dtype=torch.uint8, device=dev)
    depth = torch.empty((F, S, H, Wd), dtype=torch.uint16, device=dev)
    times = vio_ct.frame_times(sc, F)
    for f in range(F):
        syn.render_device(S, seq0, float(times[f]), gray[f], depth[f])
    nimu = int(F / sc.cam_rate * sc.imu_rate) + 64
    b = P.VioBatch(cfg, S, imu_capacity=nimu + 64)
    b.set_tracker_lag(args.tracker_lag)
    imu_all = [syn.imu(seq0 + s, nimu) for s in range(S)]
    f_stream0 = F - 2 * Ks                                   # first frame of the streaming legs
    # IMU up to (and one sample past) the last non-streaming frame goes in up front, in one batched call
    k_up = vio_ct.imu_until(imu_all[0][0], 0, times[f_stream0 - 1], sc.imu_rate) if Ks > 0 else nimu
    b.push_imu_batch(np.stack([x[0][:k_up] for x in imu_all]), np.stack([x[1][:k_up] for x in imu_all]), np.stack([x[2][:k_up] for x in imu_all]))
    imu_k = [k_up] * S

    def feed(f):
        b.feed(gray[f], depth[f], np.full(S, times[f]), on_device=True)

    for f in range(n_pre + Wm):
        feed(f)
    b.sync()
    st0 = b.status_all()
    fp0 = np.array([st.frames_processed for st in st0])
    nl = np.array([st.solver_flag for st in st0])
    it0 = np.array([(st.iterations_total, st.solves_total) for st in st0], np.int64)
    b.profile_begin(R * K)
    elapsed_rep = []
    for r in range(R):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            feed(n_pre + Wm + r * K + k)
        b.sync()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if world > 1:
            dist.barrier()
        elapsed_rep.append(t1 - t0)
    nprof, kms = b.profile_end()
    st1 = b.status_all()
    it1 = np.array([(st.iterations_total, st.solves_total) for st in st1], np.int64)
    d_it = (it1 - it0).sum(0)
    iters = float(d_it[0]) / max(float(d_it[1]), 1.0)          # mean solver iterations per solve over ALL timed steps and sequences
    f_timed_end = n_pre + Wm + R * K

    # ---- PCIe-inclusive rate (never `value`): the same call handed pageable HOST buffers, S x 0.92 MB uploaded per step
    pcie = None
    if Kp > 0:
        hg = [gray[f_timed_end + k].cpu().numpy() for k in range(Kp)]
        hd = [depth[f_timed_end + k].cpu().numpy() for k in range(Kp)]
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for k in range(Kp):
            b.feed(hg[k], hd[k], np.full(S, times[f_timed_end + k]), on_device=False)
        b.sync()
        c1 = time.perf_counter()
        pcie = dict(frames_per_s=S * Kp / (c1 - c0), ms_per_step=(c1 - c0) / Kp * 1e3, steps=Kp,
                    note="vio_feed(on_device=0): pageable numpy buffers, %.1f MB uploaded per step" % (S * H * Wd * 3 / 1e6))
        # the same from page-locked buffers (vio_host_alloc)
        pg = [P.PinnedArray((S, H, Wd), np.uint8) for _ in range(Kp)]
        pd = [P.PinnedArray((S, H, Wd), np.uint16) for _ in range(Kp)]
        for k in range(Kp):
            pg[k].a[...] = gray[f_timed_end + Kp + k].cpu().numpy(); pd[k].a[...] = depth[f_timed_end + Kp + k].cpu().numpy()
        # every buffer is DMA-ed once before the timed loop: the FIRST transfer out of a fresh page-locked allocation runs at a third of the
        # link rate (tools/h2d_pattern.py: 17.6 GB/s first use, 56 GB/s afterwards), and a deployment rotates a few long-lived image sets --
        # rounds 3 - 5 timed 20 first uses and reported page-locked uploads as slower than pageable ones for that reason
        warm_g = torch.empty((S, H, Wd), dtype=torch.uint8, device=dev); warm_d = torch.empty((S, H, Wd), dtype=torch.int16, device=dev)
        for k in range(Kp):
            warm_g.copy_(torch.from_numpy(pg[k].a), non_blocking=True); warm_d.copy_(torch.from_numpy(pd[k].a.view(np.int16)), non_blocking=True)
        torch.cuda.synchronize()
        del warm_g, warm_d
        b.sync()
        c0 = time.perf_counter()
        for k in range(Kp):
            b.feed(pg[k].a, pd[k].a, np.full(S, times[f_timed_end + Kp + k]), on_device=False)
        b.sync()
        c1 = time.perf_counter()
        pcie["pinned"] = dict(frames_per_s=S * Kp / (c1 - c0), ms_per_step=(c1 - c0) / Kp * 1e3,
                              note="the next %d frames from vio_host_alloc (page-locked) buffers, each DMA-ed once before the timed loop" % Kp)
        for x in pg + pd:
            x.free()

    # ---- streaming legs (never `value`): IMU arrives between frames, device-resident images.  (a) the reference's callback pattern:
    # one vio_push_imu per sequence per frame from Python; (b) one vio_push_imu_batch per frame (SoA over sequences)
    stream = None
    if Ks > 0:
        legs = {}
        for name, f0 in (("per_sequence_calls", f_stream0), ("batched_call", f_stream0 + Ks)):
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            for k in range(Ks):
                f = f0 + k
                k2 = [vio_ct.imu_until(imu_all[s][0], imu_k[s], times[f], sc.imu_rate) for s in range(S)]
                if name == "per_sequence_calls":
                    for s in range(S):
                        ti, ai, gi = imu_all[s]
                        if k2[s] > imu_k[s]:
                            b.push_imu(s, ti[imu_k[s]:k2[s]], ai[imu_k[s]:k2[s]], gi[imu_k[s]:k2[s]])
                else:
                    m = max(max(k2[s] - imu_k[s] for s in range(S)), 1)
                    tt, aa, gg = np.zeros((S, m)), np.zeros((S, m, 3)), np.zeros((S, m, 3))
                    for s in range(S):
                        q = k2[s] - imu_k[s]
                        tt[s, :q] = imu_all[s][0][imu_k[s]:k2[s]]; aa[s, :q] = imu_all[s][1][imu_k[s]:k2[s]]; gg[s, :q] = imu_all[s][2][imu_k[s]:k2[s]]
                    b.push_imu_batch(tt, aa, gg, n=[k2[s] - imu_k[s] for s in range(S)])
                imu_k = k2
                feed(f)
            b.sync()
            c1 = time.perf_counter()
            legs[name] = dict(frames_per_s=S * Ks / (c1 - c0), ms_per_step=(c1 - c0) / Ks * 1e3, steps=Ks)
        stream = dict(legs, note="20 IMU samples per sequence pushed before every frame (host time of the pushes included): "
                                 "%d vio_push_imu calls per step vs one vio_push_imu_batch call" % S)

    # ---- validity + accuracy (outside the timed region)
    stats = b.status_all()
    fp1 = np.array([st.frames_processed for st in stats])
    all_processed = bool(np.all(fp1 - fp0 == R * K + 2 * Kp + 2 * Ks) and np.all(nl == 1))   # timed steps + both PCIe legs + both streaming legs
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
    job = [shard.job_totals(S * K, el, sq_err, len(ates), device=dev if backend == "nccl" else None) for el in elapsed_rep]
    rates = [tf / el for (tf, el, _, _) in job]               # frames of ALL ranks / MAX-over-ranks time, per repeat
    order = int(np.argsort(rates)[len(rates) // 2])
    # per-rank table (N > 1): who ran where and how fast -- a slow or a duplicated device shows up in the record
    mine = dict(rank=rank, device=device_identity(torch, local_rank), host=socket.gethostname(), first_sequence=int(seq0), sequences=S,
                frames_per_s=float(S * K / elapsed_rep[order]), ms_per_step=float(elapsed_rep[order] / K * 1e3),
                ate_rms_m=(float(np.sqrt(sq_err / len(ates))) if ates else None),
                # (round 6: `valid` also demands a sane trajectory error -- an ATE of metres on this workload means the estimator computed garbage
                # fast, which is how a broken 1024-thread ps_serial build went unnoticed at "60 k frames/s"; healthy runs sit at 15 - 17 mm)
                valid=bool(all_processed and (not ates or float(np.sqrt(sq_err / len(ates))) < 0.1)))
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        ids = [json.dumps({k: v for k, v in r["device"].items() if k != "index"} if any(k in r["device"] for k in ("uuid", "pci_bus_id")) else r["device"],
                          sort_keys=True) + r["host"] for r in per_rank]
        if len(set(ids)) != world and os.environ.get("VIO_BENCH_DEVICE") is None:
            raise SystemExit("bench.py --gpus %d: ranks share a device: %s" % (world, ids))
    all_valid = all(r["valid"] for r in per_rank)
    total_frames, elapsed, sq_err_all, n_ate_all = job[order]
    worst = int(np.argmax(ates)) if ates else -1
    nres = float(np.mean([st.n_residuals for st in stats]))
    nvar = float(np.mean([st.n_var_landmarks for st in stats]))
    ninp = float(np.mean([st.n_in_problem for st in stats]))
    ntrk = float(np.mean([st.n_tracks for st in stats]))
    reboots = int(np.sum([st.reboot_count for st in stats]))

    # ---- rooflines: algorithmic work of ONE kernel per launch (S sequences) / its average launch duration (HIP events on the
    # handle's own streams, averaged over the R x K timed steps)
    k_obs = nres / max(ninp, 1.0) + 1.0
    flops = backend_flops(max(iters, 1.0), nres, nvar, k_obs, cfg.window_size, sc.imu_rate / sc.cam_rate)
    fbytes = frontend_bytes(Wd, H, cfg.max_cnt, cfg.lk_max_level)
    be_ms = sum(v for k, v in kms.items() if k.startswith("be_"))
    fe_ms = sum(v for k, v in kms.items() if k.startswith("fe_"))
    tj = None
    for name in ("round6_pmc_traffic.json", "round5_pmc_traffic.json", "round4_pmc_traffic.json", "round3_pmc_traffic.json", "round2_pmc_traffic.json", "round1_pmc_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", name)
        if os.path.exists(tpath):  # written by profiles/collect.sh from separate rocprofv3 --pmc passes over this same command
            tj = (name, json.load(open(tpath)))
            break

    def traffic_of(kname):
        if not tj or tj[1].get("sequences_per_gpu") != S or tj[1].get("sequences_per_launch", S) != S_launch:
            return None
        kk = "be_solve_phased" if (kname == "be_solve" and os.environ.get("VIO_SOLVE_MODE", "1") != "0") else kname
        ent = next((v for k, v in tj[1].get("kernels", {}).items() if k.startswith(kk)), None)
        return ent["hbm_bytes_per_launch"] if ent else None

    # the HIP events bracket the launches of stream group 0, which cover S / n_groups sequences each
    per_group = int(os.environ.get("VIO_GROUP_SEQS", S))
    n_groups = max(1, -(-S // max(per_group, 1)))
    S_launch = min(S, per_group)

    def roof_flops(kname, sym, fl, note):
        ach = fl * S_launch / (kms[kname] * 1e-3) / 1e12
        return dict(bound="mfma", kernel=sym, achieved=ach, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / FP64_PEAK_TFLOPS, traffic=traffic_of(kname),
                    ms=kms[kname], algorithmic_flops_per_launch=fl * S_launch, sequences_per_launch=S_launch, note=note)

    def roof_bytes(kname, sym, by, note):
        ach = by * S_launch / (kms[kname] * 1e-3) / 1e9
        tr = traffic_of(kname)
        return dict(bound="hbm", kernel=sym, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=tr, ms=kms[kname],
                    algorithmic_bytes_per_launch=by * S_launch, sequences_per_launch=S_launch,
                    traffic_over_algorithmic=(tr / (by * S_launch) if tr else None), note=note)

    phased = os.environ.get("VIO_SOLVE_MODE", "1") != "0"
    solve_sym = ("ps_* (phased solver: ps_setup, slots of ps_eval / ps_asm_a / ps_asm_b / ps_schur / ps_serial, ps_final)" if phased else
                 ("be_solve_kernel_512" if int(os.environ.get("VIO_BE_THREADS", "512")) <= 512 else "be_solve_kernel"))
    rk = {
        "be_solve": roof_flops("be_solve", solve_sym, flops["solve"],
                               "FP64; solve-only algorithmic flops per launch = S x I x per-iteration flops (SURVEY.md 8d, measured I = mean "
                               "iterations over all timed solves, O residuals, F variable landmarks) / average launch duration"),
        "be_marg": roof_flops("be_marg", "be_marg_kernel", flops["marg"],
                              "FP64; marginalisation flops of SURVEY.md 8d without the 10 n_p^3 eigen-decomposition the hot path does not perform"),
        "be_ingest": roof_flops("be_ingest", "be_ingest_kernel", flops["ingest"], "FP64 vector; pre-integration flops (SURVEY.md 8d)"),
        "fe_lk": roof_bytes("fe_lk", "fe_lk_kernel", fbytes["lk"], "algorithmic patch bytes: N x (L + 1) x 2 x 23^2 (SURVEY.md 8d)"),
        "fe_pyrdown": roof_bytes("fe_pyrdown", "fe_pyrdown_kernel", fbytes["pyrdown"], "read the new frame + write the pyramid levels"),
        "fe_fast": roof_bytes("fe_fast", "fe_fast_kernel", fbytes["fast"], "read the new frame once"),
    }
    dom = max(kms, key=lambda k: kms[k])
    roof = dict(rk[dom]) if dom in rk else dict(rk["be_solve"])
    if tj and roof.get("traffic") is not None:
        roof["traffic_source"] = "profiles/%s (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction)" % tj[0]
    roof["frontend_GBps"] = sum(fbytes.values()) * S_launch / (fe_ms * 1e-3) / 1e9 if fe_ms > 0 else None
    roof["backend_TFLOPs"] = (flops["solve"] + flops["marg"] + flops["ingest"]) * S_launch / (be_ms * 1e-3) / 1e12 if be_ms > 0 else None
    roof["stream_groups"] = n_groups
    roofline_kernels = [rk[k] for k in sorted(rk, key=lambda k: -kms[k]) if k != dom]

    # ---- CPU baseline: the oracle (port of the reference algorithm) on the same workload, rank 0 at N = 1 only
    cpu1 = None
    parity = None
    if rank == 0 and world == 1 and args.cpu_seqs > 0:
        ncs = min(args.cpu_seqs, S)
        Fc = min(F, n_pre + Wm + 44)      # bounded sample: ~44 steady-state frames per sequence
        tcpu, nfr, rm, ate_pairs = 0.0, 0, [], []
        for s in range(ncs):
            o = vio_ct.OraclePipeline(cfg)
            o.set_tracker_lag(args.tracker_lag)
            ti, ai, gi = syn.imu(seq0 + s, nimu)
            o.push_imu(ti, ai, gi)
            traj = []
            for f in range(Fc):
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
        cpu1 = dict(value=nfr / tcpu if tcpu > 0 else None, unit="frames/s", cores=1, kind="port",
                    sample="%d sequences x %d steady-state frames of the same rendered workload through oracle/ (-O3, 1 thread; "
                           "the reference binary needs ROS/OpenCV/Ceres and cannot be built here)" % (ncs, nfr // max(ncs, 1)),
                    cpu_seconds=tcpu)
        ah = float(np.mean([a for a, _ in ate_pairs])) if ate_pairs else None
        ao = float(np.mean([b_ for _, b_ in ate_pairs])) if ate_pairs else None
        rel = [abs(a - b_) / b_ for a, b_ in ate_pairs if b_ > 0]
        parity = dict(traj_rmse_hip_vs_oracle_m=float(np.max(rm)) if rm else None, sequences=ncs, per_sequence=rm,
                      ate_hip_m=ah, ate_oracle_m=ao, ate_rel_diff=(abs(ah - ao) / ao if ao else None),
                      ate_rel_diff_worst_sequence=(float(np.max(rel)) if rel else None),
                      note="north-star tolerance: ATE of the HIP path within 1 % of the reference algorithm (oracle) on identical input")
    cpu = cpu1
    cpu_all = None
    eff, eff_how = effective_cores()
    nproc = eff if args.cpu_procs < 0 else min(args.cpu_procs, os.cpu_count() or 1)
    if rank == 0 and world == 1 and nproc > 1:
        # all-core leg: one oracle process per EFFECTIVE core (cgroup quota, not the 256 hardware threads the box shows), the frames rendered
        # beforehand by the parent, every worker past its initialisation before any of them is timed
        import multiprocessing as mp
        import shutil
        import tempfile
        from concurrent.futures import ThreadPoolExecutor
        n_cf = 36
        shm = tempfile.mkdtemp(prefix="vio_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            ctimes = vio_ct.frame_times(sc, n_cf)
            nimu_c = int(n_cf / sc.cam_rate * sc.imu_rate) + 64

            def _prep(i):
                seq = seq0 + 1000 + i
                fr = [syn.render_host(seq, float(tf)) for tf in ctimes]
                ti, ai, gi = syn.imu(seq, nimu_c)
                pth = os.path.join(shm, "seq_%d.npz" % i)
                np.savez(pth, gray=np.stack([x[0] for x in fr]), depth=np.stack([x[1] for x in fr]), times=np.asarray(ctimes), imu_t=ti, imu_acc=ai, imu_gyr=gi)
                return pth
            with ThreadPoolExecutor(max_workers=nproc) as tp:
                paths = list(tp.map(_prep, range(nproc)))
            cfg_bytes = bytes(cfg)
            ctx = mp.get_context("spawn")
            # the same worker alone on the machine first: the one-core rate of THIS harness (the `one_core` leg above runs inside the bench process,
            # on the bench's own sequences, beside the HIP runtime's threads)
            with ctx.Pool(1, initializer=_cpu_init, initargs=(None,)) as pool:
                res1 = pool.map(_cpu_worker, [(paths[0], cfg_bytes)], chunksize=1)
            c0 = time.perf_counter()
            with ctx.Pool(nproc, initializer=_cpu_init, initargs=(ctx.Barrier(nproc),)) as pool:
                res = pool.map(_cpu_worker, [(pth, cfg_bytes) for pth in paths], chunksize=1)
            wall = time.perf_counter() - c0
        finally:
            shutil.rmtree(shm, ignore_errors=True)
        cpu_all = dict(value=float(sum(n / t for t, n in res if t > 0)), unit="frames/s", cores=nproc, cores_are="effective cores: " + eff_how,
                       hardware_threads_visible=len(os.sched_getaffinity(0)), physical_cores=physical_cores(), kind="port",
                       sample="%d oracle processes (one per effective core; they load only oracle/liboracle.so), one pre-rendered sequence of %d frames each, "
                              "timed frames started behind a barrier; sum of the per-process steady-state rates (the reference's own effective threading is "
                              "one back-end thread per estimator)" % (nproc, n_cf),
                       wall_seconds=wall,
                       one_worker_alone=(float(res1[0][1] / res1[0][0]) if res1[0][0] > 0 else None))
        cpu = dict(cpu_all)
        cpu["one_core"] = cpu1

    def r3(x):
        return None if x is None else float("%.4g" % x)

    roof_c = {k: roof.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
    roof_c = {k: (r3(v) if isinstance(v, float) else v) for k, v in roof_c.items()}
    roof_c["kernel"] = "ps_* (phased dogleg solver, %d sequences per launch chain)" % S_launch if dom == "be_solve" and phased else roof.get("kernel")
    roof_c["ms"] = r3(roof.get("ms"))
    # device-wide fraction next to the per-launch-chain one: the algorithmic FP64 flops of ALL stream groups' back-end
    # work of one step (solve + marginalisation + pre-integration, S sequences) over the step time the driver sees, against the same peak
    dev_tflops = (flops["solve"] + flops["marg"] + flops["ingest"]) * S / (elapsed / K) / 1e12
    roof_c["device_wide"] = {"achieved": r3(dev_tflops), "frac": r3(dev_tflops / FP64_PEAK_TFLOPS),
                             "note": "all %d stream groups: S x back-end flops per frame / ms_per_step" % n_groups}
    cpu_c = None
    if cpu:
        cpu_c = {k: cpu.get(k) for k in ("value", "unit", "cores", "cores_are", "hardware_threads_visible", "physical_cores", "kind", "one_worker_alone") if k in cpu}
        if cpu_c.get("one_worker_alone"):
            cpu_c["one_worker_alone"] = r3(cpu_c["one_worker_alone"])
        cpu_c["value"] = r3(cpu_c.get("value"))
        cpu_c["sample"] = ("one pre-rendered sequence of 36 frames per effective core, timed behind a barrier, sum of steady-state rates" if cpu_all else
                           "%d sequences of the bench workload on one core" % min(args.cpu_seqs, S))
        if cpu_all and cpu1:
            cpu_c["one_core_value"] = r3(cpu1["value"])
    # ONE compact line for the driver (its record keeps the tail of stdout); everything else goes to the detail file
    line = {
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
        "valid": bool(all_valid),
        "config": {"workload": "BASELINE configs[2]: %d independent synthetic 640x480 RGB-D + 200 Hz IMU sequences per MI355X, 150 features, "
                               "10-keyframe window" % S, "sequences_per_gpu": S, "tracker_lag": args.tracker_lag,
                   "parallelism": "sequences sharded per GPU, no data-path collective"},
        "repeats": {"n": R, "frames_per_s": [round(x) for x in rates], "timed_frames_per_sequence": R * K},
        "ate_m": {"mean": r3(float(np.mean(ates))) if ates else None, "job_rms": r3(shard.ate_from_sums(sq_err_all, n_ate_all))},
        "solver": {"mean_iterations": r3(iters), "timed_solves": int(d_it[1]), "reboots": reboots},
        "frontend_ms": r3(fe_ms), "backend_ms": r3(be_ms),
        "roofline": roof_c,
        "cpu_baseline": cpu_c,
    }
    if parity:
        line["parity"] = {"sequences": parity["sequences"], "traj_rmse_hip_vs_oracle_m": r3(parity["traj_rmse_hip_vs_oracle_m"]),
                          "ate_rel_diff": r3(parity["ate_rel_diff"])}
    if pcie:
        line["pcie_inclusive_frames_per_s"] = {"pageable": round(pcie["frames_per_s"]), "pinned": round(pcie["pinned"]["frames_per_s"])}
    if world > 1:
        line["per_rank"] = [dict(rank=r["rank"], device=r["device"].get("uuid", r["device"].get("pci_bus_id", r["device"]["index"])),
                                 frames_per_s=round(r["frames_per_s"]), ate_rms_m=r3(r["ate_rms_m"]), valid=r["valid"]) for r in per_rank]
    detail = {
        "config": {"workload": "BASELINE configs[2]: batch of %d independent synthetic 640x480 RGB-D + 200 Hz IMU sequences per MI355X, "
                               "150 max features, 5x6 grid, 10-keyframe window, landmarks free (fix_depth 0)" % S,
                   "sequences_per_gpu": S, "image": [Wd, H], "max_cnt": cfg.max_cnt, "window_size": cfg.window_size,
                   "parallelism": "independent sequences sharded per GPU, no data-path collective",
                   "tracker_lag": args.tracker_lag,
                   "tracker_lag_note": "1 = the tracker of frame f+1 overlaps the optimisation of frame f and predicts with latest_Bg / td as of "
                                       "frame f-1 (the reference's process_tracker / process threads with the estimator one frame behind); "
                                       "0 = it waits for the optimisation of frame f.  The oracle in `parity` runs the same ordering."},
        "repeats": {"n": R, "frames_per_s": rates, "median_index": order, "spread_rel": (max(rates) - min(rates)) / (total_frames / elapsed),
                    "timed_frames_per_sequence": R * K},
        "valid": bool(all_valid),
        "per_rank": per_rank,
        "ate_m": {"mean": float(np.mean(ates)) if ates else None, "max": float(np.max(ates)) if ates else None, "sequences": len(ates),
                  "median": float(np.median(ates)) if ates else None, "worst_sequence": seq0 + worst,
                  "job_rms": shard.ate_from_sums(sq_err_all, n_ate_all)},
        "solver": {"mean_iterations": iters, "timed_solves": int(d_it[1]), "mean_residuals": nres, "mean_var_landmarks": nvar, "mean_tracks": ntrk,
                   "reboots": reboots},
        "kernels_ms": kms,
        "frontend_ms": fe_ms,
        "backend_ms": be_ms,
        "roofline": roof,
        "roofline_kernels": roofline_kernels,
        "cpu_baseline": cpu,
        "parity": parity,
        "pcie_inclusive": pcie,
        "imu_streaming": stream,
    }
    if args.dump:
        np.savez(args.dump + ".rank%d.npz" % rank, seq0=seq0, windows=np.stack([b.window(s) for s in range(S)]),
                 odometry=np.stack([hist[s][-min(len(hist[x]) for x in range(S)):] for s in range(S)]))
    b.close()
    del gray, depth
    torch.cuda.empty_cache()
    if args.aux and rank == 0 and world == 1:
        # auxiliary data points on the same device, never `value`: the other tracker ordering at S, and larger batches
        detail["lag0"] = aux_rate(P, vio_ct, torch, cfg, sc, dev, S, n_pre, Wm, K, lag=0, seq0=seq0)
        detail["lag0"]["note"] = ("tracker lag 0: the tracker of frame f+1 waits for the optimisation of frame f (the front-end is on the critical "
                                  "path); `value` is measured with config.tracker_lag = %d" % args.tracker_lag)
        for s_aux in (256, 512):
            os.environ["VIO_GROUP_SEQS"] = str(s_aux // 2)
            detail["aux_s%d" % s_aux] = aux_rate(P, vio_ct, torch, cfg, sc, dev, s_aux, n_pre, Wm, K, lag=args.tracker_lag)
        # round 6: the fused evaluate + assemble kernel (VIO_FUSE = 1, read at vio_create; default off -- DESIGN.md 4) at the three batch sizes
        had_fuse = os.environ.get("VIO_FUSE")
        os.environ["VIO_FUSE"] = "1"
        for s_aux in (S, 256, 512):
            os.environ["VIO_GROUP_SEQS"] = str(s_aux // 2)
            detail["fused_s%d" % s_aux] = aux_rate(P, vio_ct, torch, cfg, sc, dev, s_aux, n_pre, Wm, K, lag=args.tracker_lag)
            detail["fused_s%d" % s_aux]["note"] = "VIO_FUSE = 1: ps_evalf_kernel instead of ps_eval + ps_asm_a (residual records stay in LDS)"
        if had_fuse is None:
            del os.environ["VIO_FUSE"]
        else:
            os.environ["VIO_FUSE"] = had_fuse
        # BASELINE configs[4] at its per-GPU batch (64 sequences of 1280x720 / 300 features / W = 20): the phased solver with the Schur
        # complement in HBM / L2 (ps_serial_big_kernel)
        os.environ["VIO_GROUP_SEQS"] = "32"
        cfg5 = config5(P)
        c5 = aux_rate(P, vio_ct, torch, cfg5, vio_ct.synth_like(cfg5), dev, 64, cfg5.window_size + 8, min(Wm, 6), min(K, 20), lag=args.tracker_lag)
        sl = c5["solver"]
        fl5 = backend_flops(max(sl["mean_iterations_last_frame"], 1.0), sl["mean_residuals"], sl["mean_var_landmarks"],
                            sl["mean_residuals"] / max(sl["mean_in_problem"], 1.0) + 1.0, cfg5.window_size, sc.imu_rate / sc.cam_rate)
        c5["roofline"] = dict(bound="mfma", kernel="ps_* (phased solver, HBM-resident Schur complement)", unit="TFLOP/s", peak=FP64_PEAK_TFLOPS,
                              achieved=fl5["solve"] * 32 / (c5["kernels_ms"]["be_solve"] * 1e-3) / 1e12,
                              frac=fl5["solve"] * 32 / (c5["kernels_ms"]["be_solve"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, sequences_per_launch=32,
                              note="SURVEY.md 8d flops with the counts of the last timed frame")
        c5["workload"] = "BASELINE configs[4]: 64 sequences of 1280x720, 300 features, 7x8 grid, 20-keyframe window per GPU"
        detail["config5"] = c5
        os.environ["VIO_GROUP_SEQS"] = str(per_group)
        # the literal marginalisation (vio_config.marg_exact = 1: marginalization_factor.cpp:281-315 followed step by step, both
        # eigen-decompositions LDS-resident since round 5) on the headline workload -- what deviations 10 / 13 save
        cfg_x = P.canonical_config(marg_exact=1)
        detail["marg_exact"] = aux_rate(P, vio_ct, torch, cfg_x, sc, dev, S, n_pre, Wm, K, lag=args.tracker_lag, seq0=seq0)
        detail["marg_exact"]["note"] = "marg_exact = 1; be_marg per launch %.3f ms against %.3f ms of the default form" % (
            detail["marg_exact"]["kernels_ms"].get("be_marg", float("nan")), kms.get("be_marg", float("nan")))
        # marg_exact = 2: the same algorithm with the first eigen-decomposition replaced by a certified inverse (include/vio_abi.h)
        cfg_c = P.canonical_config(marg_exact=2)
        detail["marg_certified"] = aux_rate(P, vio_ct, torch, cfg_c, sc, dev, S, n_pre, Wm, K, lag=args.tracker_lag, seq0=seq0)
        detail["marg_certified"]["note"] = "marg_exact = 2; be_marg per launch %.3f ms" % detail["marg_certified"]["kernels_ms"].get("be_marg", float("nan"))
        aux_keys = ("lag0", "aux_s256", "aux_s512", "fused_s%d" % S, "fused_s256", "fused_s512", "config5", "marg_exact", "marg_certified")
        line["aux_frames_per_s"] = {k: round(detail[k]["frames_per_s"]) for k in aux_keys}
        line["aux_valid"] = all(detail[k]["valid"] for k in aux_keys)
    if rank == 0:
        # detail: everything the compact line summarises (per-kernel rooflines, ATE lists, PCIe / streaming legs, aux legs with their kernel
        # times), to a side file and to stderr; the LAST line of stdout is the one JSON line of the contract
        full = dict(line)
        full.update(detail)
        dpath = os.environ.get("VIO_BENCH_DETAIL", os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
        try:
            os.makedirs(os.path.dirname(dpath), exist_ok=True)
            json.dump(full, open(dpath, "w"))
            line["detail_file"] = os.path.relpath(dpath, ROOT)
        except OSError:
            pass
        print(json.dumps(full), file=sys.stderr)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
