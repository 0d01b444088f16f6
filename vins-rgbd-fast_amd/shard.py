"""Sharding of independent sequences over GPUs: block partition, no data-path collective.

Sequences never exchange data (all state of a sequence lives behind its slot of one vio_batch), so the only
cross-rank traffic is the timing barrier and one small reduction of the job totals at the end.  Works with any
torch.distributed backend: "nccl" (= RCCL over xGMI) on the GPU box, "gloo" in the CPU tests."""
import numpy as np


def sequence_shard(rank, world, seqs_per_gpu):
    """Weak scaling: rank r owns the global sequence ids [r * S, (r + 1) * S)."""
    if not (0 <= rank < world) or seqs_per_gpu <= 0:
        raise ValueError("bad shard request rank=%r world=%r seqs_per_gpu=%r" % (rank, world, seqs_per_gpu))
    return range(rank * seqs_per_gpu, (rank + 1) * seqs_per_gpu)


def block_partition(n_total, world):
    """Strong-scaling variant: split n_total sequences into `world` contiguous blocks whose sizes differ by at most 1."""
    base, rem = divmod(n_total, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append(range(lo, hi))
        lo = hi
    return out


def job_totals(frames_local, elapsed_local, sq_err_local=0.0, n_pose_local=0, device=None):
    """Whole-job numbers from per-rank ones: frames = SUM, elapsed = MAX over ranks (the slowest rank bounds the job),
    position error sums = SUM.  Returns (frames, elapsed_s, sq_err, n_pose).  No-op when torch.distributed is not initialised."""
    try:
        import torch
        import torch.distributed as dist
    except ImportError:  # single process without torch
        return frames_local, elapsed_local, sq_err_local, n_pose_local
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return frames_local, elapsed_local, sq_err_local, n_pose_local
    kw = {"device": device} if device is not None else {}
    s = torch.tensor([float(frames_local), float(sq_err_local), float(n_pose_local)], dtype=torch.float64, **kw)
    m = torch.tensor([float(elapsed_local)], dtype=torch.float64, **kw)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return int(round(s[0].item())), float(m.item()), float(s[1].item()), int(round(s[2].item()))


def frames_per_second(frames_total, elapsed_max):
    return frames_total / elapsed_max if elapsed_max > 0 else float("nan")


def ate_from_sums(sq_err, n_pose):
    return float(np.sqrt(sq_err / n_pose)) if n_pose > 0 else None
