"""Worker of test_host_cpu.py::test_two_rank_sharding_gloo: the N>1 control flow of bench.py on CPU.

Each rank takes its block of sequences (vins-rgbd-fast_amd/shard.py), runs them through the CPU oracle's front-end (the
checker stands in for the GPU here -- this test is about the sharding / reduction logic, not the kernels), brackets the
work with barriers exactly like bench.py and reduces the job totals."""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import vio_ct  # noqa: E402


def run_sequence(P, cfg, sc, seq, n_frames):
    syn = P.Synth(sc)
    tr = vio_ct.OracleTracker(cfg)
    sig = []
    for t in 2.0 + np.arange(n_frames) * 0.1:
        g, _ = syn.render_host(seq, float(t))
        tr.read(g, float(t), None, True)
        ids, cnt, cur, _, _ = tr.tracks()
        sig.append([int(len(ids)), int(ids.sum()), int(cnt.sum()), float(np.float64(cur).sum())])
    return sig


def main():
    out_dir, seqs_per_rank, n_frames = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    P = vio_ct.pkg()
    shard = importlib.import_module("vins-rgbd-fast_amd.shard")
    cfg = P.default_config(width=320, height=240, max_cnt=60, fx=300.0, fy=300.0, cx=160.0, cy=120.0)
    sc = vio_ct.synth_like(cfg)
    mine = shard.sequence_shard(rank, world, seqs_per_rank)
    dist.barrier()
    t0 = time.perf_counter()
    res = {s: run_sequence(P, cfg, sc, s, n_frames) for s in mine}
    if rank == 1:
        time.sleep(0.25)  # make the ranks' clocks differ: the job time must be the MAX
    el = time.perf_counter() - t0
    dist.barrier()
    frames, elapsed, sq, n = shard.job_totals(len(mine) * n_frames, el, sq_err_local=float(rank + 1), n_pose_local=len(mine))
    json.dump(dict(rank=rank, world=world, seqs=list(mine), local_elapsed=el, frames=frames, elapsed=elapsed, sq=sq, n=n, res=res),
              open(os.path.join(out_dir, "rank%d.json" % rank), "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
