"""Oracle-vs-oracle control experiment and the lag-1 ATE fixture (test infrastructure, CPU only; VERDICT r3 "next round" item 1).

Question: the HIP path and the oracle separate by millimetres on 126 of 128 sequences within 300 frames.  Is that a
property of the ALGORITHM (any two arithmetically different implementations separate like that) or a systematic difference of the HIP
path?  The control: the oracle against two other builds of ITS OWN SOURCES that differ only in round-off --
    liboracle_fma.so    -mfma -ffp-contract=fast          (fused multiply-adds wherever the compiler contracts)
    liboracle_order.so  -DORACLE_PERTURB_ORDER            (projection-factor sums accumulated last-to-first; identical front-end code)
on the same 128 sequences x 300 frames the HIP comparison uses (profiles/round3_parity_300_s128.json).  If these separate with the same
statistics, the separation is the estimator's own sensitivity.

The same pass renders every sequence once on the host and also runs the oracle with tracker lag 1 (the ordering bench.py measures),
which gives tests/golden/oracle_ate_300_lag1.npz for 1024 sequences.

    python tests/oracle_control.py run --seqs 1024 --control 128 --procs 6      # resumable: one .npz per sequence in --scratch
    python tests/oracle_control.py assemble                                      # -> profiles/round4_oracle_self_divergence.json,
                                                                                 #    tests/golden/oracle_ate_300_lag1.npz
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

STATUS_KEYS = ("solver_flag", "frame_count", "marginalization_flag", "n_landmarks", "last_track_num", "iterations", "successful_steps",
               "n_residuals", "n_in_problem", "n_var_landmarks")
VARIANTS = {   # name -> (library under oracle/, tracker lag)
    "base": ("liboracle.so", 0),
    "fma": ("liboracle_fma.so", 0),
    "order": ("liboracle_order.so", 0),
    "lag1": ("liboracle.so", 1),
    "lag1_order": ("liboracle_order.so", 1),
    "befma": ("liboracle_befma.so", 0),   # fused multiply-adds in the back-end only (front-end code identical to liboracle.so)
    # a perturbation of known size: every new prior's Jacobian scaled by (1 + eps) (oracle/backend.cpp marg_finish, OVIO_PERTURB_EPS)
    "eps12": ("liboracle.so", 0, {"OVIO_PERTURB_EPS": "1e-12"}),
    "eps9": ("liboracle.so", 0, {"OVIO_PERTURB_EPS": "1e-9"}),
    "eps6": ("liboracle.so", 0, {"OVIO_PERTURB_EPS": "1e-6"}),
    # round 5, attribution of the HIP-vs-oracle difference: the oracle with the HIP path's equivalent formulations switched on one at a time
    # and all together (oracle/oracle.h ODEV_*: 1 = IMU whitening chol(cov)^-1 (deviation 8), 2 = quadratic-form prior (13), 4 = analytic
    # landmark elimination in the marginalisation (10), 8 = frame-pair projection factors (11))
    "dev8": ("liboracle.so", 0, {"OVIO_DEVIATIONS": "1"}),
    "dev13": ("liboracle.so", 0, {"OVIO_DEVIATIONS": "2"}),
    "dev10": ("liboracle.so", 0, {"OVIO_DEVIATIONS": "4"}),
    "dev11": ("liboracle.so", 0, {"OVIO_DEVIATIONS": "8"}),
    "devall": ("liboracle.so", 0, {"OVIO_DEVIATIONS": "15"}),
    "dev10c": ("liboracle.so", 0, {"OVIO_DEVIATIONS": "20"}),     # analytic landmark elimination + Cholesky inverse of the remaining block (oracle.h ODEV_CHOL_PINV)
    "devallc": ("liboracle.so", 0, {"OVIO_DEVIATIONS": "31"}),    # every formulation of the HIP path
    "devall_lag1": ("liboracle.so", 1, {"OVIO_DEVIATIONS": "15"}),
}
DEV_NAMES = ("dev8", "dev13", "dev10", "dev11", "devall", "dev10c", "devallc")


def run_variants(seq, n_frames, names, cfg_kw=None):
    """render sequence seq once (host renderer), run the named oracle variants over the same frames"""
    import vio_ct
    P = vio_ct.pkg()
    cfg = P.canonical_config(**(cfg_kw or {}))
    sc = vio_ct.synth_like(cfg)
    syn = P.Synth(sc)
    frames = [syn.render_host(seq, float(tf)) for tf in vio_ct.frame_times(sc, n_frames)]
    out = {}
    for name in names:
        so, lag = VARIANTS[name][:2]
        env = VARIANTS[name][2] if len(VARIANTS[name]) > 2 else {}
        os.environ.update(env)
        try:
            o = vio_ct.run_oracle_sequence(cfg, sc, seq, n_frames, frames=frames, tracker_lag=lag, lib=os.path.join(vio_ct.ORACLE_DIR, so))
        finally:
            for k_ in env:
                os.environ.pop(k_, None)
        out[name + "_frames"] = np.array([x[0] for x in o["traj"]], np.int32)
        out[name + "_pos"] = np.array([x[1] for x in o["traj"]])
        out[name + "_status"] = np.array([[st[k] for k in STATUS_KEYS] for st in o["status"]])
        out[name + "_reboots"] = int(o["oracle"].status()["reboot_count"])
        if "gt" not in out:
            out["gt"] = np.array(o["gt"])
            out["gt_frames"] = out[name + "_frames"]
        o["oracle"] = None
    return out


def _worker(job):
    seq, n_frames, names, scratch = job
    path = os.path.join(scratch, "seq_%05d.npz" % seq)
    if os.path.exists(path):
        return seq, 0.0
    t0 = time.time()
    out = run_variants(seq, n_frames, names)
    np.savez_compressed(path + ".tmp.npz", names=np.array(names), **out)
    os.replace(path + ".tmp.npz", path)
    return seq, time.time() - t0


def first_flip(sa, sb):
    """first frame at which the per-frame decisions of two runs differ, and which decision: returns (frame, kind) or (None, None)"""
    n = min(len(sa), len(sb))
    col = {k: i for i, k in enumerate(STATUS_KEYS)}
    for f in range(n):
        a, b = sa[f], sb[f]
        if np.array_equal(a, b):
            continue
        # ordered from the earliest stage of a frame to the latest
        if a[col["last_track_num"]] != b[col["last_track_num"]]:
            return f, "tracked set (LK status / F-RANSAC inlier / previous frame's cull)"
        if a[col["marginalization_flag"]] != b[col["marginalization_flag"]]:
            return f, "keyframe decision"
        if a[col["n_in_problem"]] != b[col["n_in_problem"]] or a[col["n_residuals"]] != b[col["n_residuals"]] or a[col["n_var_landmarks"]] != b[col["n_var_landmarks"]]:
            return f, "landmark set of the solve (triangulation / depth flag / dynamic flag)"
        if a[col["iterations"]] != b[col["iterations"]]:
            return f, "iteration count (convergence test)"
        if a[col["successful_steps"]] != b[col["successful_steps"]]:
            return f, "accepted / rejected step"
        if a[col["n_landmarks"]] != b[col["n_landmarks"]]:
            return f, "outlier cull after the solve"
        return f, "solver_flag / frame_count"
    return None, None


def pair_rows(za, zb, na, nb, seq):
    """separation statistics of two runs of one sequence (same definitions as parity_long.compare)"""
    import vio_ct
    fa, pa, fb, pb = za[na + "_frames"], za[na + "_pos"], zb[nb + "_frames"], zb[nb + "_pos"]
    n = min(len(pa), len(pb))
    same = len(fa) == len(fb) and np.array_equal(fa, fb)
    dist = np.linalg.norm(pa[:n] - pb[:n], axis=1)
    gt = za["gt"][:n]
    sep = np.nonzero(dist > 1e-6)[0]
    ff, kind = first_flip(za[na + "_status"], zb[nb + "_status"])
    return dict(sequence=int(seq), frames=int(n), same_frames=bool(same), ate_a_m=vio_ct.ate_rmse(pa[:n], gt), ate_b_m=vio_ct.ate_rmse(pb[:n], gt),
                max_distance_m=float(dist.max()), final_distance_m=float(dist[-1]), first_frame_beyond_1um=(int(fa[sep[0]]) if len(sep) else None),
                first_flip_frame=ff, first_flip_kind=kind)


def summarise(rows):
    aa = np.array([r["ate_a_m"] for r in rows]); ab = np.array([r["ate_b_m"] for r in rows])
    md = np.array([r["max_distance_m"] for r in rows]); diff = ab - aa
    fs = [r["first_frame_beyond_1um"] for r in rows if r["first_frame_beyond_1um"] is not None]
    kinds = {}
    for r in rows:
        kinds[str(r["first_flip_kind"])] = kinds.get(str(r["first_flip_kind"]), 0) + 1
    flips = [r["first_flip_frame"] for r in rows if r["first_flip_frame"] is not None]
    return dict(sequences=len(rows), mean_ate_a_m=float(aa.mean()), mean_ate_b_m=float(ab.mean()),
                signed_rel_diff_of_means=float((ab.mean() - aa.mean()) / aa.mean()),
                standard_error_rel=float(diff.std(ddof=1) / np.sqrt(len(diff)) / aa.mean()),
                sigma_paired_diff_m=float(diff.std(ddof=1)), max_rel_diff_one_sequence=float(np.max(np.abs(diff) / aa)),
                separated_beyond_1um=int((md > 1e-6).sum()), separated_beyond_100um=int((md > 1e-4).sum()), separated_beyond_1mm=int((md > 1e-3).sum()),
                median_max_distance_m=float(np.median(md)), max_distance_m=float(md.max()),
                median_first_frame_beyond_1um=(float(np.median(fs)) if fs else None),
                a_lower_ate=int((aa < ab).sum()), b_lower_ate=int((ab < aa).sum()),
                sequences_with_a_decision_flip=len(flips), median_first_flip_frame=(float(np.median(flips)) if flips else None),
                first_flip_census=kinds)


def cmd_run(a):
    os.makedirs(a.scratch, exist_ok=True)
    import vio_ct
    for so in {v[0] for k, v in VARIANTS.items() if not a.only or k in a.only.split(",")}:
        vio_ct.oracle(os.path.join(vio_ct.ORACLE_DIR, so))   # build what is missing before the pool starts
    jobs = []
    for i in range(a.seqs):
        names = ["base", "fma", "order", "lag1", "lag1_order"] if i < a.control else ["lag1"]
        if a.only:   # a later pass that adds variants for the control sequences into its own scratch directory (merged by assemble)
            names = a.only.split(",")
        jobs.append((a.seq0 + i, a.frames, names, a.scratch))
    t0 = time.time()
    done = 0
    with mp.get_context("spawn").Pool(a.procs) as pool:
        for seq, dt in pool.imap_unordered(_worker, jobs, chunksize=1):
            done += 1
            if done % 16 == 0:
                print("%d / %d sequences, %.0f s" % (done, len(jobs), time.time() - t0), flush=True)


def cmd_assemble(a):
    import vio_ct
    have = sorted(int(f[4:9]) for f in os.listdir(a.scratch) if f.startswith("seq_") and f.endswith(".npz") and ".tmp" not in f)
    Z = {s: dict(np.load(os.path.join(a.scratch, "seq_%05d.npz" % s))) for s in have}
    extra_pairs = []
    for x in [d for d in a.extra.split(",") if d]:   # scratch directories of --only passes: their arrays join the sequence's record
        for f in sorted(os.listdir(x)):
            if f.startswith("seq_") and f.endswith(".npz") and ".tmp" not in f and int(f[4:9]) in Z:
                zx = np.load(os.path.join(x, f))
                for k in zx.files:
                    if k not in ("gt", "gt_frames", "names"):
                        Z[int(f[4:9])][k] = zx[k]
    for nm in ("befma", "eps12", "eps9", "eps6"):
        if any(nm + "_pos" in Z[s] for s in have) and all(nm + "_pos" in Z[s] for s in have if "base_pos" in Z[s]):
            extra_pairs.append(("base", nm))
    ctl = [s for s in have if "base_pos" in Z[s]]
    rep = dict(what="oracle vs builds of its own sources that differ only in round-off (tests/oracle_control.py); canonical workload, "
                    "%d frames, sequences %d..%d" % (a.frames, ctl[0], ctl[-1]) if ctl else "",
               variants={k: dict(library=v[0], tracker_lag=v[1], env=(v[2] if len(v) > 2 else {})) for k, v in VARIANTS.items()}, status_keys=list(STATUS_KEYS), pairs={})
    # the base run must be the oracle the committed lag-0 fixture came from
    fx = np.load(os.path.join(HERE, "golden", "oracle_ate_300.npz"))
    nfx = 0
    for s in ctl:
        i = s - int(fx["seq0"])
        if 0 <= i < len(fx["positions"]):
            z = Z[s]
            assert np.array_equal(fx["positions"][i][z["base_frames"]], z["base_pos"]), "base run differs from the committed fixture (seq %d)" % s
            nfx += 1
    rep["base_bit_identical_to_lag0_fixture_sequences"] = nfx
    for na, nb in [("base", "fma"), ("base", "order"), ("fma", "order"), ("lag1", "lag1_order"), ("base", "lag1")] + extra_pairs:
        rows = [pair_rows(Z[s], Z[s], na, nb, s) for s in ctl]
        rep["pairs"]["%s_vs_%s" % (na, nb)] = dict(summary=summarise(rows), rows=rows)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(rep, open(a.out, "w"), indent=1)
    # per-frame decisions + positions of the oracle for the control sequences, both tracker orderings: what tools/flip_census.py compares
    # the HIP path with on the GPU box (gpurun_out/ does not travel)
    if ctl:
        dec = np.zeros((2, len(ctl), a.frames, len(STATUS_KEYS)), np.int16)   # [tracker lag][sequence][frame][STATUS_KEYS]
        for i, s_ in enumerate(ctl):
            for li, nm in enumerate(("base", "lag1")):
                st = Z[s_][nm + "_status"]
                dec[li, i, :len(st)] = st
        np.savez_compressed(os.path.join(HERE, "golden", "oracle_decisions_300.npz"), seq0=ctl[0], frames=a.frames, keys=np.array(STATUS_KEYS), decisions=dec)
    for k, v in rep["pairs"].items():
        print(k, json.dumps(v["summary"]))
    # lag-1 fixture, same layout as oracle_ate_300.npz
    l1 = [s for s in have if "lag1_pos" in Z[s]]
    if len(l1) < a.seqs:
        print("lag-1 fixture not written: %d of %d sequences done" % (len(l1), a.seqs))
    elif l1 == list(range(l1[0], l1[0] + len(l1))):
        n = len(l1)
        ate = np.zeros(n); nfr = np.zeros(n, np.int32); reb = np.zeros(n, np.int32); first = np.zeros(n, np.int32)
        pos = np.zeros((min(a.keep, n), a.frames, 3))
        for i, s in enumerate(l1):
            z = Z[s]
            fr, po = z["lag1_frames"], z["lag1_pos"]
            ate[i] = vio_ct.ate_rmse(po, z["gt"][:len(po)]) if "gt" in z and len(z["gt"]) == len(po) else np.nan
            nfr[i] = len(po); reb[i] = int(z["lag1_reboots"]); first[i] = fr[0]
            if i < len(pos):
                pos[i, fr] = po
        assert np.isfinite(ate).all()
        np.savez_compressed(a.fixture, seq0=l1[0], frames=a.frames, tracker_lag=1, ate=ate, n_rows=nfr, reboots=reb, first_frame=first, positions=pos)
        print("lag-1 fixture: %d sequences, mean ATE %.4f mm, reboots %d -> %s" % (n, ate.mean() * 1e3, reb.sum(), a.fixture))


def cmd_fixtures(a):
    """round 6 (one deterministic renderer + the projected line search in the oracle): the three committed fixtures from ONE pass that ran
    `run --only base,lag1` over all sequences -- tests/golden/oracle_ate_300.npz (lag 0), oracle_ate_300_lag1.npz (lag 1) and the per-frame
    decisions of the first --control sequences (oracle_decisions_300.npz, what tools/flip_census.py compares the HIP path with)"""
    import vio_ct
    have = sorted(int(f[4:9]) for f in os.listdir(a.scratch) if f.startswith("seq_") and f.endswith(".npz") and ".tmp" not in f)
    assert have == list(range(a.seq0, a.seq0 + a.seqs)), "scratch holds %d sequences, wanted %d from %d" % (len(have), a.seqs, a.seq0)
    Z = {s: dict(np.load(os.path.join(a.scratch, "seq_%05d.npz" % s))) for s in have}
    n = len(have)
    for nm, lag, path in (("base", 0, os.path.join(HERE, "golden", "oracle_ate_300.npz")), ("lag1", 1, a.fixture)):
        ate = np.zeros(n); nfr = np.zeros(n, np.int32); reb = np.zeros(n, np.int32); first = np.zeros(n, np.int32)
        pos = np.zeros((min(a.keep, n), a.frames, 3))
        for i, s_ in enumerate(have):
            z = Z[s_]
            fr, po = z[nm + "_frames"], z[nm + "_pos"]
            gt = z["gt"] if len(z["gt"]) == len(po) else None
            assert gt is not None and np.array_equal(z["gt_frames"], fr), s_
            ate[i] = vio_ct.ate_rmse(po, gt); nfr[i] = len(po); reb[i] = int(z[nm + "_reboots"]); first[i] = fr[0]
            if i < len(pos):
                pos[i, fr] = po
        assert np.isfinite(ate).all()
        kw = dict(tracker_lag=1) if lag else {}
        np.savez_compressed(path, seq0=have[0], frames=a.frames, ate=ate, n_rows=nfr, reboots=reb, first_frame=first, positions=pos, **kw)
        print("lag-%d fixture: %d sequences, mean ATE %.4f mm, reboots %d -> %s" % (lag, n, ate.mean() * 1e3, reb.sum(), path))
    ctl = have[:a.control]
    dec = np.zeros((2, len(ctl), a.frames, len(STATUS_KEYS)), np.int16)   # [tracker lag][sequence][frame][STATUS_KEYS]
    for i, s_ in enumerate(ctl):
        for li, nm in enumerate(("base", "lag1")):
            st = Z[s_][nm + "_status"]
            dec[li, i, :len(st)] = st
    np.savez_compressed(os.path.join(HERE, "golden", "oracle_decisions_300.npz"), seq0=ctl[0], frames=a.frames, keys=np.array(STATUS_KEYS), decisions=dec)


def cmd_census_fixture(a):
    """tests/golden/oracle_dev31_300.npz: positions of the oracle with every HIP formulation switched on (variant devallc, OVIO_DEVIATIONS = 31)
    for the --seqs sequences of a `run --only devallc` pass: what tests/test_gpu_parity3.py::test_census_128_sequences_against_the_matched_oracle
    compares the HIP path with."""
    have = sorted(int(f[4:9]) for f in os.listdir(a.scratch) if f.startswith("seq_") and f.endswith(".npz") and ".tmp" not in f)
    assert have == list(range(a.seq0, a.seq0 + a.seqs)), (len(have), a.seqs)
    n = len(have)
    pos = np.zeros((n, a.frames, 3)); nfr = np.zeros(n, np.int32); first = np.zeros(n, np.int32); reb = np.zeros(n, np.int32)
    for i, s_ in enumerate(have):
        z = np.load(os.path.join(a.scratch, "seq_%05d.npz" % s_))
        fr, po = z["devallc_frames"], z["devallc_pos"]
        pos[i, fr] = po; nfr[i] = len(po); first[i] = fr[0]; reb[i] = int(z["devallc_reboots"])
    assert reb.sum() == 0
    out = os.path.join(HERE, "golden", "oracle_dev31_300.npz")
    np.savez_compressed(out, seq0=have[0], frames=a.frames, deviations=31, n_rows=nfr, first_frame=first, positions=pos)
    print("census fixture: %d sequences -> %s" % (n, out))


def early_rows(za, na, nb, n_early=30):
    """largest distance of two runs over the first n_early published positions (the window in which implementations still agree)"""
    pa, pb = za[na + "_pos"], za[nb + "_pos"]
    n = min(len(pa), len(pb), n_early)
    return float(np.linalg.norm(pa[:n] - pb[:n], axis=1).max())


def cmd_attribution(a):
    """round 5: base vs each single deviation and vs all of them (same statistics as the control pairs) + the early-frame distances"""
    have = sorted(int(f[4:9]) for f in os.listdir(a.scratch) if f.startswith("seq_") and f.endswith(".npz") and ".tmp" not in f)
    Z = {s: dict(np.load(os.path.join(a.scratch, "seq_%05d.npz" % s))) for s in have}
    ctl = [s for s in have if "base_pos" in Z[s]]
    fx = np.load(os.path.join(HERE, "golden", "oracle_ate_300.npz"))
    nfx = 0
    for s in ctl:
        i = s - int(fx["seq0"])
        if 0 <= i < len(fx["positions"]):
            z = Z[s]
            assert np.array_equal(fx["positions"][i][z["base_frames"]], z["base_pos"]), "base run differs from the committed fixture (seq %d)" % s
            nfx += 1
    rep = dict(what="the oracle against itself with the HIP path's equivalent formulations switched on (OVIO_DEVIATIONS, oracle/oracle.h ODEV_*), "
                    "%d frames, sequences %d..%d, tracker lag 0" % (a.frames, ctl[0], ctl[-1]),
               variants={k: VARIANTS[k][2] for k in DEV_NAMES}, base_bit_identical_to_lag0_fixture_sequences=nfx, pairs={})
    for x in [d for d in a.extra.split(",") if d]:   # scratch directories of later --only passes (more variants of the same sequences)
        for f in sorted(os.listdir(x)):
            if f.startswith("seq_") and f.endswith(".npz") and ".tmp" not in f and int(f[4:9]) in Z:
                zx = np.load(os.path.join(x, f))
                for k in zx.files:
                    if k not in ("gt", "gt_frames", "names"):
                        Z[int(f[4:9])][k] = zx[k]
    names = [nm for nm in DEV_NAMES if all(nm + "_pos" in Z[s] for s in ctl)]
    top = "devallc" if "devallc" in names else "devall"
    for na, nb in [("base", nm) for nm in names] + [(top, nm) for nm in names if nm != top]:
        rows = [pair_rows(Z[s], Z[s], na, nb, s) for s in ctl]
        e30 = np.array([early_rows(Z[s], na, nb) for s in ctl])
        sm = summarise(rows)
        sm.update(early_30_rows_max_distance_m=dict(median=float(np.median(e30)), p90=float(np.percentile(e30, 90)), max=float(e30.max()),
                                                    below_1e_11=int((e30 <= 1e-11).sum()), below_1e_9=int((e30 <= 1e-9).sum())))
        rep["pairs"]["%s_vs_%s" % (na, nb)] = dict(summary=sm, rows=rows)
        print(na, nb, json.dumps({k: sm[k] for k in ("separated_beyond_1um", "separated_beyond_1mm", "median_max_distance_m", "median_first_frame_beyond_1um",
                                                     "median_first_flip_frame", "signed_rel_diff_of_means", "standard_error_rel", "early_30_rows_max_distance_m")}))
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(rep, open(a.out, "w"), indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cmd", choices=("run", "assemble", "attribution", "fixtures", "census-fixture"))
    ap.add_argument("--seqs", type=int, default=1024)
    ap.add_argument("--control", type=int, default=128, help="leading sequences that also run the control variants")
    ap.add_argument("--seq0", type=int, default=700)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--procs", type=int, default=6)
    ap.add_argument("--keep", type=int, default=128)
    ap.add_argument("--only", default="", help="run: only these variants (comma separated) for every sequence of the range")
    ap.add_argument("--extra", default="", help="assemble: scratch directories of --only passes to merge (comma separated)")
    ap.add_argument("--scratch", default=os.path.join(ROOT, "gpurun_out", "oracle_control"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "round4_oracle_self_divergence.json"))
    ap.add_argument("--fixture", default=os.path.join(HERE, "golden", "oracle_ate_300_lag1.npz"))
    a = ap.parse_args()
    {"run": cmd_run, "assemble": cmd_assemble, "attribution": cmd_attribution, "fixtures": cmd_fixtures, "census-fixture": cmd_census_fixture}[a.cmd](a)


if __name__ == "__main__":
    main()
