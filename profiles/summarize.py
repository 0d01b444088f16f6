"""Condenses the rocprofv3 CSVs written by profiles/collect.sh into the small files kept under profiles/.
usage: summarize.py <rocprof_out_dir> <summary_dir> <round> <sequences_per_gpu>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(root, sub, suffix):
    return sorted(glob.glob(os.path.join(root, sub, "**", "*" + suffix), recursive=True))


def short(name):
    return name.split("(")[0].strip()


def main():
    root, out, rnd, seqs = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    os.makedirs(out, exist_ok=True)
    # ---- pass 1: kernel stats
    stats = find(root, "stats", "kernel_stats.csv")
    rows = []
    for f in stats:
        rows += list(csv.DictReader(open(f)))
    keep = [r for r in rows if "vio" in r.get("Name", "") or "_kernel" in r.get("Name", "")]
    with open(os.path.join(out, rnd + "_kernel_stats.csv"), "w") as fo:
        cols = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]
        fo.write(",".join(cols) + "\n")
        for r in rows:
            fo.write(",".join('"%s"' % short(r.get(c, "")) if c == "Name" else str(r.get(c, "")) for c in cols) + "\n")
    print("kernel stats: %d rows from %s" % (len(rows), stats))
    # steady-state average from the trace itself (last half of the dispatches of each kernel)
    tr = find(root, "stats", "kernel_trace.csv")
    dur = defaultdict(list)
    for f in tr:
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    steady = {k: dict(calls=len(v), steady_avg_ms=sum(v[len(v) // 2:]) / max(1, len(v) - len(v) // 2) / 1e6, max_ms=max(v) / 1e6)
              for k, v in dur.items()}
    # ---- pmc passes
    def pmc(sub):
        acc = defaultdict(lambda: defaultdict(list))
        for f in find(root, sub, "counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        tot = {k: {c: (sum(v[len(v) // 2:]), len(v) - len(v) // 2) for c, v in d.items()} for k, d in acc.items()}
        pmc_totals[sub] = tot
        return {k: {c: sum(v[len(v) // 2:]) / max(1, len(v) - len(v) // 2) for c, v in d.items()} for k, d in acc.items()}
    pmc_totals = {}
    fetch, write, sq, l2 = pmc("fetch"), pmc("write"), pmc("sq"), pmc("l2")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        fs, ws = fetch.get(k, {}).get("FETCH_SIZE"), write.get(k, {}).get("WRITE_SIZE")
        if fs is None and ws is None:
            continue
        # FETCH_SIZE / WRITE_SIZE are reported in KB; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts 128-B requests as 64 B -> x2
        hb = (2.0 * (fs or 0.0) + (ws or 0.0)) * 1024.0
        kernels[k] = dict(FETCH_SIZE_KB=fs, WRITE_SIZE_KB=ws, hbm_bytes_per_launch=hb)
        if k in steady:
            kernels[k]["steady_avg_ms"] = steady[k]["steady_avg_ms"]
            kernels[k]["hbm_GBps"] = hb / (steady[k]["steady_avg_ms"] * 1e-3) / 1e9 if steady[k]["steady_avg_ms"] > 0 else None
        if k in l2 and (l2[k].get("TCC_HIT_sum", 0) + l2[k].get("TCC_MISS_sum", 0)) > 0:
            kernels[k]["l2_hit_rate"] = l2[k]["TCC_HIT_sum"] / (l2[k]["TCC_HIT_sum"] + l2[k]["TCC_MISS_sum"])
    # the phased solver is a chain of small kernels (ps_*): its traffic per solve = everything those kernels moved in the steady half of
    # the run / the number of steps in it (one be_marg_kernel launch per step and stream group)
    def per_step(sub, counter):
        tot = pmc_totals.get(sub, {})
        steps = next((v[counter][1] for k, v in tot.items() if k.startswith("be_marg_kernel") and counter in v), 0)
        if not steps:
            return None
        return sum(v[counter][0] for k, v in tot.items() if k.startswith("ps_") and counter in v) / steps
    fs, ws = per_step("fetch", "FETCH_SIZE"), per_step("write", "WRITE_SIZE")
    if fs is not None or ws is not None:
        ent = dict(FETCH_SIZE_KB=fs, WRITE_SIZE_KB=ws, hbm_bytes_per_launch=(2.0 * (fs or 0.0) + (ws or 0.0)) * 1024.0,
                   note="sum over the ps_* kernels of one solve (all iteration slots of one stream group)")
        ms = sum(v["steady_avg_ms"] * v["calls"] for k, v in steady.items() if k.startswith("ps_")) / max(1, steady.get("be_marg_kernel", {}).get("calls", 0))
        ent["busy_ms_per_step"] = ms
        kernels["be_solve_phased"] = ent
    per_launch = int(os.environ.get("VIO_GROUP_SEQS", "64"))
    json.dump(dict(round=rnd, sequences_per_gpu=seqs, sequences_per_launch=min(seqs, per_launch), note="averages over the second half of each kernel's dispatches (steady state); "
                   "hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KB, gfx950 correction per MI355X_MICROARCH.md; WRITE_SIZE uncalibrated",
                   kernels=kernels), open(os.path.join(out, rnd + "_pmc_traffic.json"), "w"), indent=1)
    json.dump(dict(round=rnd, sequences_per_gpu=seqs, steady=steady, sq=sq, l2=l2), open(os.path.join(out, rnd + "_pmc_sq.json"), "w"), indent=1)
    for k, v in sorted(steady.items(), key=lambda kv: -kv[1]["steady_avg_ms"])[:12]:
        print("%-28s calls %4d steady avg %8.3f ms  traffic %s" % (k, v["calls"], v["steady_avg_ms"],
              ("%.1f MB" % (kernels[k]["hbm_bytes_per_launch"] / 1e6)) if k in kernels else "-"))
    for k in ("be_solve_kernel", "be_marg_kernel", "fe_lk_kernel"):
        if k in sq:
            print(k, {c: round(v) for c, v in sq[k].items()})


if __name__ == "__main__":
    main()
