#!/bin/bash
# Runs ON THE GPU BOX: fe_lk with / without the XCD-aware block map (VIO_FE_XCD_MAP): bench rate, then FETCH_SIZE and L2 hit rate of the
# front-end kernels from two separate rocprofv3 --pmc passes (FETCH_SIZE takes three of the four TCC slots).  bash tools/lk_xcd_experiment.sh
export TMPDIR=/tmp
R=$(pwd)
for cfgname in ${CONFIGS:-nopart nopart_xcd}; do
  case $cfgname in part64) E="VIO_FE_CUS=64";; nopart) E="VIO_FE_CUS=0";; nopart_xcd) E="VIO_FE_CUS=0 VIO_FE_XCD_MAP=1";; part64_xcd) E="VIO_FE_CUS=64 VIO_FE_XCD_MAP=1";; esac
  echo "== $cfgname"
  env $E timeout 200 python bench.py --steps 20 --warmup 5 --aux 0 --cpu-seqs 0 --cpu-procs 0 --pcie-steps 0 --stream-steps 0 --repeats 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['frontend_ms'], d['backend_ms'])"
  B="python $R/bench.py --steps 10 --warmup 4 --repeats 1 --cpu-seqs 0 --cpu-procs 0 --pcie-steps 0 --stream-steps 0 --aux 0"
  (cd /tmp && env $E timeout 150 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/lkx/$cfgname/f -o p -- $B > /dev/null 2>&1)
  (cd /tmp && env $E timeout 150 rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/lkx/$cfgname/l -o p -- $B > /dev/null 2>&1)
  python - "$R/gpurun_out/lkx/$cfgname" <<'P'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in ("fe_lk_kernel", "fe_pyrdown_kernel", "fe_fast_kernel"):
    d = acc.get(k)
    if d and d.get("FETCH_SIZE") and d.get("TCC_HIT_sum"):
        n = len(d["FETCH_SIZE"]) // 2
        fs = sum(d["FETCH_SIZE"][n:]) / max(1, len(d["FETCH_SIZE"]) - n)
        hit = sum(d["TCC_HIT_sum"][n:]); mis = sum(d["TCC_MISS_sum"][n:])
        print("  %-20s fetch %.1f MB (x2 corrected) per launch, L2 hit %.2f" % (k, 2 * fs / 1024, hit / max(1, hit + mis)))
P
done
