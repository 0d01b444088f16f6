#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel statistics of ONE stream group of n sequences (no second group beside it), n = 16 / 64 / 128:
# how each kernel of the chain grows with the number of sequences it covers
R=$(pwd); mkdir -p $R/gpurun_out/iso; cd /tmp; export TMPDIR=/tmp
ARGS="--steps 20 --warmup 6 --repeats 2 --cpu-seqs 0 --cpu-procs 0 --pcie-steps 0 --stream-steps 0 --aux 0"
for n in ${NS:-16 64 128}; do
  VIO_GROUP_SEQS=$n timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/iso/n$n -o p -- python $R/bench.py $ARGS --seqs $n > $R/gpurun_out/iso/n$n.log 2>&1
  f=$(find $R/gpurun_out/iso/n$n -name "*kernel_stats.csv" | head -1)
  echo "== n=$n  $f"; tail -2 $R/gpurun_out/iso/n$n.log | cut -c1-200
  python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if r['Name'].startswith(('ps_','be_','fe_')): print("%-28s calls %6s avg %9.1f us"%(r['Name'][:28], r['Calls'], float(r['AverageNs'])/1e3))
P
  find $R/gpurun_out/iso/n$n -name "*.csv" ! -name "*kernel_stats.csv" -delete
done
