cd $GRAFT_REPO_ROOT
rm -f gpurun_out/lag.txt
for cfg in "128 0" "128 1" "64 1" "16 1" "1 1" "192 1"; do set -- $cfg
  for fe in 0 64; do
    r=$(VIO_FE_CUS=$fe python tools/lag_rate.py --seqs $1 --lag $2 2>/dev/null | tail -1)
    echo "S=$1 lag=$2 FE_CUS=$fe $r" >> gpurun_out/lag.txt
  done
done
cat gpurun_out/lag.txt
