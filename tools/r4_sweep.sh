#!/bin/bash
# Runs ON THE GPU BOX: short bench runs under runtime knobs, one line each.  bash tools/r4_sweep.sh [set]
run() { tag=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --aux 0 --cpu-seqs 0 --cpu-procs 0 --pcie-steps 0 --stream-steps 0 --repeats 6 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag', round(d['value']), [round(x) for x in d['repeats']['frames_per_s']], 'fe', d['frontend_ms'], 'be', d['backend_ms'])"; }
case "${1:-cus}" in
cus)
  run base A=1
  run fe32 VIO_FE_CUS=32
  run fe96 VIO_FE_CUS=96
  run slots1 VIO_EXTRA_SLOTS=1
  run nopart VIO_FE_CUS=0 ;;
groups)
  run g43 VIO_GROUP_SEQS=43
  run g43_nopart VIO_GROUP_SEQS=43 VIO_FE_CUS=0
  run g32_nopart VIO_GROUP_SEQS=32 VIO_FE_CUS=0
  run g128_nopart VIO_GROUP_SEQS=128 VIO_FE_CUS=0
  run g64_q16 GPU_MAX_HW_QUEUES=16 VIO_FE_CUS=0 ;;
esac
