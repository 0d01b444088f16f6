#!/bin/bash
# configs[4] rate against the stream-group size (GPU box): one line per setting into gpurun_out/c5_groups.txt
rm -f gpurun_out/c5_groups.txt
for q in 8 16; do for g in 32 16 8 4; do
  r=$(GPU_MAX_HW_QUEUES=$q VIO_GROUP_SEQS=$g python tools/config5_rate.py --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['frames_per_s']), d['valid'], d['kernels_ms']['be_solve'], d['kernels_ms']['be_marg'])")
  echo "queues=$q group=$g $r" >> gpurun_out/c5_groups.txt
done; done
cat gpurun_out/c5_groups.txt
