#!/bin/bash
# host-fed rate (bench.py's pcie_inclusive leg only) against the host-path knobs: one line per setting into gpurun_out/pcie.txt
run() { env "$@" python bench.py --seqs 128 --steps 10 --warmup 5 --repeats 1 --aux 0 --cpu-seqs 0 --cpu-procs 0 --stream-steps 0 --pcie-steps ${PS:-40} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['pcie_inclusive_frames_per_s'])" >> gpurun_out/pcie.txt; }
rm -f gpurun_out/pcie.txt
run A=base
run VIO_COPY_STREAMS=1
run A=base2
cat gpurun_out/pcie.txt
