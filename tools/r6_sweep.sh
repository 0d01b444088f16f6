#!/bin/bash
# round-6 knob sweep (GPU box, from the repo root): python bench.py short form, one line per setting into gpurun_out/sweep.txt
run() { env "$@" python bench.py --seqs ${SEQS:-128} --steps 20 --warmup 5 --repeats ${REP:-10} --aux 0 --cpu-seqs 0 --cpu-procs 0 --pcie-steps 0 --stream-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('S=${SEQS:-128}', '$*', round(d['value']), d.get('valid'), round(d['solver']['mean_iterations'],2))" >> gpurun_out/sweep.txt; }
c5() { r=$(env "$@" python tools/config5_rate.py --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['frames_per_s']), d['valid'], d['kernels_ms']['be_solve'], d['kernels_ms']['be_marg'])"); echo "config5 $* $r" >> gpurun_out/sweep.txt; }
rm -f gpurun_out/sweep.txt
run A=base
run A=base2
c5 A=base
c5 A=base2
SEQS=512 REP=5 run A=base
