run() { env "$@" python bench.py --steps 20 --warmup 5 --aux 0 --cpu-seqs 0 --cpu-procs 0 --pcie-steps 0 --stream-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['value']), d.get('valid'))" >> gpurun_out/sweep.txt; }
rm -f gpurun_out/sweep.txt
run A=base
run VIO_EVAL_RPT=1
run VIO_EVAL_RPT=3
run VIO_EXTRA_SLOTS=1
run VIO_ASM_B_BLOCKS=12
run VIO_ASM_B_BLOCKS=48
run VIO_XCD_MAP=0
run VIO_GRAPH=1
run A=base2
