B="python bench.py --repeats 1 --cpu-seqs 0 --cpu-procs 0 --pcie-steps 0 --stream-steps 0"
pr() { python -c "
import json,sys;d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]);k=d['kernels_ms'];print(sys.argv[1], round(d['value']),round(d['ms_per_step'],3),'ingest %.3f solve %.3f marg %.3f fe %.3f'%(k['be_ingest'],k['be_solve'],k['be_marg'],d['frontend_ms']))" $1; }
$B > gpurun_out/bq_a.json 2>/dev/null; pr gpurun_out/bq_a.json
VIO_BE_CU_ALL=1 $B > gpurun_out/bq_b.json 2>/dev/null; pr gpurun_out/bq_b.json
VIO_BE_CU_SPLIT=1 $B > gpurun_out/bq_c.json 2>/dev/null; pr gpurun_out/bq_c.json
VIO_SERIAL_THREADS=1024 $B > gpurun_out/bq_d.json 2>/dev/null; pr gpurun_out/bq_d.json
VIO_SERIAL_THREADS=1024 VIO_FE_CUS=96 $B > gpurun_out/bq_e.json 2>/dev/null; pr gpurun_out/bq_e.json
VIO_SERIAL_THREADS=1024 VIO_BE_CU_SPLIT=1 $B > gpurun_out/bq_f.json 2>/dev/null; pr gpurun_out/bq_f.json
python tools/phase_profile.py 2>&1 | grep -E "ser "
python -m pytest tests/test_gpu_batch.py tests/test_gpu_pipeline.py -m gpu -q -x 2>&1 | tail -2
