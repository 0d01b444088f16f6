B="python bench.py --repeats 1 --cpu-seqs 4 --cpu-procs 0 --pcie-steps 0 --stream-steps 0"
pr() { python -c "
import json,sys;d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]);k=d['kernels_ms'];print(sys.argv[1], round(d['value']),round(d['ms_per_step'],3),'ingest %.3f solve %.3f marg %.3f fe %.3f'%(k['be_ingest'],k['be_solve'],k['be_marg'],d['frontend_ms']), d['parity']['traj_rmse_hip_vs_oracle_m'] if d.get('parity') else None)" $1; }
python -m pytest tests/test_gpu_batch.py tests/test_gpu_pipeline.py tests/test_gpu_golden.py tests/test_gpu_stages.py -m gpu -q -x 2>&1 | tail -2
$B > gpurun_out/bq_a.json 2>/dev/null; pr gpurun_out/bq_a.json
