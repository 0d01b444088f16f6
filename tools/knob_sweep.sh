#!/bin/bash
# Runs ON THE GPU BOX: short bench runs under different runtime knobs; one JSON line each into gpurun_out/sweep_<tag>.json
set -u
ARGS="--steps ${STEPS:-20} --warmup 6 --repeats 3 --cpu-seqs 0 --cpu-procs 0 --pcie-steps 0 --stream-steps 0 --aux 0"
run() { tag=$1; shift; env "$@" timeout 300 python bench.py $ARGS > gpurun_out/sweep_$tag.json 2> gpurun_out/sweep_$tag.err; python - "$tag" <<'P'
import json, sys
try:
    j = json.loads(open("gpurun_out/sweep_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
    print("%-14s %8.0f frames/s  %.3f ms/step  valid %s  fe %.2f be %.2f solve %.2f marg %.2f" % (sys.argv[1], j["value"], j["ms_per_step"], j["valid"], j["frontend_ms"], j["backend_ms"], j["kernels_ms"]["be_solve"], j["kernels_ms"]["be_marg"]))
except Exception as e:
    print(sys.argv[1], "failed", e)
P
}
mkdir -p gpurun_out
for spec in "$@"; do
    tag=${spec%%:*}; envs=${spec#*:}
    run "$tag" ${envs//,/ }
done
