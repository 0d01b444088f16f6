#!/bin/bash
# Kernel trace of the canonical bench (GPU box, repo root): every launch of one stream group's back-end chain for one step, with durations
#   bash tools/chain_profile.sh [seqs]   ->  gpurun_out/chain_s<seqs>.txt
S=${1:-128}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/chprof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/chprof -o ch -- python bench.py --seqs $S --steps 20 --warmup 5 --repeats 1 --aux 0 --cpu-seqs 0 --cpu-procs 0 --pcie-steps 0 --stream-steps 0 > gpurun_out/chprof.log 2>&1
python tools/trace_chain.py $(ls gpurun_out/chprof/*/*kernel_trace.csv gpurun_out/chprof/*kernel_trace.csv 2>/dev/null | head -1) 3 -v > gpurun_out/chain_s$S.txt 2>&1
rm -rf gpurun_out/chprof
head -64 gpurun_out/chain_s$S.txt
