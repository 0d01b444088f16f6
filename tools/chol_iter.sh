#!/bin/bash
# Tuning loop of the tile Cholesky (HERE, then on the GPU box): rebuild only the stage translation unit, relink, run tools/chol_bench.py
set -e
cd /root/repo/vins-rgbd-fast_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $EXTRA -c stage_linalg.hip -o build/stage_linalg.hip.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libvio_hip.so build/*.o
cd /root/repo
/usr/local/graft/bin/gpurun --timeout 300 -- 'CHOL_MICRO=1 python tools/chol_bench.py 11 50 1; python tools/chol_bench.py 7 50 1; python tools/chol_bench.py 3 50 1' 2>&1 | grep -v "^\[gpurun\]" | tail -16
