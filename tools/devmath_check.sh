#!/bin/bash
# Build (here or on the GPU box; hipcc cross-compiles) and run (GPU box) the device-vs-host bit check of the double-precision operations
# predictMotion relies on.  The binary is git-ignored (tools/*.bin) but travels with the gpurun snapshot.
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-result tools/devmath_check.hip -o tools/devmath_check.bin
./tools/devmath_check.bin
