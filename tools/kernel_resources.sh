#!/bin/bash
# VGPR / SGPR / scratch / spill counts of every gfx950 kernel in libvio_hip.so (from the code-object notes).  No GPU needed.
set -e
LIB=${1:-$(dirname "$0")/../vins-rgbd-fast_amd/libvio_hip.so}
TMP=$(mktemp -d)
cp "$LIB" "$TMP/lib.so"
(cd "$TMP" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null 2>&1)
for f in "$TMP"/lib.so.*gfx950; do
    /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" 2>/dev/null | grep -E "\.name:|\.vgpr_count|private_segment_fixed_size|\.sgpr_count|vgpr_spill_count|group_segment_fixed_size" | paste - - - - - -
done | sed 's/  */ /g; s/\.private_segment_fixed_size/scratch/; s/\.group_segment_fixed_size/lds/; s/\.vgpr_spill_count/spill/; s/\.vgpr_count/vgpr/; s/\.sgpr_count/sgpr/; s/\.name: //' | sort
rm -rf "$TMP"
