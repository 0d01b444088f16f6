#!/bin/bash
# Runs ON THE GPU BOX (via gpurun) from the repo root: rocprofv3 passes over the bench command, then summaries into gpurun_out/.
#   pass 1: --kernel-trace --stats            (per-kernel durations; must agree with bench.py's HIP-event numbers)
#   pass 2..4: --pmc only (separate passes: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2; SQ counters in their own pass)
# Copy gpurun_out/profile_summary/* into profiles/ afterwards (gpurun_out is scratch).
set -u
ROUND=${ROUND:-round4}
REPO=$(pwd)
export TMPDIR=/tmp
OUT=$REPO/gpurun_out/prof_$ROUND
rm -rf "$OUT"; mkdir -p "$OUT" "$REPO/gpurun_out/profile_summary"
PASSES=${PASSES:-"1 2 3 4 5"}   # e.g. PASSES="1" refreshes only the kernel-trace statistics
has() { [[ " $PASSES " == *" $1 "* ]]; }
BENCH="python $REPO/bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-6} --repeats 1 --cpu-seqs 0 --cpu-procs 0 --pcie-steps 0 --stream-steps 0 --aux 0 --seqs ${SEQS:-128}"
cd /tmp
has 1 && echo "== pass 1: kernel trace + stats" && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $BENCH > "$OUT/stats.log" 2>&1; tail -1 "$OUT/stats.log" | cut -c1-300
has 2 && echo "== pass 2: FETCH_SIZE" && timeout 300 rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch -- $BENCH > "$OUT/fetch.log" 2>&1; tail -1 "$OUT/fetch.log" | cut -c1-200
has 3 && echo "== pass 3: WRITE_SIZE" && timeout 300 rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/write" -o write -- $BENCH > "$OUT/write.log" 2>&1; tail -1 "$OUT/write.log" | cut -c1-200
has 4 && echo "== pass 4: SQ" && timeout 300 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU -d "$OUT/sq" -o sq -- $BENCH > "$OUT/sq.log" 2>&1; tail -1 "$OUT/sq.log" | cut -c1-200
has 5 && echo "== pass 5: L2" && timeout 300 rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/l2" -o l2 -- $BENCH > "$OUT/l2.log" 2>&1; tail -1 "$OUT/l2.log" | cut -c1-200
cd "$REPO"
find "$OUT" -name "*.csv" | head -40
python profiles/summarize.py "$OUT" "$REPO/gpurun_out/profile_summary" "$ROUND" "${SEQS:-128}"
