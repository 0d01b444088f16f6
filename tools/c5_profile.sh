cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python tools/config5_rate.py --steps 20 > gpurun_out/c5_rate.txt 2>&1

rm -rf gpurun_out/c5prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5prof -o c5 -- python tools/config5_rate.py --steps 20 > gpurun_out/c5_prof.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/c5prof/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
out=open('gpurun_out/c5_stats.txt','w')
for r in rows[:25]:
    out.write("%-60s calls %6s avg %10.1f us total %8.1f ms pct %s\n"%(r['Name'][:60],r['Calls'],float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/1e6,r['Percentage']))
PY
python tools/trace_chain.py $(ls gpurun_out/c5prof/*/*kernel_trace.csv gpurun_out/c5prof/*kernel_trace.csv 2>/dev/null | head -1) 3 -v > gpurun_out/c5_chain.txt 2>&1
rm -rf gpurun_out/c5prof
cat gpurun_out/c5_chain.txt
tail -3 gpurun_out/c5_rate.txt; cat gpurun_out/c5_stats.txt
