cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2s
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2s/stats -- python bench.py --global-batch 16 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/r2s/prof.log 2>&1
find gpurun_out/r2s/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r2s/kernel_stats_bs16.csv \;
find gpurun_out/r2s/stats -name "*kernel_trace.csv" -exec cp {} gpurun_out/r2s/kernel_trace_bs16.csv \;
rm -rf gpurun_out/r2s/stats
tail -1 gpurun_out/r2s/prof.log | cut -c1-200
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r2s/kernel_trace_bs16.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 2 steps worth: compute GPU idle gaps between consecutive kernels
n=len(rows); sel=rows[n//2:]
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in sel)
span=int(sel[-1]['End_Timestamp'])-int(sel[0]['Start_Timestamp'])
gaps=[int(b['Start_Timestamp'])-int(a['End_Timestamp']) for a,b in zip(sel,sel[1:])]
print("kernels",len(sel),"span ms",span/1e6,"busy ms",busy/1e6,"gap ms",sum(g for g in gaps if g>0)/1e6, "mean gap us", sum(g for g in gaps if g>0)/len(gaps)/1e3)
PY
