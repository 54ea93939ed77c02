#!/bin/bash
# round 3: timeline of the default cfg4 loop (kernel trace): what sits between two sweeps on the sweep stream
R=$PWD; OUT=$R/gpurun_out/r03_ak; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-cfg3 --no-parity --no-steady --no-onepass --steps 256 --warmup 64 > $OUT/trace.log 2>&1; echo "trace rc=$?"
cd $R
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python - "$T" <<'PY'
import csv,sys
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0].replace('void lpxk::','').replace('lpxk::','')[:28], r.get('Queue_Id','?')))
rows.sort()
sw=[i for i,r in enumerate(rows) if r[2].startswith('k_sweep32_pull')]
sw=sw[len(sw)//2:]   # the timed half
print("sweeps", len(sw))
import statistics as st
per=[(rows[sw[k+1]][0]-rows[sw[k]][0])/1e3 for k in range(len(sw)-1)]
dur=[(rows[i][1]-rows[i][0])/1e3 for i in sw]
print("sweep start to next sweep start: mean %.1f us; sweep duration mean %.1f us; idle between sweeps %.1f us" % (st.mean(per), st.mean(dur), st.mean(per)-st.mean(dur[:-1])))
# what runs between the end of sweep k and the start of sweep k+1 (any queue)
k=sw[len(sw)//2]; e=rows[k][1]; nxt=rows[sw[len(sw)//2+1]][0]
print("between one sweep's end and the next sweep's start (%.1f us):" % ((nxt-e)/1e3))
for r in rows:
    if r[1] > e and r[0] < nxt and not r[2].startswith('k_sweep32_pull'):
        print("   %-28s q%s  start %+8.1f us  dur %7.1f us" % (r[2], r[3], (r[0]-e)/1e3, (r[1]-r[0])/1e3))
PY
