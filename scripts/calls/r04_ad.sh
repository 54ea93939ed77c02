#!/bin/bash
# round 4, call ad: kernel trace of the steady loop at cfg3 (decision-bound): what sits between two decision kernels
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04_ad_trace -- python3 $R/scripts/arith_grid.py cfg3 "fused=1" 512 64 > $R/gpurun_out/r04_ad.log 2>&1
cd $R
tail -2 gpurun_out/r04_ad.log
T=$(find gpurun_out/r04_ad_trace -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$T")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[-90]["Start_Timestamp"])
for r in rows[-90:]:
    s=(int(r["Start_Timestamp"])-t0)/1e3; e=(int(r["End_Timestamp"])-t0)/1e3
    print("%9.1f -> %9.1f (%7.1f us) q%s %s"%(s,e,e-s,r.get("Queue_Id","?"),r["Kernel_Name"].split("(")[0][-46:]))
PY
