#!/bin/bash
# sp_trace.sh <out.csv> -- kernel trace (start / end per launch) of `bench.py --workload sp_single`, compacted to name,start,end [ns]
R=$(pwd); OUT=$(realpath $1); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/sptr
rocprofv3 --kernel-trace --output-format csv -d /tmp/sptr -- python3 $R/bench.py --workload sp --steps 9 --warmup 3 > /tmp/sp.out 2>/tmp/sp.err
f=$(find /tmp/sptr -name "*kernel_trace.csv" | head -1)
python3 - $f $OUT <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
with open(sys.argv[2],"w") as f:
    for r in rows:
        f.write("%s,%d,%d\n"%(r["Kernel_Name"].split("(")[0][:60].replace(",",";"),int(r["Start_Timestamp"])-t0,int(r["End_Timestamp"])-t0))
PY
