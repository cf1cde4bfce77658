#!/bin/bash
# sp_trace.sh <out.csv> [sp_single|sp] -- kernel trace (start / end of every launch) of `bench.py --workload sp_single` (default) or
# `--workload sp` (csmp_sp_batch), compacted to "name,start,end" in nanoseconds from the first launch.  Run from the repository
# root on the GPU box; profiles/r04_sp_single_trace.csv and r04_sp_batch_trace_before_gate.csv came out of it.
R=$(pwd); OUT=$(realpath $1); W=${2:-sp_single}
if [ "$W" = sp ]; then ARGS="--workload sp --steps 9 --warmup 3"; else ARGS="--workload sp_single --steps 3 --warmup 1"; fi
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/sptr
rocprofv3 --kernel-trace --output-format csv -d /tmp/sptr -- python3 $R/bench.py $ARGS > /tmp/sp.out 2>/tmp/sp.err
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
