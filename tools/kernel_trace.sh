#!/bin/bash
# kernel_trace.sh <out.csv> [workload] [steps] [warmup] -- start / end of every launch of `bench.py --workload <workload>` (default
# sp_single 3 1; `sp` = csmp_sp_batch), compacted to "name,start,end" in nanoseconds from the first launch.  Run from the
# repository root on the GPU box; profiles/r04_sp_single_trace.csv and r04_sp_batch_trace_before_gate.csv came out of it.
R=$(pwd); OUT=$(realpath $1); W=${2:-sp_single}; ST=${3:-3}; WU=${4:-1}
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/ktr
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktr -- python3 $R/bench.py --workload $W --steps $ST --warmup $WU > /tmp/ktr.out 2>/tmp/ktr.err
tail -1 /tmp/ktr.out | cut -c1-300
f=$(find /tmp/ktr -name "*kernel_trace.csv" | head -1)
python3 - $f $OUT <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[0]["Start_Timestamp"])
with open(sys.argv[2],"w") as f:
    for r in rows:
        f.write("%s,%d,%d\n"%(r["Kernel_Name"].split("(")[0][:60].replace(",",";"),int(r["Start_Timestamp"])-t0,int(r["End_Timestamp"])-t0))
PY
