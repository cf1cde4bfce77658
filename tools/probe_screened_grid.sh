#!/bin/bash
# kernel durations of the screened lone solve under several sweep grids (experiments build: CSMP_SCR_NBLK)
cd /tmp && export TMPDIR=/tmp
for g in ${GRIDS:-256 512}; do
 for dp in ${DEPTHS:-3}; do
 for nt in ${NTS:-1}; do
 for rv in ${REVS:-0}; do
  export CSMP_SCR_NBLK=$g CSMP_SCR_DEPTH=$dp CSMP_SCR_NT=$nt CSMP_SCR_REVERSE=$rv
  rm -rf /tmp/ps_$g
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$g -o t -- python3 $GRAFT_REPO_ROOT/tools/probe_screened.py 2 > /tmp/ps_$g.log 2>&1
  echo "== grid $g depth $dp nt $nt reverse $rv"; grep -E "\"screened\": 1|rror" /tmp/ps_$g.log | cut -c 1-160 | tail -n 3 | cut -c 1-110; grep equals_exact /tmp/ps_$g.log
  f=$(find /tmp/ps_$g -name "*kernel_stats.csv" | head -n 1)
  if [ -z "$f" ]; then tail -n 5 /tmp/ps_$g.log; find /tmp/ps_$g | head; continue; fi
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(t in n for t in ("k_sweep_bf16", "k_pick1")):
        print("  %-60s calls %6s avg %8.1f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
 done
 done
 done
done
