#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for n in 4096 16384 32768 65536 131072; do
  rm -rf /tmp/pn_$n
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pn_$n -o t -- python3 $GRAFT_REPO_ROOT/tools/probe_screened_n.py $n > /tmp/pn_$n.log 2>&1
  echo "== N $n"; grep -E "^[01] " /tmp/pn_$n.log
  f=$(find /tmp/pn_$n -name "*kernel_stats.csv" | head -n 1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(t in n for t in ("k_sweep", "k_pick1", "k_qr1", "k_qr2")):
        print("  %-50s calls %6s avg %8.1f us" % (n[:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
