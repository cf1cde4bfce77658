# Round 6 probe: ompr / srr with three solves in flight on three streams (api.solve_in_flight), their stand-alone sweeps with and
# without the one-workgroup-per-CU LDS request (csmp_tune sweep_lds_kib).  Usage (GPU box): bash tools/probes/inflight_residency.sh
for w in ompr srr; do
  for t in "" sweep_lds_kib=81; do
    python bench.py --workload $w --steps 9 --warmup 1 --no-cpu-baseline ${t:+--tune $t} 2>/dev/null | tail -1 > /tmp/l.json
    python - "$w" "${t:-default}" <<'PY'
import json, sys
d = json.load(open('/tmp/l.json'))
try:
    det = json.load(open('bench_secondary.json'))
except Exception:
    det = d
t3 = (det.get('three_in_flight') or d.get('three_in_flight') or {})
print(sys.argv[1], sys.argv[2], 'one at a time', round(d['value'], 2), 'solves/s; three in flight', t3.get('solves_per_s'))
PY
  done
done
