# Round 6 probe: the screened (image) sweep with its column groups dealt out statically instead of by ticket counters
# (csmp_tune screen_static = 1).  Usage (GPU box): bash tools/probes/screen_static.sh
for w in "screened" "gomp_single --screened"; do
  for t in "" screen_static=1; do
    python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline ${t:+--tune $t} 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', '${t:-tickets}', round(d['value'],1), d['unit'], 'sweep', r.get('avg_launch_us'), 'us', r['frac'], d.get('matches_exact_path_on_sample'))"
  done
done
