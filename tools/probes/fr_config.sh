# Round 6 probe: block size and workgroup count of the forward-regression sweep after its body was brought up to the product sweep's
# (csmp_tune sweep_unit / sweep_grid reach fr_config).  Usage (GPU box): bash tools/probes/fr_config.sh
for t in "" sweep_unit=8,sweep_grid=192 sweep_unit=8,sweep_grid=224 sweep_unit=16,sweep_grid=176 sweep_unit=16,sweep_grid=192 sweep_unit=16,sweep_grid=208 sweep_unit=16,sweep_grid=224 sweep_unit=16,sweep_grid=256; do
  for w in fr srr; do
    python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline ${t:+--tune $t} 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', '${t:-default}', d['value'], r['frac'], r.get('kernel'), r.get('avg_launch_us'))"
  done
done
