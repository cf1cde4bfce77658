# Round 6 probe: csmp_omp_batch with 2..7 signals -- one pipeline of up to three signals against two pipelines side by side
# (csmp_tune pipelines = 1 / 2).  Usage (GPU box): bash tools/probes/few_signals.sh
for n in 2 3 4 5 6 7 8 10 20; do
  for p in 1 0; do
    python bench.py --steps $n --warmup 3 --no-secondary --no-cpu-baseline --tune pipelines=$p 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('signals', d['steps'], 'pipelines', $p, 'atoms/s', d['value'], 'frac', d['roofline']['frac'], 'in flight', d['roofline'].get('launches_in_flight'))
"
  done
done
