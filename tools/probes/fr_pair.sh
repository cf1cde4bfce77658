# Round 6 probe: csmp_fr_batch with two pipelines side by side (csmp_tune pipelines = 2), with and without the LDS request that holds
# the ticks to one workgroup per CU.  Usage (GPU box): bash tools/probes/fr_pair.sh
for t in "" pipelines=2 pipelines=2,pair_lds_kib=81 pipelines=2,pair_lds_kib=120; do
  for n in 6 18; do
    python bench.py --workload fr --steps $n --warmup 2 --no-cpu-baseline ${t:+--tune $t} 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']; print('fr', $n, 'signals', '${t:-default}', d['value'], r['frac'], r.get('avg_launch_us'))"
  done
done
python -m pytest tests -m gpu -q -k "fr_batch or fr_matches or forward_regression" 2>&1 | grep -E "passed|failed" | tail -1
