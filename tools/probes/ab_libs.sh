# Round 6 probe: A/B of TWO builds of libcsmp.so on one GPU box, twice each (boxes of the pool differ by 1-2 %, builds by less).
# Before the gpurun call: keep the build to compare against as compressedsensing.jl_amd/csrc/libcsmp_prev.so (it travels with the
# snapshot; *.so is git-ignored), build the new one in place.  Usage (GPU box): bash tools/probes/ab_libs.sh
set -u
L=compressedsensing.jl_amd/csrc
O=gpurun_out/r06/ab_defer.txt
mkdir -p gpurun_out/r06
cp $L/libcsmp.so /tmp/new.so
: > $O
for round in 1 2; do
  for which in new prev; do
    if [ $which = new ]; then cp /tmp/new.so $L/libcsmp.so; else cp $L/libcsmp_prev.so $L/libcsmp.so; fi
    echo "== $which (round $round)" >> $O
    python bench.py --steps 18 --warmup 6 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('bench', d['value'], d['roofline']['frac'], d['roofline'].get('avg_launch_us'), d['roofline'].get('launch_duration_us'))
" >> $O 2>&1
    python tools/sweep_shapes.py --M 256,512,1000 --no-check 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('sweep', d['M'], d['dtype'], d['us'], d['frac'])
" >> $O 2>&1
  done
done
cp /tmp/new.so $L/libcsmp.so
cat $O
