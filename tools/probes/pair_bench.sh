#!/bin/bash
# round 6: the headline bench with two pipelines side by side (default) against one (csmp_tune pipelines=1)
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1],d['value'],d['roofline']['avg_launch_us'],d['roofline']['launches_timed'])" "$1"; }
mkdir -p gpurun_out
for i in 1 2; do python bench.py --no-secondary --cpu-seconds 1 > gpurun_out/bench_pair_$i.json 2> /dev/null; show gpurun_out/bench_pair_$i.json; done
python bench.py --no-secondary --cpu-seconds 1 --tune pipelines=1 > gpurun_out/bench_one.json 2>/dev/null; show gpurun_out/bench_one.json
python bench.py --no-secondary --cpu-seconds 1 --steps 36 --warmup 6 > gpurun_out/bench_pair_36.json 2>/dev/null; show gpurun_out/bench_pair_36.json
