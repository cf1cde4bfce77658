#!/bin/bash
# round 6: the headline bench with two pipelines side by side against one, over the LDS request (residency) and the sweep grid; ONE box
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[2],round(d['value'],1),round(d['roofline']['avg_launch_us'],2),d['roofline']['launches_timed'])" "$1" "$2"; }
mkdir -p gpurun_out
run() { python bench.py --no-secondary --cpu-seconds 1 --tune "$1" > gpurun_out/bench_tmp.json 2>/dev/null; show gpurun_out/bench_tmp.json "$1"; }
run pipelines=1
for split in 1 0; do for grid in 176 208 224 240 256; do run pipelines=2,pair_split=$split,tick_grid=$grid; done; done
run pipelines=1
run pipelines=2
