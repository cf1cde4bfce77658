#!/bin/bash
# round 6: the batch drivers that keep several solves in flight on several streams (gomp, sp at configs[4]; fr, ompr, srr at configs[1]),
# with the stand-alone sweep's LDS request raised so that its workgroups sit one (81 KiB) or two (54 KiB) to a CU and the rest queue
show() { python -c "
import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[2],round(d['value'],2),d.get('unit'),round(d['roofline'].get('frac',0),4) if isinstance(d.get('roofline'),dict) else '')" "$1" "$2"; }
mkdir -p gpurun_out
for wl in gomp sp gomp_single sp_single; do
  for lds in 0 81; do
    python bench.py --workload $wl --tune sweep_lds_kib=$lds > gpurun_out/res_tmp.json 2>/dev/null; show gpurun_out/res_tmp.json "$wl sweep_lds_kib=$lds"
  done
done
