#!/bin/bash
# collect_profiles.sh <outdir> -- the rocprofv3 evidence of a round, on the GPU box (gpurun): kernel-trace statistics of the
# default bench command and of the secondary workloads, and the PMC passes (separate runs, --kernel-trace only, as the
# pool requires).  Run from the repository root; summaries land in <outdir> (copy the ones to keep into profiles/).
set -u
OUT=$(realpath "$1"); mkdir -p "$OUT"
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
stats() {  # stats <name> <bench args...>
  local name=$1; shift
  rm -rf /tmp/prof_$name
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 $R/bench.py "$@" > $OUT/${name}_line.json 2> $OUT/${name}.err
  local f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep -E '^"Name"|csmp::' "$f" > $OUT/${name}_kernel_stats.csv
  grep '^{' $OUT/${name}_line.json > $OUT/${name}_line.tmp && mv $OUT/${name}_line.tmp $OUT/${name}_line.json
}
pmc() {  # pmc <name> <counters> <bench args...>
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$name -- python3 $R/bench.py "$@" > /dev/null 2> $OUT/pmc_${name}.err
  echo /tmp/pmc_$name
}
stats bench --steps 18 --warmup 3 --no-cpu-baseline --no-secondary
stats bench_batched --workload batched --steps 2 --warmup 1 --batch-screen bf16
stats bench_batched_gram --workload batched --steps 2 --warmup 1 --batch-gram --batch-screen bf16
stats bench_batched_rigorous --workload batched --steps 2 --warmup 1 --batch-cert rigorous
stats bench_batched_int8 --workload batched --steps 2 --warmup 1 --batch-screen int8
stats bench_batched_int8_gram --workload batched --steps 2 --warmup 1 --batch-screen int8 --batch-gram
stats bench_sp --workload sp --steps 9 --warmup 3
stats bench_sp_single --workload sp_single --steps 3 --warmup 1
stats bench_gomp --workload gomp --steps 6 --warmup 2
stats bench_gomp_single --workload gomp_single --steps 2 --warmup 1
# HBM traffic of the steady-state tick (two passes: the TCC block cannot hold both counters)
d1=$(pmc fetch FETCH_SIZE --steps 3 --warmup 0 --no-cpu-baseline --no-secondary)
d2=$(pmc write WRITE_SIZE --steps 3 --warmup 0 --no-cpu-baseline --no-secondary)
python3 $R/tools/pmc_traffic.py $d1 $d2 > $OUT/sweep_traffic.json
# matrix-core counters of the screening kernel
d3=$(pmc mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --workload batched --steps 1 --warmup 0)
python3 - "$d3" > $OUT/batched_mfma_pmc.json <<'PY'
import csv, glob, json, os, sys
acc = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_b_screen" not in row.get("Kernel_Name", ""):
            continue
        acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
out = {k: {"dispatches": len(v), "avg": sum(v) / len(v)} for k, v in acc.items()}
if "SQ_VALU_MFMA_BUSY_CYCLES" in out and "GRBM_GUI_ACTIVE" in out:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; MFMA busy cycles over the 256 CUs x 4 SIMDs
    out["mfma_util"] = out["SQ_VALU_MFMA_BUSY_CYCLES"]["avg"] / (out["GRBM_GUI_ACTIVE"]["avg"] / 8.0 * 1024.0)
    out["note"] = "utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), per screening launch (k_b_screen256p, all 1024 signals of the batch)"
print(json.dumps(out, indent=1))
PY
# HBM traffic of the batched path's kernels (default options and with the Gram matrix): FETCH_SIZE only -- the WRITE_SIZE pass of
# this workload hung under the profiler in round 3 (tools/pmc_batched.py)
d4=$(pmc bfetch FETCH_SIZE --workload batched --steps 1 --warmup 0)
python3 $R/tools/pmc_batched.py $d4 - > $OUT/batched_traffic.json
d6=$(pmc bgfetch FETCH_SIZE --workload batched --steps 1 --warmup 0 --batch-gram)
python3 $R/tools/pmc_batched.py $d6 - gram > $OUT/batched_gram_traffic.json
ls -la $OUT
