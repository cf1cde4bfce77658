#!/bin/bash
# collect_profiles.sh <outdir> -- the rocprofv3 evidence of a round, on the GPU box (gpurun): kernel-trace statistics of the default
# bench command and of the secondary workloads, and the PMC passes (separate runs, --kernel-trace only, as the pool requires).
# Run from the repository root; summaries land in <outdir> (copy the ones to keep into profiles/, named per round).
set -u
OUT=$(realpath "$1"); mkdir -p "$OUT"
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
stats() {  # stats <name> <bench args...>: headline line + csmp kernel rows of rocprofv3 --kernel-trace --stats
  local name=$1; shift
  rm -rf /tmp/prof_$name /tmp/w_$name; mkdir -p /tmp/w_$name; cd /tmp/w_$name
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 $R/bench.py "$@" > $OUT/${name}_stdout.txt 2> $OUT/${name}.err
  local f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep -E '^"Name"|csmp::|_ZN4csmp' "$f" > $OUT/${name}_kernel_stats.csv
  grep '^{' $OUT/${name}_stdout.txt | tail -1 > $OUT/${name}_line.json; rm -f $OUT/${name}_stdout.txt
  [ -f bench_secondary.json ] && cp bench_secondary.json $OUT/${name}_detail.json
  cd /tmp
}
pmc() {  # pmc <name> <counters> <bench args...> -> prints the output directory
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/pmc_$name
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$name -- python3 $R/bench.py "$@" > /dev/null 2> $OUT/pmc_${name}.err
  echo /tmp/pmc_$name
}
# the driver's command, unprofiled: the compact headline (last stdout line) and the detail file
mkdir -p /tmp/w_plain; cd /tmp/w_plain
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default_stdout.txt 2> $OUT/bench_default.err
tail -1 $OUT/bench_default_stdout.txt > $OUT/bench_default_line.json; cp bench_secondary.json $OUT/bench_default_detail.json; rm -f $OUT/bench_default_stdout.txt
cd /tmp
stats bench --steps 18 --warmup 3 --no-cpu-baseline --no-secondary
# two pipelines side by side: the sweep launches of the two streams overlap -- the bandwidth the kernel trace itself supports
python3 $R/tools/trace_union.py /tmp/prof_bench "k_tick<float, 16, false, true" 1073741824 > $OUT/bench_kernel_union.json 2> $OUT/bench_kernel_union.err
stats bench_one_pipeline --steps 18 --warmup 3 --no-cpu-baseline --no-secondary --tune pipelines=1
stats bench_batched --workload batched --steps 2 --warmup 1
stats bench_batched_gram --workload batched --steps 2 --warmup 1 --batch-gram
stats bench_sp --workload sp --steps 9 --warmup 3
stats bench_sp_single --workload sp_single --steps 3 --warmup 1
stats bench_gomp --workload gomp --steps 6 --warmup 2
stats bench_gomp_single --workload gomp_single --steps 2 --warmup 1
stats bench_screened_f16 --workload screened --steps 6 --warmup 2
stats bench_gomp_single_screened_f16 --workload gomp_single --steps 2 --warmup 1 --screened
stats bench_ompr --workload ompr --steps 3 --warmup 1 --no-in-flight
stats bench_srr --workload srr --steps 3 --warmup 1 --no-in-flight
stats bench_fr --workload fr --steps 6 --warmup 3
# the shape table of the product sweep (M = 256 .. 32768, f32 and f64): per-shape rows from the kernel trace
rm -rf /tmp/prof_shapes /tmp/w_shapes; mkdir -p /tmp/w_shapes; cd /tmp/w_shapes
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_shapes -- python3 $R/bench.py --workload shapes --steps 10 > $OUT/shapes_stdout.txt 2> $OUT/shapes.err
tail -1 $OUT/shapes_stdout.txt > $OUT/bench_shapes_line.json; rm -f $OUT/shapes_stdout.txt
cp $R/profiles/r06_sweep_shapes.json $OUT/sweep_shapes.json
python3 $R/tools/shape_profile.py /tmp/prof_shapes $OUT/sweep_shapes.json > $OUT/sweep_shapes_kernel_stats.csv
cd /tmp
# HBM traffic of the steady-state tick (two passes: the TCC block cannot hold both counters)
# (six signals: two pipelines, the tick's sweep a launch of its own -- the steady-state symbol names exactly those launches)
d1=$(pmc fetch FETCH_SIZE --steps 6 --warmup 0 --no-cpu-baseline --no-secondary)
d2=$(pmc write WRITE_SIZE --steps 6 --warmup 0 --no-cpu-baseline --no-secondary)
python3 $R/tools/pmc_traffic.py --kernel "k_tick<float, 16, false, true" $d1 $d2 > $OUT/sweep_traffic.json
# HBM fetch bytes of the binary16-image sweep
d5=$(pmc scr FETCH_SIZE --workload screened --steps 2 --warmup 1)
python3 $R/tools/pmc_traffic.py --kernel k_sweep_f16 $d5 > $OUT/screened_f16_traffic.json
# matrix-core counters of the screening kernel (binary16 operands: the default)
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
    out["note"] = "utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), per screening launch (k_b_screen256p<f16>, all 1024 signals of the batch)"
print(json.dumps(out, indent=1))
PY
# HBM fetch bytes of the batched path's kernels (default options)
d4=$(pmc bfetch FETCH_SIZE --workload batched --steps 1 --warmup 0)
python3 $R/tools/pmc_batched.py $d4 - > $OUT/batched_traffic.json
ls -la $OUT
