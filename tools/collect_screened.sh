#!/bin/bash
# collect_screened.sh <outdir> -- rocprofv3 evidence of the screened single-signal sweep (bench.py --workload screened): kernel
# statistics and the HBM fetch bytes of k_sweep_bf16 (one PMC pass, --kernel-trace only, under a timeout).
set -u
OUT=$(realpath "$1"); mkdir -p "$OUT"
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for cert in statistical rigorous; do
  rm -rf /tmp/prof_scr_$cert
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_scr_$cert -- python3 $R/bench.py --workload screened --batch-cert $cert > $OUT/screened_${cert}_line.json 2> $OUT/screened_${cert}.err
  f=$(find /tmp/prof_scr_$cert -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep -E '^"Name"|csmp::' "$f" > $OUT/screened_${cert}_kernel_stats.csv
  grep '^{' $OUT/screened_${cert}_line.json > $OUT/x.tmp && mv $OUT/x.tmp $OUT/screened_${cert}_line.json
done
# the int8 image, and GOMP at configs[4] on both images
rm -rf /tmp/prof_scr_i8
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_scr_i8 -- python3 $R/bench.py --workload screened --screen-image int8 > $OUT/screened_int8_line.json 2> $OUT/screened_int8.err
f=$(find /tmp/prof_scr_i8 -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -E '^"Name"|csmp::' "$f" > $OUT/screened_int8_kernel_stats.csv
grep '^{' $OUT/screened_int8_line.json > $OUT/x.tmp && mv $OUT/x.tmp $OUT/screened_int8_line.json
for img in bf16 int8; do
  rm -rf /tmp/prof_gscr_$img
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_gscr_$img -- python3 $R/bench.py --workload gomp_single --screened --screen-image $img > $OUT/gomp_screened_${img}_line.json 2> $OUT/gomp_screened_${img}.err
  f=$(find /tmp/prof_gscr_$img -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep -E '^"Name"|csmp::' "$f" > $OUT/gomp_screened_${img}_kernel_stats.csv
  grep '^{' $OUT/gomp_screened_${img}_line.json > $OUT/x.tmp && mv $OUT/x.tmp $OUT/gomp_screened_${img}_line.json
done
rm -rf /tmp/pmc_scr
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_scr -- python3 $R/bench.py --workload screened --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_screened.err
python3 $R/tools/pmc_traffic.py --kernel k_sweep_bf16 /tmp/pmc_scr > $OUT/screened_traffic.json
rm -rf /tmp/pmc_scr8
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_scr8 -- python3 $R/bench.py --workload screened --screen-image int8 --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_screened_int8.err
python3 $R/tools/pmc_traffic.py --kernel k_sweep_i8 /tmp/pmc_scr8 > $OUT/screened_int8_traffic.json
ls -la $OUT
