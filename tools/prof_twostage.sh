R=$(pwd); O=$R/gpurun_out/r05; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wl in ompr srr; do
  rm -rf /tmp/prof_$wl /tmp/w_$wl; mkdir -p /tmp/w_$wl; cd /tmp/w_$wl
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --no-in-flight $TUNE > $O/${wl}_stdout.txt 2> $O/${wl}.err
  f=$(find /tmp/prof_$wl -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep -E '^"Name"|csmp::|_ZN4csmp' "$f" > $O/r05_${wl}_kernel_stats_before.csv
  tail -1 $O/${wl}_stdout.txt > $O/r05_${wl}_line_before.json
  cd /tmp
done
