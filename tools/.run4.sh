R=$(pwd); O=$R/gpurun_out/r05i; mkdir -p $O; : > $O/fr3.txt
python bench.py --workload srr --steps 4 --warmup 1 --no-in-flight | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('srr default', d['value'], d['roofline']['avg_launch_us'])" >> $O/fr3.txt
for cfg in "8 256" "8 320" "8 384" "8 448" "8 512" "16 256" "16 384" "16 512" "16 192"; do
  set -- $cfg
  CSMP_FR_U=$1 CSMP_FR_GRID=$2 python bench.py --workload fr --steps 6 --warmup 2 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fr U=$1 grid=$2', d['value'], d['roofline']['avg_launch_us'])" >> $O/fr3.txt
done
cat $O/fr3.txt
