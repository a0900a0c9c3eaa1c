#!/bin/bash
# The measurements DESIGN.md / profiles/ quote, in one GPU call.  usage: scratch/final_measure.sh TAG
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
echo "== driver command"; timeout -k 10 300 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver.json 2> $O/${TAG}_bench_driver.err; cut -c1-400 $O/${TAG}_bench_driver.json
d=$O/${TAG}_prof_driver; rm -rf $d
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $d.json 2> $d.err
cp $(find $d -name "*kernel_stats.csv" | head -1) $O/${TAG}_driver_cmd_kernel_stats.csv; head -6 $O/${TAG}_driver_cmd_kernel_stats.csv | cut -c1-160
echo "== classic path"; timeout -k 10 200 python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pic > $O/${TAG}_bench_classic.json 2>/dev/null; cut -c1-200 $O/${TAG}_bench_classic.json
echo "== long run"; timeout -k 10 200 python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/${TAG}_bench_200.json 2>/dev/null; cut -c1-200 $O/${TAG}_bench_200.json
for v in pic classic; do
  extra=""; [ $v = classic ] && extra="--no-pic"
  d=$O/${TAG}_prof_$v; rm -rf $d
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --kernel-reps 2 $extra > $d.json 2> $d.err
  cp $(find $d -name "*kernel_stats.csv" | head -1) $O/${TAG}_${v}_kernel_stats.csv; head -6 $O/${TAG}_${v}_kernel_stats.csv | cut -c1-160
done
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  d=$O/pmc_${TAG}_$i; rm -rf $d
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python3 $R/bench.py --steps 16 --warmup 8 --no-cpu-baseline --kernel-reps 1 > $d.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $d.log; }
  i=$((i+1))
done
python3 $R/scratch/pmc_agg.py $O/pmc_${TAG}_* > $O/${TAG}_pmc_traffic_per_kernel_avg.json; cat $O/${TAG}_pmc_traffic_per_kernel_avg.json | head -40
echo "== other sizes"
for cfg in "256 f32" "1024 f32" "2048 f32" "4096 f16" "8192 f32" "16384 f32" "16384 f16"; do
  set -- $cfg
  timeout -k 10 300 python3 $R/bench.py --size $1 --fields $2 --steps 100 --warmup 10 --no-cpu-baseline --kernel-reps 1 > $O/${TAG}_size_$1_$2.json 2>/dev/null
  python3 -c "import json;d=json.load(open('$O/${TAG}_size_$1_$2.json'));print('$1 $2', d['value'], d['step_ms']['median'], d['roofline']['kernels_us'])"
done
echo "== replicas"; timeout -k 10 200 python3 $R/bench.py --replicas 16 --size 1024 --steps 200 > $O/${TAG}_rep16_1024.json 2>/dev/null; cut -c1-600 $O/${TAG}_rep16_1024.json
timeout -k 10 200 python3 $R/bench.py --replicas 16 --size 256 --steps 300 > $O/${TAG}_rep16_256.json 2>/dev/null; cut -c1-600 $O/${TAG}_rep16_256.json
echo "== 2 ranks over gloo on the one GPU"; DIE_DIST_BACKEND=gloo timeout -k 10 300 python3 $R/bench.py --gpus 2 --steps 20 --warmup 5 > $O/${TAG}_bench_g2.json 2> $O/${TAG}_bench_g2.err; cut -c1-700 $O/${TAG}_bench_g2.json
