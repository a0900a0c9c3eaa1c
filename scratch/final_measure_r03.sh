#!/bin/bash
# The measurements DESIGN.md / profiles/ quote for round 3, in one GPU call.  usage: scratch/final_measure_r03.sh TAG
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
echo "== driver command"; timeout -k 10 400 $B --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver.json 2> $O/${TAG}_bench_driver.err; cut -c1-300 $O/${TAG}_bench_driver.json
d=$O/${TAG}_prof_driver; rm -rf $d
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- $B --gpus 1 --steps 20 --warmup 5 --no-extras > $d.json 2> $d.err
cp $(find $d -name "*kernel_stats.csv" | head -1) $O/${TAG}_driver_cmd_kernel_stats.csv; head -6 $O/${TAG}_driver_cmd_kernel_stats.csv | cut -c1-170
echo "== 200 steps: two launches / three launches / classic"
timeout -k 10 200 $B --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/${TAG}_bench_200.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/${TAG}_bench_200.json'));print('two', d['value'], d['ms_per_step'], d['step_ms']['median'], d['roofline']['kernels_us'])"
timeout -k 10 200 $B --steps 200 --warmup 20 --no-cpu-baseline --no-extras --pic-three-launches > $O/${TAG}_bench_three.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/${TAG}_bench_three.json'));print('three', d['value'], d['ms_per_step'], d['step_ms']['median'], d['roofline']['kernels_us'])"
timeout -k 10 200 $B --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-pic > $O/${TAG}_bench_classic.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/${TAG}_bench_classic.json'));print('classic', d['value'], d['ms_per_step'], d['step_ms']['median'], d['roofline']['kernels_us'])"
for v in two three; do
  extra=""; [ $v = three ] && extra="--pic-three-launches"
  d=$O/${TAG}_prof_$v; rm -rf $d
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- $B --steps 100 --warmup 20 --no-cpu-baseline --no-extras --kernel-reps 2 $extra > $d.json 2> $d.err
  cp $(find $d -name "*kernel_stats.csv" | head -1) $O/${TAG}_${v}_kernel_stats.csv; head -5 $O/${TAG}_${v}_kernel_stats.csv | cut -c1-170
done
echo "== PMC traffic (two separate passes)"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  d=$O/pmc_${TAG}_$i; rm -rf $d
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- $B --steps 16 --warmup 8 --no-cpu-baseline --no-extras --kernel-reps 1 > $d.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $d.log; }
  i=$((i+1))
done
python3 $R/scratch/pmc_agg.py $O/pmc_${TAG}_* > $O/${TAG}_pmc_traffic_per_kernel_avg.json; grep -A6 "forward_move<float, 1, true, false, true\|resolve_diffuse" $O/${TAG}_pmc_traffic_per_kernel_avg.json | head -20
echo "== other sizes"
for cfg in "256 f32" "1024 f32" "2048 f32" "4096 f16" "8192 f32" "16384 f32" "16384 f16"; do
  set -- $cfg
  timeout -k 10 300 $B --size $1 --fields $2 --steps 100 --warmup 10 --no-cpu-baseline --no-extras --kernel-reps 1 > $O/${TAG}_size_$1_$2.json 2>/dev/null
  python3 -c "import json;d=json.load(open('$O/${TAG}_size_$1_$2.json'));print('$1 $2', d['value'], d['step_ms']['median'], d['config']['step_kind'][:28], d['roofline']['kernels_us'])"
done
echo "== replicas"
timeout -k 10 200 $B --replicas 16 --size 1024 --steps 200 > $O/${TAG}_rep16_1024.json 2>/dev/null; cut -c1-200 $O/${TAG}_rep16_1024.json; python3 -c "import json;d=json.load(open('$O/${TAG}_rep16_1024.json'));print(d['config'])"
timeout -k 10 200 $B --replicas 16 --size 256 --steps 300 > $O/${TAG}_rep16_256.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/${TAG}_rep16_256.json'));print(d['value'], d['config'])"
timeout -k 10 400 $B --replicas 2 --size 16384 --fields f16 --steps 30 --warmup 10 > $O/${TAG}_rep2_16384_f16.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/${TAG}_rep2_16384_f16.json'));print(d['value'], d['config'])"
echo "== phase stamps (diagnostic build)"
DIE_AMD_LIB=$R/scratch/libs/libdie_stamps.so timeout -k 10 300 python3 $R/scratch/pic_stamps.py > $O/${TAG}_phase_stamps.txt 2>&1; grep -v amdgpu $O/${TAG}_phase_stamps.txt
echo "== 2 ranks over gloo on the one GPU"; DIE_DIST_BACKEND=gloo timeout -k 10 300 $B --gpus 2 --steps 24 --warmup 8 --no-cpu-baseline > $O/${TAG}_bench_g2.json 2> $O/${TAG}_bench_g2.err; cut -c1-300 $O/${TAG}_bench_g2.json
echo "== one rank, decomposed path"; timeout -k 10 300 $B --force-dist --steps 40 --warmup 8 --no-cpu-baseline --no-extras > $O/${TAG}_bench_dist1.json 2>/dev/null; cut -c1-200 $O/${TAG}_bench_dist1.json
