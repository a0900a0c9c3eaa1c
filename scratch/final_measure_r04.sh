#!/bin/bash
# The measurements DESIGN.md / profiles/ quote for round 4.  usage (GPU box): scratch/final_measure_r04.sh TAG a|b
#   a: the driver's command (bench line, rocprofv3 kernel stats), 200-step runs of the three step forms, PMC traffic and unit counters
#   b: other sizes, replicas, phase stamps (diagnostic build), decomposed runs over gloo on the one GPU
TAG=$1; PART=$2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
line() { python3 -c "import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{\"metric\"')][-1]);print('$2', d['value'], d['ms_per_step'], d.get('step_ms',{}).get('median'), d.get('roofline',{}).get('kernels_us'))"; }
if [ "$PART" = a ]; then
  echo "== driver command"; timeout -k 10 400 $B --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver.json 2> $O/${TAG}_bench_driver.err; cut -c1-200 $O/${TAG}_bench_driver.json
  d=$O/${TAG}_prof_driver; rm -rf $d
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- $B --gpus 1 --steps 20 --warmup 5 --no-extras > $d.json 2> $d.err
  cp $(find $d -name "*kernel_stats.csv" | head -1) $O/${TAG}_driver_cmd_kernel_stats.csv; head -6 $O/${TAG}_driver_cmd_kernel_stats.csv | cut -c1-170
  echo "== 200 steps: two launches / three launches / classic"
  timeout -k 10 200 $B --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/${TAG}_bench_200.json 2>/dev/null; line $O/${TAG}_bench_200.json two
  timeout -k 10 200 $B --steps 200 --warmup 20 --no-cpu-baseline --no-extras --pic-three-launches > $O/${TAG}_bench_three.json 2>/dev/null; line $O/${TAG}_bench_three.json three
  timeout -k 10 200 $B --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-pic > $O/${TAG}_bench_classic.json 2>/dev/null; line $O/${TAG}_bench_classic.json classic
  echo "== PMC traffic (two separate passes)"
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    d=$O/pmc_${TAG}_$i; rm -rf $d
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- $B --steps 16 --warmup 8 --no-cpu-baseline --no-extras --kernel-reps 1 > $d.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $d.log; }
    i=$((i+1))
  done
  python3 $R/scratch/pmc_agg.py $O/pmc_${TAG}_[0-9] > $O/${TAG}_pmc_traffic_per_kernel_avg.json; grep -A7 "forward_move<float, 1, true, false, true\|resolve_diffuse" $O/${TAG}_pmc_traffic_per_kernel_avg.json | head -24
  echo "== PMC unit counters (separate passes)"
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    d=$O/pmcu_${TAG}_$i; rm -rf $d
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- $B --steps 16 --warmup 8 --no-cpu-baseline --no-extras --kernel-reps 1 > $d.log 2>&1 || { echo "unit pass $i failed"; tail -3 $d.log; }
    i=$((i+1))
  done
  python3 $R/scratch/pmc_agg.py $O/pmcu_${TAG}_[0-9] > $O/${TAG}_pmc_units_per_kernel_avg.json; grep -A16 "forward_move<float, 1, true, false, true" $O/${TAG}_pmc_units_per_kernel_avg.json | head -20
else
  echo "== other sizes"
  for cfg in "256 f32" "1024 f32" "2048 f32" "4096 f16" "8192 f32" "16384 f32" "16384 f16"; do
    set -- $cfg
    timeout -k 10 300 $B --size $1 --fields $2 --steps 100 --warmup 10 --no-cpu-baseline --no-extras --kernel-reps 1 > $O/${TAG}_size_$1_$2.json 2>/dev/null
    python3 -c "import json;d=json.load(open('$O/${TAG}_size_$1_$2.json'));print('$1 $2', d['value'], d['step_ms']['median'], d['config']['step_kind'][:28], d['roofline']['kernels_us'])"
  done
  echo "== replicas"
  timeout -k 10 200 $B --replicas 16 --size 1024 --steps 200 > $O/${TAG}_rep16_1024.json 2>/dev/null; cut -c1-160 $O/${TAG}_rep16_1024.json
  timeout -k 10 200 $B --replicas 16 --size 256 --steps 300 > $O/${TAG}_rep16_256.json 2>/dev/null; cut -c1-160 $O/${TAG}_rep16_256.json
  timeout -k 10 400 $B --replicas 2 --size 16384 --fields f16 --steps 30 --warmup 10 > $O/${TAG}_rep2_16384_f16.json 2>/dev/null; cut -c1-160 $O/${TAG}_rep2_16384_f16.json
  echo "== phase stamps (diagnostic build)"
  DIE_AMD_LIB=$R/scratch/libs/libdie_stamps.so timeout -k 10 300 python3 $R/scratch/pic_stamps.py > $O/${TAG}_phase_stamps.txt 2>&1; grep -v amdgpu $O/${TAG}_phase_stamps.txt
  echo "== decomposed: 2 and 4 ranks over gloo on the one GPU (refresh overlapped / not), one rank"
  DIE_DIST_BACKEND=gloo timeout -k 10 300 $B --gpus 2 --steps 24 --warmup 8 --no-cpu-baseline > $O/${TAG}_bench_g2.json 2> $O/${TAG}_bench_g2.err; line $O/${TAG}_bench_g2.json g2
  DIE_DIST_BACKEND=gloo timeout -k 10 300 $B --gpus 4 --steps 24 --warmup 8 --no-cpu-baseline > $O/${TAG}_bench_g4.json 2> $O/${TAG}_bench_g4.err; line $O/${TAG}_bench_g4.json g4
  DIE_DIST_BACKEND=gloo timeout -k 10 300 $B --gpus 4 --steps 24 --warmup 8 --no-cpu-baseline --no-refresh-overlap > $O/${TAG}_bench_g4_noov.json 2> $O/${TAG}_bench_g4_noov.err; line $O/${TAG}_bench_g4_noov.json g4-no-overlap
  timeout -k 10 300 $B --force-dist --steps 40 --warmup 8 --no-cpu-baseline --no-extras > $O/${TAG}_bench_dist1.json 2>/dev/null; line $O/${TAG}_bench_dist1.json dist1
  timeout -k 10 300 $B --force-dist --steps 40 --warmup 8 --no-cpu-baseline --no-extras --no-refresh-overlap > $O/${TAG}_bench_dist1_noov.json 2>/dev/null; line $O/${TAG}_bench_dist1_noov.json dist1-no-overlap
fi
