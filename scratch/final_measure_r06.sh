#!/bin/bash
# The measurements DESIGN.md / profiles/ quote for round 6.  usage (GPU box): scratch/final_measure_r06.sh TAG a|b|c
#   a: the driver's command (bench line, rocprofv3 kernel stats), 200-step runs of the three step forms, PMC traffic and unit counters (fp32)
#   b: fp16 field channels — kernel stats + PMC traffic at 4096^2 and 16384^2 (2 replicas: one GPU's share of BASELINE configs[4])
#   c: other sizes, replicas, decomposed runs over gloo on the one GPU
TAG=$1; PART=$2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
line() { python3 -c "import json,sys;d=json.loads([l for l in open('$1') if l.startswith('{\"metric\"')][-1]);print('$2', d['value'], d['ms_per_step'], d.get('step_ms',{}).get('median'), d.get('roofline',{}).get('kernels_us'))"; }
stats() { # DIR TAGNAME ARGS...: rocprofv3 kernel stats of bench.py ARGS -> $O/TAGNAME_kernel_stats.csv
  d=$1; name=$2; shift 2; rm -rf $d
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- $B "$@" > $d.json 2> $d.err || { echo "stats $name failed"; tail -3 $d.err; return; }
  cp $(find $d -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv; head -5 $O/${name}_kernel_stats.csv | cut -c1-170
}
pmc() { # PREFIX OUTNAME SETS... -- ARGS...
  pre=$1; out=$2; shift 2; sets=(); while [ "$1" != "--" ]; do sets+=("$1"); shift; done; shift
  i=0
  for set in "${sets[@]}"; do
    d=$O/${pre}_$i; rm -rf $d
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- $B "$@" > $d.log 2>&1 || { echo "pmc pass $i ($set) failed"; tail -3 $d.log; }
    i=$((i+1))
  done
  python3 $R/scratch/pmc_agg.py $O/${pre}_[0-9] > $O/$out
}
if [ "$PART" = a ]; then
  echo "== driver command"; timeout -k 10 400 $B --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver.json 2> $O/${TAG}_bench_driver.err; cut -c1-200 $O/${TAG}_bench_driver.json
  stats $O/${TAG}_prof_driver ${TAG}_driver_cmd --gpus 1 --steps 20 --warmup 5 --no-extras
  echo "== 200 steps: two launches / three launches / classic / action stored"
  timeout -k 10 200 $B --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/${TAG}_bench_200.json 2>/dev/null; line $O/${TAG}_bench_200.json two
  timeout -k 10 200 $B --steps 200 --warmup 20 --no-cpu-baseline --no-extras --pic-three-launches > $O/${TAG}_bench_three.json 2>/dev/null; line $O/${TAG}_bench_three.json three
  timeout -k 10 200 $B --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-pic > $O/${TAG}_bench_classic.json 2>/dev/null; line $O/${TAG}_bench_classic.json classic
  timeout -k 10 200 $B --steps 200 --warmup 20 --no-cpu-baseline --no-extras --eager-actions > $O/${TAG}_bench_eager.json 2>/dev/null; line $O/${TAG}_bench_eager.json action-stored
  echo "== PMC traffic (two separate passes)"
  pmc pmc_${TAG} ${TAG}_pmc_traffic_per_kernel_avg.json "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" -- --steps 16 --warmup 8 --no-cpu-baseline --no-extras --kernel-reps 1
  grep -A7 "forward_move<float, 1, true, false, true\|resolve_diffuse" $O/${TAG}_pmc_traffic_per_kernel_avg.json | head -24
  echo "== PMC unit counters (separate passes)"
  pmc pmcu_${TAG} ${TAG}_pmc_units_per_kernel_avg.json "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" -- --steps 16 --warmup 8 --no-cpu-baseline --no-extras --kernel-reps 1
  grep -A16 "forward_move<float, 1, true, false, true" $O/${TAG}_pmc_units_per_kernel_avg.json | head -20
elif [ "$PART" = b ]; then
  echo "== fp16 field channels, 4096^2"
  timeout -k 10 400 $B --gpus 1 --steps 20 --warmup 5 --fields f16 > $O/${TAG}_bench_f16_4096.json 2> $O/${TAG}_bench_f16_4096.err; cut -c1-200 $O/${TAG}_bench_f16_4096.json
  stats $O/${TAG}_prof_f16_4096 ${TAG}_f16_4096 --steps 40 --warmup 10 --fields f16 --no-extras --no-cpu-baseline
  pmc pmc_${TAG}_f16 ${TAG}_f16_4096_pmc_traffic_per_kernel_avg.json "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" -- --steps 16 --warmup 8 --fields f16 --no-cpu-baseline --no-extras --kernel-reps 1
  grep -A7 "forward_move<__half, 1, true, false, true\|resolve_diffuse" $O/${TAG}_f16_4096_pmc_traffic_per_kernel_avg.json | head -24
  echo "== fp16 field channels, 16384^2, one world and 2 replicas (configs[4]: one GPU's share of 16 replicas over 8)"
  timeout -k 10 600 $B --size 16384 --fields f16 --steps 30 --warmup 10 --no-cpu-baseline --no-extras --kernel-reps 1 > $O/${TAG}_bench_f16_16384.json 2> $O/${TAG}_bench_f16_16384.err; line $O/${TAG}_bench_f16_16384.json 16384-f16
  timeout -k 10 600 $B --replicas 2 --size 16384 --fields f16 --steps 30 --warmup 10 > $O/${TAG}_rep2_16384_f16.json 2>/dev/null; cut -c1-200 $O/${TAG}_rep2_16384_f16.json
  stats $O/${TAG}_prof_f16_16384 ${TAG}_f16_16384 --size 16384 --fields f16 --steps 12 --warmup 4 --prewarm 17 --no-extras --no-cpu-baseline --kernel-reps 1
  pmc pmc_${TAG}_f16big ${TAG}_f16_16384_pmc_traffic_per_kernel_avg.json "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" -- --size 16384 --fields f16 --steps 8 --warmup 4 --prewarm 17 --no-cpu-baseline --no-extras --kernel-reps 1
  grep -A7 "forward_move<__half, 1, true, false, true\|resolve_diffuse" $O/${TAG}_f16_16384_pmc_traffic_per_kernel_avg.json | head -24
else
  echo "== other sizes"
  for cfg in "256 f32" "1024 f32" "2048 f32" "4096 f16" "8192 f32" "16384 f32" "16384 f16"; do
    set -- $cfg
    timeout -k 10 300 $B --size $1 --fields $2 --steps 100 --warmup 10 --no-cpu-baseline --no-extras --kernel-reps 1 > $O/${TAG}_size_$1_$2.json 2>/dev/null
    python3 -c "import json;d=json.load(open('$O/${TAG}_size_$1_$2.json'));print('$1 $2', d['value'], d['step_ms']['median'], d['config']['step_kind'][:28], d['roofline']['kernels_us'], d['roofline']['step'])"
  done
  echo "== replicas"
  timeout -k 10 200 $B --replicas 16 --size 1024 --steps 200 > $O/${TAG}_rep16_1024.json 2>/dev/null; cut -c1-160 $O/${TAG}_rep16_1024.json
  timeout -k 10 200 $B --replicas 16 --size 256 --steps 300 > $O/${TAG}_rep16_256.json 2>/dev/null; cut -c1-160 $O/${TAG}_rep16_256.json
  echo "== decomposed: 2 and 4 ranks over gloo on the one GPU (refresh overlapped / not), one rank"
  DIE_DIST_BACKEND=gloo timeout -k 10 300 $B --gpus 2 --steps 24 --warmup 8 --no-cpu-baseline > $O/${TAG}_bench_g2.json 2> $O/${TAG}_bench_g2.err; line $O/${TAG}_bench_g2.json g2
  DIE_DIST_BACKEND=gloo timeout -k 10 300 $B --gpus 4 --steps 24 --warmup 8 --no-cpu-baseline > $O/${TAG}_bench_g4.json 2> $O/${TAG}_bench_g4.err; line $O/${TAG}_bench_g4.json g4
  timeout -k 10 300 $B --force-dist --steps 40 --warmup 8 --no-cpu-baseline --no-extras > $O/${TAG}_bench_dist1.json 2>/dev/null; line $O/${TAG}_bench_dist1.json dist1
fi
