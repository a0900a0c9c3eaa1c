#!/bin/bash
# usage: scratch/small_sizes.sh LIB...  — step time of the small worlds with each library
R=$GRAFT_REPO_ROOT; cd /tmp
for v in "$@"; do
  lib=$R/scratch/libs/libdie_$v.so; [ $v = hip ] && lib=$R/die_amd/libdie_hip.so
  for sz in 256 512 1024 2048; do
    DIE_AMD_LIB=$lib timeout -k 10 120 python3 $R/bench.py --size $sz --steps 300 --warmup 30 --no-cpu-baseline --kernel-reps 5 > $R/gpurun_out/small_$v_$sz.json 2>/dev/null || exit 1
    python3 -c "import json;d=json.load(open('$R/gpurun_out/small_$v_$sz.json'));print('$v $sz', d['value'], d['step_ms']['median'], d['roofline']['kernels_us'])"
  done
done
