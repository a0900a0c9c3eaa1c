#!/bin/bash
# usage (GPU box): scratch/r4_sizes_variant.sh NAME ... — bench.py's median step and per-kernel times at several sizes / field types
# with variant libraries scratch/libs/libdie_NAME.so ("hip" = the tree's)
R=$GRAFT_REPO_ROOT
for cfg in "4096 f32" "4096 f16" "8192 f32" "2048 f32"; do
  set -- $cfg; S=$1; F=$2
  for v in $VARIANTS; do
    lib=$R/scratch/libs/libdie_$v.so; [ $v = hip ] && lib=$R/die_amd/libdie_hip.so
    DIE_AMD_LIB=$lib timeout -k 10 200 python3 $R/bench.py --size $S --fields $F --steps 150 --warmup 20 --no-cpu-baseline --no-extras --kernel-reps 10 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$S $F %-10s' % '$v', d['value'], d['step_ms']['median'], d['roofline']['kernels_us'])"
  done
done
