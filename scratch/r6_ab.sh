#!/bin/bash
# usage (GPU box): scratch/r6_ab.sh NAME ... — for every library (hip = the tree's, else scratch/libs/libdie_NAME.so): FIRST the tile-binned parity tests on that
# very build (a variant's timing counts only if they pass: VERDICT r5 item 2b), THEN rocprofv3 kernel averages of bench.py (100 steps).  AB_EXTRA: more bench arguments.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$R/scratch/libs/libdie_$v.so; [ $v = hip ] && lib=$R/die_amd/libdie_hip.so
  export DIE_AMD_LIB=$lib
  if [ -z "$AB_SKIP_PARITY" ]; then
    (cd $R && timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tile_binned_step_vs_oracle or tile_binned_step_equals_classic or crowd" > $R/gpurun_out/ab_$v.parity.log 2>&1)
    par=$(tail -1 $R/gpurun_out/ab_$v.parity.log)
    case "$par" in *failed*|*error*) echo "== $v: PARITY FAILED ($par) — not timed"; continue;; esac
  else par="parity not run"; fi
  d=$R/gpurun_out/ab_$v; rm -rf $d
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --kernel-reps 1 $AB_EXTRA > $d.json 2> $d.err || { echo "== $v FAILED"; tail -5 $d.err; continue; }
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== $v [$par]: $(python3 -c "import json;d=json.load(open('$d.json'));print(d['value'], 'steps/s, median step', d['step_ms']['median'], 'ms')")"
  python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('k_pic_forward_move','k_pic_resolve')): print('    %-90s %6s calls %8.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
