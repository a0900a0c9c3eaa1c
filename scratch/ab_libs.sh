#!/bin/bash
# usage: scratch/ab_libs.sh NAME... — rocprofv3 kernel averages of bench.py with each variant library, same box
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$R/scratch/libs/libdie_$v.so; [ $v = hip ] && lib=$R/die_amd/libdie_hip.so
  d=$R/gpurun_out/ab_$v; rm -rf $d
  DIE_AMD_LIB=$lib timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --kernel-reps 1 $AB_EXTRA > $d.json 2> $d.err || { tail -5 $d.err; exit 1; }
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== $v [$(grep " $v:" $R/scratch/variants.log | tail -1)]: $(python3 -c "import json;d=json.load(open('$d.json'));print(d['value'], d['step_ms']['median'])")"
  python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('k_pic_forward_move','k_pic_resolve','k_diffuse_rows<float, 2, 2','k_forward_move_claim','k_diffuse_rows<float, 2, 1')): print('    %-60s %8.1f us' % (r['Name'][:60], float(r['AverageNs'])/1e3))
PY
done
