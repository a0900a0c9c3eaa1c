#!/bin/bash
# usage (GPU box): scratch/r5_ab.sh NAME[:threads] ... — kernel averages of bench.py under rocprofv3 with variant libraries scratch/libs/libdie_NAME.so
# ("hip" = the tree's); :threads = DIE_PIC_THREADS for that run
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  v=${spec%%:*}; thr=0; [ "$spec" != "$v" ] && thr=${spec##*:}
  lib=$R/scratch/libs/libdie_$v.so; [ $v = hip ] && lib=$R/die_amd/libdie_hip.so
  d=$R/gpurun_out/ab_$v; rm -rf $d
  export DIE_AMD_LIB=$lib DIE_PIC_THREADS=$thr
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --kernel-reps 1 $AB_EXTRA > $d.json 2> $d.err || { echo "== $spec FAILED"; tail -5 $d.err; continue; }
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== $spec: $(python3 -c "import json;d=json.load(open('$d.json'));print(d['value'], d['step_ms']['median'])")"
  python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('k_pic_forward_move','k_pic_resolve')): print('    %-70s %6s calls %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
