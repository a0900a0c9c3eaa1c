#!/bin/bash
# usage (GPU box): scratch/r4_ab2.sh "NAME:ENV=VAL ENV2=VAL2" ... — kernel averages of bench.py (100 steps) per configuration of the working tree
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
  d=$R/gpurun_out/ab_$name; rm -rf $d
  ( export $envs; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --kernel-reps 1 $AB_EXTRA > $d.json 2> $d.err ) || { echo "== $name FAILED"; tail -8 $d.err; continue; }
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== $name [$envs]: $(python3 -c "import json;d=json.load(open('$d.json'));print(d['value'], d['step_ms']['median'])")"
  python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('k_pic_forward_move','k_pic_resolve','k_turn_bits','k_pic_agents')): print('    %-70s %6s calls %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
