#!/bin/bash
# round 6: the direct-delivery agent kernel priced beside the shipped one (die_pic.hip dd_probe_run).  $@ = values of DIE_DD_PROBE (1 ahead / 2 behind)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
DIE_DD_PROBE=1 DIE_DD_PROBE_CHECK=1 timeout -k 10 200 python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --kernel-reps 1 > $R/gpurun_out/r6_dd_check.json 2> $R/gpurun_out/r6_dd_check.err || { echo CHECK FAILED; tail -5 $R/gpurun_out/r6_dd_check.err; exit 1; }
grep "dd probe" $R/gpurun_out/r6_dd_check.err | tail -4
for mode in "$@"; do
  d=$R/gpurun_out/r6_dd_$mode; rm -rf $d
  export DIE_DD_PROBE=$mode
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --kernel-reps 1 $AB_EXTRA > $d.json 2> $d.err || { echo "== $mode FAILED"; tail -5 $d.err; continue; }
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== DIE_DD_PROBE=$mode: $(python3 -c "import json;d=json.load(open('$d.json'));print(d['value'], d['step_ms']['median'])")"
  python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('k_pic_forward_move','k_pic_resolve','k_pic_dd')): print('    %-90s %6s calls %8.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
