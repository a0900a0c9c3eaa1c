#!/bin/bash
# round 6: order-table experiment (DIE_PIC_ORDER, die_pic.hip) — kernel averages under rocprofv3; $@ = modes (0 = shipped mapping)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in "$@"; do
  d=$R/gpurun_out/r6_order_$mode; rm -rf $d
  export DIE_PIC_ORDER=$mode
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --kernel-reps 1 $AB_EXTRA > $d.json 2> $d.err || { echo "== $mode FAILED"; tail -5 $d.err; continue; }
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== DIE_PIC_ORDER=$mode W0=$DIE_PIC_ORDER_W0: $(python3 -c "import json;d=json.load(open('$d.json'));print(d['value'], d['step_ms']['median'])") $(grep 'order table' $d.err | head -1)"
  python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('k_pic_forward_move','k_pic_resolve')): print('    %-90s %6s calls %8.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
