#!/bin/bash
# usage (on the GPU box): scratch/r4_ab.sh [variant-lib-names...] — kernel averages (rocprofv3) of bench.py for the working tree's
# library, each named variant library (scratch/libs/libdie_NAME.so) and the round-3 baseline tree (scratch/base_r03), same box
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() {  # name, bench.py path, lib
  d=$R/gpurun_out/ab_$1; rm -rf $d
  DIE_AMD_LIB=$3 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $2 --steps 100 --warmup 20 --no-cpu-baseline --no-extras --kernel-reps 1 > $d.json 2> $d.err || { echo "== $1 FAILED"; tail -5 $d.err; return 1; }
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== $1: $(python3 -c "import json;d=json.load(open('$d.json'));print(d['value'], d['step_ms']['median'])")"
  python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('k_pic_forward_move','k_pic_resolve','k_turn_bits','k_pic_agents')): print('    %-70s %6s calls %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
}
run hip $R/bench.py $R/die_amd/libdie_hip.so
for v in "$@"; do run $v $R/bench.py $R/scratch/libs/libdie_$v.so; done
(cd $R/scratch/base_r03 && run base $R/scratch/base_r03/bench.py $R/scratch/base_r03/die_amd/libdie_hip.so)
