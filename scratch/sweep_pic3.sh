#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$R/scratch/libs/libdie_$v.so
  d=$R/gpurun_out/sw3_$v; rm -rf $d
  DIE_AMD_LIB=$lib timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --kernel-reps 1 > $d.json 2> $d.err || { tail -5 $d.err; }
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== $v"
  python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('k_pic_forward_move','k_pic_resolve','k_diffuse_rows<float, 2, 2')): print('    %-60s %8.1f us' % (r['Name'][:60], float(r['AverageNs'])/1e3))
PY
done
