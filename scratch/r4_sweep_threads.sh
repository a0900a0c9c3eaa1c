#!/bin/bash
# usage (GPU box): scratch/r4_sweep_threads.sh LIBNAME T1 T2 ... — k_pic_forward_move at several workgroup sizes (bench.py --pic-threads)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
lib=$R/scratch/libs/libdie_$1.so; shift
for t in "$@"; do
  d=$R/gpurun_out/sw_$t; rm -rf $d
  DIE_AMD_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras --kernel-reps 1 --pic-threads $t > $d.json 2> $d.err || { echo "== $t FAILED"; tail -5 $d.err; continue; }
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== threads $t: $(python3 -c "import json;d=json.load(open('$d.json'));print(d['value'], d['step_ms']['median'])")"
  python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('k_pic_forward_move','k_pic_agents')): print('    %-70s %6s calls %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
