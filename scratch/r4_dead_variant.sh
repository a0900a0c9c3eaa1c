#!/bin/bash
# usage (GPU box): scratch/r4_dead_variant.sh NAME ... — the default slot layout's step per kernel (rocprofv3 --stats) with variant libraries
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$R/scratch/libs/libdie_$v.so; [ $v = hip ] && lib=$R/die_amd/libdie_hip.so
  for par in bench default; do
    d=$R/gpurun_out/dead_${v}_$par; rm -rf $d
    PYTHONPATH=$R DIE_AMD_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/scratch/default_slots.py 4096 60 $par > $d.log 2>&1 || { echo "== $v $par FAILED"; tail -3 $d.log; continue; }
    echo "== $v $par: $(grep steps/s $d.log)"
    f=$(find $d -name "*kernel_stats.csv" | head -1)
    python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('k_pic_forward_move','k_pic_resolve','k_pic_dead','k_pic_mark','fillBuffer')): print('    %-60s %6s calls %8.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
  done
done
