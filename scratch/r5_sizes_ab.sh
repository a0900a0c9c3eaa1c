#!/bin/bash
# mid-size worlds: rocprofv3 kernel averages + steps/s of bench.py --size S with variant libraries: scratch/r5_sizes_ab.sh "2048 3072" hip nofood8 ...
R=$GRAFT_REPO_ROOT
sizes=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$R/scratch/libs/libdie_$v.so; [ $v = hip ] && lib=$R/die_amd/libdie_hip.so
  for size in $sizes; do
    d=$R/gpurun_out/sz_${v}_$size; rm -rf $d
    DIE_AMD_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --size $size --steps 200 --warmup 20 --no-cpu-baseline --no-extras --kernel-reps 1 > $d.json 2> $d.err || { echo "== $v $size FAILED"; tail -3 $d.err; continue; }
    f=$(find $d -name "*kernel_stats.csv" | head -1)
    echo "== $v $size: $(python3 -c "import json;d=json.load(open('$d.json'));print(d['value'], d['ms_per_step'])")"
    python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('k_pic_forward_move','k_pic_resolve_diffuse','k_forward_move_claim','k_diffuse_rows')) and int(r['Calls']) > 50: print('    %-70s %6s calls %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
  done
done
