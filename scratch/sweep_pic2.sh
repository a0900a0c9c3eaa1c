#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "hip 512" "hip 384" "nofood 512" "nofood 384" "nofood 448" "nofood 320"; do
  set -- $cfg
  lib=$R/die_amd/libdie_hip.so; [ $1 != hip ] && lib=$R/scratch/libs/libdie_$1.so
  d=$R/gpurun_out/sw2_$1_$2; rm -rf $d
  DIE_AMD_LIB=$lib timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --kernel-reps 1 --pic-threads $2 > $d.json 2> $d.err || { tail -5 $d.err; exit 1; }
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== $1 threads $2: $(python3 -c "import json;d=json.load(open('$d.json'));print(d['value'], d['step_ms']['median'])")"
  python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('k_pic_forward_move','k_pic_resolve','k_diffuse_rows<float, 2, 2')): print('    %-60s %8.1f us' % (r['Name'][:60], float(r['AverageNs'])/1e3))
PY
done
