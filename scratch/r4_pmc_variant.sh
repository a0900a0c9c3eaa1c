#!/bin/bash
# usage (GPU box): scratch/r4_pmc_variant.sh NAME ... — kernel averages (rocprofv3 --stats) and FETCH_SIZE / TCC hit, miss of the two
# kernels of the binned step with variant libraries scratch/libs/libdie_NAME.so ("hip" = the tree's)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$R/scratch/libs/libdie_$v.so; [ $v = hip ] && lib=$R/die_amd/libdie_hip.so
  d=$R/gpurun_out/abp_$v; rm -rf $d ${d}_f ${d}_t
  DIE_AMD_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --kernel-reps 1 > $d.json 2> $d.err || { echo "== $v FAILED"; tail -5 $d.err; continue; }
  DIE_AMD_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d ${d}_f -- python3 $R/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-extras --kernel-reps 1 > ${d}_f.log 2>&1 || { echo "== $v pmc FAILED"; continue; }
  DIE_AMD_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d ${d}_t -- python3 $R/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-extras --kernel-reps 1 > ${d}_t.log 2>&1 || { echo "== $v pmc2 FAILED"; continue; }
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== $v: $(python3 -c "import json;d=json.load(open('$d.json'));print(d['value'], d['step_ms']['median'])")"
  python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('k_pic_forward_move','k_pic_resolve')): print('    %-70s %6s calls %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
  python3 $R/scratch/pmc_agg.py ${d}_f ${d}_t | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if isinstance(v,dict) and ('forward_move<float, 1, true, false' in k or 'resolve_diffuse' in k): print('    ', k[:58], {a:round(b) for a,b in v.items()})"
done
