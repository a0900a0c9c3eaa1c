#!/bin/bash
# round 6: class granularity / rebuild period of the order table (variant libraries), bench's window and world step 3 000
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for pre in 192 3000; do for v in "$@"; do
  lib=$R/scratch/libs/libdie_$v.so; [ $v = hip ] && lib=$R/die_amd/libdie_hip.so
  export DIE_AMD_LIB=$lib
  d=$R/gpurun_out/ov_${pre}_$v; rm -rf $d
  timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --prewarm $pre --steps 100 --warmup 10 --no-cpu-baseline --no-extras --kernel-reps 1 > $d.json 2> $d.err || { echo "== $pre $v FAILED"; tail -3 $d.err; continue; }
  f=$(find $d -name "*kernel_trace.csv" | head -1)
  python3 - $f $pre $v $d.json <<'PY'
import csv, sys, json
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
def avg(name, last, must=''):
    d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows if name in r['Kernel_Name'] and must in r['Kernel_Name']]
    d = d[-last:]
    return (sum(d) / len(d) / 1e3, len(d)) if d else (0.0, 0)
k1, n1 = avg('k_pic_forward_move', 200, 'false, true, false, false'); kb, n2 = avg('k_pic_resolve_diffuse', 200)
v = json.load(open(sys.argv[4]))
print(f'prewarm {sys.argv[2]:>5} {sys.argv[3]:>8}: {v["value"]:8.1f} steps/s (median step {v["step_ms"]["median"]} ms); agent kernel (action in registers) {k1:6.1f} us, field kernel {kb:6.1f} us')
PY
done; done
