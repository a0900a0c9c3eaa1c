#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for pre in 192; do for spec in "$@"; do
  v=${spec%%:*}; mode=${spec##*:}
  lib=$R/scratch/libs/libdie_$v.so; [ $v = hip ] && lib=$R/die_amd/libdie_hip.so
  export DIE_AMD_LIB=$lib DIE_PIC_ORDER=$mode
  d=$R/gpurun_out/ov3_${pre}_$v_$mode; rm -rf $d
  timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --prewarm $pre --steps 200 --warmup 10 --no-cpu-baseline --no-extras --kernel-reps 1 > $d.json 2> $d.err || { echo "== $pre $v FAILED"; tail -3 $d.err; continue; }
  f=$(find $d -name "*kernel_trace.csv" | head -1)
  python3 - $f $pre $spec $d.json <<'PY'
import csv, sys, json
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
def avg(name, last, must=''):
    d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows if name in r['Kernel_Name'] and must in r['Kernel_Name']]
    d = d[-last:]
    return (sum(d) / len(d) / 1e3, len(d)) if d else (0.0, 0)
k1, n1 = avg('k_pic_forward_move', 300, 'false, true, false, false'); kb, n2 = avg('k_pic_resolve_diffuse', 300)
v = json.load(open(sys.argv[4]))
print(f'prewarm {sys.argv[2]:>5} {sys.argv[3]:>14}: {v["value"]:8.1f} steps/s (median step {v["step_ms"]["median"]} ms); agent kernel {k1:6.1f} us, field kernel {kb:6.1f} us, sum {k1+kb:6.1f}')
PY
done; done
