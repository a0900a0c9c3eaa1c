#!/bin/bash
# round 6: the shipped order table (die_pic.order, rebuilt on the device every 8th step) against the band mapping (DIE_PIC_ORDER=0), at the bench's window and late in a run
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for pre in 192 1000 3000 8000; do for mode in 0 1 0 1; do
  d=$R/gpurun_out/op_${pre}_$mode; rm -rf $d
  export DIE_PIC_ORDER=$mode
  timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --prewarm $pre --steps 100 --warmup 10 --no-cpu-baseline --no-extras --kernel-reps 1 $AB_EXTRA > $d.json 2> $d.err || { echo "== $pre $mode FAILED"; tail -3 $d.err; continue; }
  f=$(find $d -name "*kernel_trace.csv" | head -1)
  python3 - $f $pre $mode $d.json <<'PY'
import csv, sys, json
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
def avg(name, last, must=''):
    d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows if name in r['Kernel_Name'] and must in r['Kernel_Name']]
    d = d[-last:]
    return (sum(d) / len(d) / 1e3, len(d)) if d else (0.0, 0)
k1, n1 = avg('k_pic_forward_move', 200); kb, n2 = avg('k_pic_resolve_diffuse', 200); ko, n3 = avg('k_pic_order', 1000)
v = json.load(open(sys.argv[4]))
print(f'prewarm {sys.argv[2]:>5} DIE_PIC_ORDER={sys.argv[3]}: {v["value"]:8.1f} steps/s (median step {v["step_ms"]["median"]} ms); last {n1} launches: agent kernel {k1:6.1f} us, field kernel {kb:6.1f} us; k_pic_order {ko:4.1f} us x {n3}')
PY
done; done
