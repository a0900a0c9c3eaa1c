#!/bin/bash
# round 6: the order table (crowded tiles first inside every XCD band; scratch/libs/libdie_order.so, DIE_PIC_ORDER) at the bench's window AND late in a run, when the
# agents have aggregated (tiles of up to 3 500 agents at world step ~3 000: profiles/r05_final_fuzz_sweeps_and_longrun.txt)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export DIE_AMD_LIB=$R/scratch/libs/libdie_order.so
for pre in 192 3000; do for mode in 0 4 3 0; do
  d=$R/gpurun_out/ol_${pre}_$mode; rm -rf $d
  export DIE_PIC_ORDER=$mode
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --prewarm $pre --steps 60 --warmup 10 --no-cpu-baseline --no-extras --kernel-reps 1 > $d.json 2> $d.err || { echo "== $pre $mode FAILED"; tail -3 $d.err; continue; }
  f=$(find $d -name "*kernel_trace.csv" | head -1)
  python3 - $f $pre $mode <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
def avg(name, last):
    d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows if name in r['Kernel_Name'] and 'false, true, false, false' in r['Kernel_Name'] or (name in r['Kernel_Name'] and name == 'k_pic_resolve_diffuse')]
    d = d[-last:]
    return sum(d) / len(d) / 1e3, len(d)
k1, n1 = avg('k_pic_forward_move', 120); kb, n2 = avg('k_pic_resolve_diffuse', 120)
print(f'prewarm {sys.argv[2]:>5} DIE_PIC_ORDER={sys.argv[3]}: last {n1} launches: agent kernel {k1:6.1f} us, field kernel {kb:6.1f} us')
PY
done; done
