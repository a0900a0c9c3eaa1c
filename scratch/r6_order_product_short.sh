#!/bin/bash
# world steps 192 … 1000 of the 4096² bench world: no table / the product's table / a table that always sorts (scratch/libs/libdie_ord_always.so)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for pre in 192 400 700 1000; do for mode in 0 1 always 0 1 always; do
  d=$R/gpurun_out/ops_${pre}_$mode
  unset DIE_AMD_LIB; export DIE_PIC_ORDER=$mode
  if [ $mode = always ]; then export DIE_PIC_ORDER=1 DIE_AMD_LIB=$R/scratch/libs/libdie_ord_always.so; fi
  timeout -k 10 600 python3 $R/bench.py --prewarm $pre --steps 100 --warmup 10 --no-cpu-baseline --no-extras --kernel-reps 2 > $d.json 2> $d.err || { echo "== $pre $mode FAILED"; tail -3 $d.err; continue; }
  python3 -c "
import json
d=json.load(open('$d.json'))
print('prewarm $pre order=$mode:', d['value'], 'steps/s, median step', d['step_ms']['median'], 'ms, kernels', d['roofline'].get('kernels_us'))
"
done; done
