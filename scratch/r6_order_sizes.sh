#!/bin/bash
# round 6: the order table against the band mapping at the other sizes (bench.py value + per-kernel event times)
R=$GRAFT_REPO_ROOT
for cfg in "2048 f32" "4096 f32" "4096 f16" "8192 f32" "16384 f16"; do set -- $cfg; for mode in 0 1 0 1; do
  DIE_PIC_ORDER=$mode timeout -k 10 400 python3 $R/bench.py --size $1 --fields $2 --steps 60 --warmup 10 --no-cpu-baseline --no-extras --kernel-reps 2 > /tmp/o.json 2> /tmp/o.err || { echo "FAILED $cfg $mode"; tail -3 /tmp/o.err; continue; }
  python3 -c "
import json
d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1])
print('$1 $2 DIE_PIC_ORDER=$mode:', d['value'], 'steps/s, median step', d['step_ms']['median'], 'ms, kernels', d['roofline'].get('kernels_us'), 'order_table', d['config'].get('order_table'))
"
done; done
