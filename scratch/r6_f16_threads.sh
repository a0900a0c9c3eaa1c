#!/bin/bash
# round 6: fp16 planes, 32x128 tiles — workgroup size of the agent kernel (its windows are 27 KB there: five workgroups per CU fit by LDS)
R=$GRAFT_REPO_ROOT
for thr in 0 448 384 320 256 0; do
  timeout -k 10 300 python3 $R/bench.py --fields f16 --steps 60 --warmup 10 --no-cpu-baseline --no-extras --kernel-reps 4 --pic-threads $thr > /tmp/o.json 2> /tmp/o.err || { echo "FAILED $thr"; tail -3 /tmp/o.err; continue; }
  python3 -c "
import json
d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1])
print('f16 4096 threads $thr:', d['value'], 'steps/s, median step', d['step_ms']['median'], 'ms, kernels', d['roofline'].get('kernels_us'))
"
done
