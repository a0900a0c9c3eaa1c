#!/bin/bash
# round 6: fp16 field channels, tile shapes 64x64 / 32x128 at 4096^2 and 16384^2 — bench.py value, median step, per-kernel us
R=$GRAFT_REPO_ROOT
for size in 4096 16384; do for tile in "" "5,7"; do
  a="--fields f16 --size $size --steps 30 --warmup 8 --no-cpu-baseline --no-extras --kernel-reps 4"; [ -n "$tile" ] && a="$a --pic-tile $tile"
  timeout -k 10 400 python3 $R/bench.py $a > /tmp/o.json 2> /tmp/o.err || { echo "FAILED $size $tile"; tail -3 /tmp/o.err; continue; }
  python3 -c "
import json
d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1])
print('f16 $size tile ${tile:-6,6}:', d['value'], 'steps/s, median step', d['step_ms']['median'], 'ms, kernels', d['roofline'].get('kernels_us'))
"
done; done
