#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp
for sz in 256 512 1024; do for se in 8 0; do
  timeout -k 10 120 python3 $R/bench.py --size $sz --steps 400 --warmup 40 --no-cpu-baseline --kernel-reps 3 --sort-every $se --no-pic > $R/gpurun_out/ss.json 2>/dev/null || exit 1
  python3 -c "import json;d=json.load(open('$R/gpurun_out/ss.json'));print('size $sz sort_every $se', d['value'], d['step_ms']['median'], d['roofline']['kernels_us'])"
done; done
for se in 8 0; do timeout -k 10 120 python3 $R/bench.py --size 2048 --steps 300 --warmup 40 --no-cpu-baseline --kernel-reps 3 --sort-every $se --no-pic > $R/gpurun_out/ss.json 2>/dev/null; python3 -c "import json;d=json.load(open('$R/gpurun_out/ss.json'));print('size 2048 classic sort_every $se', d['value'], d['step_ms']['median'], d['roofline']['kernels_us'])"; done
