#!/bin/bash
# quick A/B on the GPU box: the binned-step tests + the driver's bench command; $1 = tag
tag=${1:-x}
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "binned or food_stream or fuzz or configs2 or sync" > gpurun_out/r5_${tag}_t.log 2>&1; echo rc=$? >> gpurun_out/r5_${tag}_t.log; tail -3 gpurun_out/r5_${tag}_t.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_${tag}_b.json 2> gpurun_out/r5_${tag}_b.err; tail -c 300 gpurun_out/r5_${tag}_b.err
python -c "
import json
d=json.loads(open('gpurun_out/r5_${tag}_b.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernels_us'], d['config'].get('steps_per_s_with_the_action_stored_every_step'), d['config']['side_measurements'])
"
