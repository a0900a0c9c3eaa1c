#!/bin/bash
# K1 / K2 / sweep times for several tile shapes and K1 workgroup sizes (rocprofv3 kernel stats)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "6,6 512" "6,6 256" "5,6 256" "5,6 128" "5,7 512" "5,7 256"; do
  set -- $cfg
  d=$R/gpurun_out/sw_$1_$2; rm -rf $d
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --kernel-reps 1 --pic-tile $1 --pic-threads $2 > $d.json 2> $d.err || { tail -5 $d.err; exit 1; }
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  echo "== tile $1 threads $2: $(python3 -c "import json;d=json.load(open('$d.json'));print(d['value'], d['step_ms']['median'])")"
  grep -E "k_pic_forward_move|k_pic_resolve|k_diffuse_rows<float, 2, 2" $f | awk -F, '{print "   ", $1, $2, $4}'
done
