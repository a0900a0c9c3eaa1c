#!/bin/bash
# usage: scratch/r3_ab.sh TAG — the two forms of the tile-binned step side by side on one box: bench line + rocprofv3 kernel stats each
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in two three; do
  extra=""; [ $v = three ] && extra="--pic-three-launches"
  timeout -k 10 200 python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --kernel-reps 20 $extra > $O/${TAG}_${v}_bench.json 2> $O/${TAG}_${v}_bench.err || { tail -20 $O/${TAG}_${v}_bench.err; exit 1; }
  python3 -c "import json;d=json.load(open('$O/${TAG}_${v}_bench.json'));print('$v', d['value'], d['ms_per_step'], d['step_ms']['median'], d['roofline']['kernels_us'], d['roofline']['step'])"
  d=$O/${TAG}_${v}_prof; rm -rf $d
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --kernel-reps 2 $extra > $d.json 2> $d.err || { tail -20 $d.err; exit 1; }
  cp $(find $d -name "*kernel_stats.csv" | head -1) $O/${TAG}_${v}_kernel_stats.csv; head -5 $O/${TAG}_${v}_kernel_stats.csv | cut -c1-150
done
