#!/bin/bash
# usage: scratch/prof_bench.sh TAG [bench args...] — timeout -k 10 150 rocprofv3 --kernel-trace --stats of bench.py, summary to gpurun_out/TAG_*
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${TAG}_prof
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --kernel-reps 2 "$@" > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err || { tail -20 $R/gpurun_out/${TAG}_bench.err; exit 1; }
cat $R/gpurun_out/${TAG}_bench.json
f=$(find $R/gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/${TAG}_kernel_stats.csv
head -8 $f
