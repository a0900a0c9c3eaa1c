#!/bin/bash
# usage: scratch/pmc.sh TAG "COUNTER COUNTER ..." ["COUNTER ..."]...   — one rocprofv3 --pmc pass per argument
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
i=0
for set in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$i
  rm -rf $out
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-extras --kernel-reps 1 > $out.log 2>&1 || { echo "pass $i failed"; tail -5 $out.log; exit 1; }
  i=$((i+1))
done
python3 $GRAFT_REPO_ROOT/scratch/pmc_agg.py $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_* > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}.json
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}.json
