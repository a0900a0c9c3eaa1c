#!/bin/bash
# the two --pmc passes bench.py's roofline.traffic is read from -> gpurun_out/TAG_pmc_traffic_per_kernel_avg.json
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  d=$O/pmc_${TAG}_$i; rm -rf $d
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python3 $R/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-extras --kernel-reps 1 > $d.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $d.log; exit 1; }
  i=$((i+1))
done
python3 $R/scratch/pmc_agg.py $O/pmc_${TAG}_* > $O/${TAG}_pmc_traffic_per_kernel_avg.json && grep -A6 "forward_move\|kernel_source_sha" $O/${TAG}_pmc_traffic_per_kernel_avg.json | head -20
