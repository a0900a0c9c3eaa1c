#!/bin/bash
# gpurun_out/r03_final_* (scratch/final_measure_r03.sh) → profiles/ under the names profiles/README.md lists
cd $(dirname $0)/..
G=gpurun_out P=profiles
cp $G/r03_final_bench_200.json $P/r03_final_bench_200_steps.json
grep -h '^{"metric"' $G/r03_final_bench_g2.json > $P/r03_final_bench_2ranks_gloo_one_gpu.json
cp $G/r03_final_bench_classic.json $P/
cp $G/r03_final_bench_driver.json $P/r03_final_bench_driver_cmd.json
grep -h '^{"metric"' $G/r03_final_bench_dist1.json > $P/r03_final_bench_force_dist_1rank.json
cp $G/r03_final_rep16_1024.json $P/r03_final_bench_replicas_16x1024.json
cp $G/r03_final_rep16_256.json $P/r03_final_bench_replicas_16x256.json
cp $G/r03_final_rep2_16384_f16.json $P/r03_final_bench_replicas_2x16384_f16.json
cp $G/r03_final_bench_three.json $P/r03_final_bench_three_launches.json
cp $G/r03_final_driver_cmd_kernel_stats.csv $G/r03_final_phase_stamps.txt $G/r03_final_pmc_traffic_per_kernel_avg.json $P/
cp $G/r03_final_pmc_traffic_per_kernel_avg.json $P/current_pmc_traffic_per_kernel_avg.json
cp $G/r03_final_three_kernel_stats.csv $P/r03_final_three_launches_kernel_stats.csv
cp $G/r03_final_two_kernel_stats.csv $P/r03_final_two_launches_kernel_stats.csv
python3 - <<'PY'
import json, glob
out = {f.split('size_')[1][:-5]: json.load(open(f)) for f in sorted(glob.glob('gpurun_out/r03_final_size_*.json'))}
json.dump(out, open('profiles/r03_final_other_sizes.json', 'w'), indent=1)
import bench
print('pmc sha', json.load(open('profiles/current_pmc_traffic_per_kernel_avg.json'))['kernel_source_sha'], 'sources', bench.kernel_source_sha())
PY
