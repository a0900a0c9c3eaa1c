#!/bin/bash
# gpurun_out/r06_final_* (scratch/final_measure_r06.sh) → profiles/ under the names profiles/README.md lists
cd $(dirname $0)/..
G=gpurun_out P=profiles T=r06_final
one() { grep -h '^{"metric"' $1 | tail -1 > $2; }
one $G/${T}_bench_driver.json $P/${T}_bench_driver_cmd.json
one $G/${T}_bench_200.json $P/${T}_bench_200_steps.json
one $G/${T}_bench_three.json $P/${T}_bench_three_launches.json
one $G/${T}_bench_classic.json $P/${T}_bench_classic.json
one $G/${T}_bench_eager.json $P/${T}_bench_action_stored_every_step.json
one $G/${T}_bench_g2.json $P/${T}_bench_2ranks_gloo_one_gpu.json
one $G/${T}_bench_g4.json $P/${T}_bench_4ranks_gloo_one_gpu.json
one $G/${T}_bench_dist1.json $P/${T}_bench_force_dist_1rank.json
one $G/${T}_rep16_1024.json $P/${T}_bench_replicas_16x1024.json
one $G/${T}_rep16_256.json $P/${T}_bench_replicas_16x256.json
one $G/${T}_rep2_16384_f16.json $P/${T}_bench_replicas_2x16384_f16.json
one $G/${T}_bench_f16_4096.json $P/${T}_bench_f16_4096_driver_cmd.json
one $G/${T}_bench_f16_16384.json $P/${T}_bench_f16_16384.json
cp $G/${T}_driver_cmd_kernel_stats.csv $G/${T}_f16_4096_kernel_stats.csv $G/${T}_f16_16384_kernel_stats.csv $P/
cp $G/${T}_pmc_traffic_per_kernel_avg.json $G/${T}_pmc_units_per_kernel_avg.json $G/${T}_f16_4096_pmc_traffic_per_kernel_avg.json $G/${T}_f16_16384_pmc_traffic_per_kernel_avg.json $P/
cp $G/${T}_pmc_traffic_per_kernel_avg.json $P/current_pmc_traffic_per_kernel_avg.json
python3 - <<'PY'
import json, glob
out = {f.split('size_')[1][:-5]: json.load(open(f)) for f in sorted(glob.glob('gpurun_out/r06_final_size_*.json'))}
json.dump(out, open('profiles/r06_final_other_sizes.json', 'w'), indent=1)
import bench
print('pmc sha', json.load(open('profiles/current_pmc_traffic_per_kernel_avg.json'))['kernel_source_sha'], 'sources', bench.kernel_source_sha())
PY
