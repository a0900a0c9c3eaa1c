#!/bin/bash
# round 6: the turnover micro-benchmark with WHOLE rounds of workgroups (nwg a multiple of the resident slots)
cd $GRAFT_REPO_ROOT/scratch/kbench_turnover
T=./turnover; T6=./turnover_w6
echo "== spin only, 512 threads"
for cfg in "4608 512 52000 10 0 768" "7680 512 52000 10 0 768" "4096 512 39000 10 0 1024" "8192 512 39000 10 0 1024" "4096 512 70000 10 0 512" "4096 512 52000 10 0 768" "5120 512 39000 10 0 1024"; do $T $cfg; done
echo "== spin + 24 KB in and out per workgroup (stores outstanding at the end)"
for cfg in "4608 512 52000 8 24 768" "4096 512 39000 8 24 1024" "4096 512 70000 8 24 512"; do $T $cfg; done
echo "== three per CU by REGISTERS (80 VGPRs), 30 KB LDS"
for cfg in "4608 512 30000 10 0 768" "4608 512 30000 8 24 768"; do $T6 $cfg; done
