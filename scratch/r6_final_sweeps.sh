#!/bin/bash
# round 6, final build: randomised parity sweeps (new seeds), the decomposed fuzz of the ghost refresh, the 16 000-step soak
R=$GRAFT_REPO_ROOT; cd $R
echo "Randomised parity sweeps on the final build of round 6 (python -m tests.fuzz_cases <which> <cases> <seed>), seeds 6101-6107"
i=0
for spec in "binned 400" "step 200" "forward 200" "paths 150" "batched 40" "init 60" "nca 40"; do
  set -- $spec; i=$((i+1))
  timeout -k 10 900 python -m tests.fuzz_cases $1 $2 $((6100+i)) > /tmp/fz.log 2>&1; rc=$?
  echo "  $1: $2 cases, rc $rc, $(grep -ci 'fail\|mismatch' /tmp/fz.log) lines mentioning a failure; last line: $(tail -1 /tmp/fz.log | cut -c1-150)"
done
echo; echo "Decomposed world vs single device (scratch/fuzz_dist.py, FUZZ_PIC=1), capacity 8 N (refresh in place) seeds 700-739, capacity N + 64 (merge) seeds 800-815"
FUZZ_PIC=1 FUZZ_CAP_MULT=8 timeout -k 10 1500 python scratch/fuzz_dist.py 40 700 > /tmp/fd1.log 2>&1; echo "  rc $?: $(tail -2 /tmp/fd1.log | tr '\n' ' ' | cut -c1-300)"
FUZZ_PIC=1 timeout -k 10 900 python scratch/fuzz_dist.py 16 800 > /tmp/fd2.log 2>&1; echo "  rc $?: $(tail -2 /tmp/fd2.log | tr '\n' ' ' | cut -c1-300)"
echo; echo "16 000-step soak, binned vs classic step side by side (scratch/longrun.py)"
timeout -k 10 1500 python scratch/longrun.py 2>&1 | grep -v amdgpu | cut -c1-260
