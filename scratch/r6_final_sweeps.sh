#!/bin/bash
# usage: scratch/r6_final_sweeps.sh [single] [dist] [soak]  (default: all three)
# round 6, final build: randomised parity sweeps (new seeds), the decomposed fuzz of the ghost refresh, the 16 000-step soak
R=$GRAFT_REPO_ROOT; cd $R
single() {
echo "Randomised parity sweeps on the final build of round 6 (python -m tests.fuzz_cases <which> <cases> <seed>), seeds 6101-6107"
i=0
for spec in "binned 400" "step 200" "forward 200" "paths 150" "batched 40" "init 60" "nca 40"; do
  set -- $spec; i=$((i+1))
  timeout -k 10 900 python -m tests.fuzz_cases $1 $2 $((6100+i)) > /tmp/fz.log 2>&1; rc=$?
  echo "  $1: $2 cases, rc $rc, $(grep -ci 'fail\|mismatch' /tmp/fz.log) lines mentioning a failure; last line: $(tail -1 /tmp/fz.log | cut -c1-150)"
done
}
dist() {
echo; echo "Decomposed world vs single device (scratch/fuzz_dist.py, FUZZ_PIC=1: random rank grids sharing the GPU over gloo; gathered world bit-equal to die_amd.Env's)"
tot() { python3 - "$1" <<'PY'
import re, sys
t = open(sys.argv[1]).read()
ok = [l for l in t.splitlines() if l.startswith('ok')]
s = lambda k: sum(int(m) for l in ok for m in re.findall(k + r' (\d+)', l))
print(f"    {len(ok)} cases ok, {t.count('FAIL')} FAIL; binned steps {s('binned steps')}, refreshes by tiles {s('refreshes by tiles')}, in place {s('in place')}, packed early {s('packed early')}, overlapped {s('overlapped')}; {t.strip().splitlines()[-1]}")
PY
}
echo "  capacity 8 N (room for the refresh in place), seeds 700-739"
FUZZ_PIC=1 FUZZ_CAP_MULT=8 FUZZ_FIRST=700 FUZZ_CASES=40 timeout -k 10 1500 python scratch/fuzz_dist.py > /tmp/fd1.log 2>&1; tot /tmp/fd1.log
echo "  capacity N + 64 (merge into the other layout), the action read before the step with probability 0.3, seeds 800-815"
FUZZ_PIC=1 FUZZ_MATERIALISE=0.3 FUZZ_FIRST=800 FUZZ_CASES=16 timeout -k 10 900 python scratch/fuzz_dist.py > /tmp/fd2.log 2>&1; tot /tmp/fd2.log
echo "  FUZZ_BIG=1 (ranks of 512-1024 cells per side: the headline's 64x64 tiles), capacity 3 N, seeds 900-911"
FUZZ_PIC=1 FUZZ_BIG=1 FUZZ_CAP_MULT=3 FUZZ_FIRST=900 FUZZ_CASES=12 timeout -k 10 1200 python scratch/fuzz_dist.py > /tmp/fd3.log 2>&1; tot /tmp/fd3.log
}
soak() {
echo; echo "16 000-step soak, binned vs classic step side by side (scratch/longrun.py)"
timeout -k 10 1500 python scratch/longrun.py 2>&1 | grep -v amdgpu | cut -c1-260
}
parts="${@:-single dist soak}"
for part in $parts; do $part; done
