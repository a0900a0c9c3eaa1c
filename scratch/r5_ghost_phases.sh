#!/bin/bash
# device time of the ghost refresh's phases, 2 x 2 and 1 x 2 ranks sharing the one GPU over gloo, refresh under the next step (overlap), in place / merged
cd $GRAFT_REPO_ROOT
export DIE_DIST_PROFILE=1 DIE_DIST_PROFILE_STAGGER=1 DIE_DIST_PROFILE_EVENTS=1 GHOST_OVERLAP=1
for n in 4 2; do for ip in 1 0; do
  echo "== $n ranks, DIE_REFRESH_IN_PLACE=$ip"
  DIE_REFRESH_IN_PLACE=$ip timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n * 10 + ip)) scratch/ghost_phases.py 8 2>&1 | grep "^world"
done; done
echo "== 4 ranks, refresh right behind its step (GHOST_OVERLAP=0: the merged layout, all phases visible)"
GHOST_OVERLAP=0 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29590 scratch/ghost_phases.py 8 2>&1 | grep "^world"
