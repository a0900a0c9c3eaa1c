#!/bin/bash
# rehearsal of the decomposed bench with several ranks sharing the one GPU (gloo, host-staged): mechanics and
# relative cost only.  usage: scratch/rehearse.sh NRANKS [bench args…]
N=$1; shift
export DIE_DIST_BACKEND=gloo
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $N "$@"
