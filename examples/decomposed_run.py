"""The Physarum loop of examples/minimal_run.py on a world decomposed over several GPUs (die_amd/dist.py).

One process per GPU:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 examples/decomposed_run.py \
        --tile 4096 --iters 200
Several ranks on ONE GPU (rehearsal, host-staged transport):  DIE_DIST_BACKEND=gloo … --nproc-per-node 2 …

Every rank owns a tile×tile tile of a (tile·Px)×(tile·Py) torus.  Ghost-agent mode: nothing crosses ranks for
`--refresh-every` steps; results equal the single-device run of the same world bit for bit.
"""
import argparse
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

GRIDS = {1: (1, 1), 2: (1, 2), 3: (1, 3), 4: (2, 2), 6: (2, 3), 8: (2, 4)}

if __name__ == '__main__':
    p = argparse.ArgumentParser()
    p.add_argument('--tile', type=int, default=1024)
    p.add_argument('--iters', type=int, default=200)
    p.add_argument('--refresh-every', type=int, default=8)
    p.add_argument('--seed', type=int, default=7)
    a = p.parse_args()
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0)) if torch.cuda.device_count() > 1 else 0
    torch.cuda.set_device(local)
    backend = os.environ.get('DIE_DIST_BACKEND', 'nccl')
    kw = dict(device_id=torch.device(f'cuda:{local}')) if backend == 'nccl' else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)

    import die_amd
    from die_amd.dist import DistEnv
    grid = GRIDS.get(world, (1, world))
    gW, gH = a.tile * grid[0], a.tile * grid[1]
    cells = max(gW, gH) - 1                       # offsets are fractions of the unit square: size them on the longer axis
    env = DistEnv((gW, gH), grid, die_amd.Dynamics(init_agent_ratio=0.15), probe_reach=11, device=f'cuda:{local}',
                  seed=a.seed, ghosts=True, migrate_every=a.refresh_every, max_step_cells=1.6)
    # one seed on every rank: random streams are keyed by the world slot id
    agent = die_amd.PhysarumAgent(max_agents=env.capacity, seed=a.seed, scale=1.53 / cells, sense_offset=10.2 / cells)
    obs = env._get_current_obs
    total = 0.0
    for i in range(a.iters):
        obs, result = env.step(agent.forward(obs))
        if i % 50 == 0 or i == a.iters - 1:
            reward, num_agents = env.read_result(result)          # scalar all-reduce: every rank calls it
            total += reward
            if rank == 0:
                print(f'iter {i:4d}  reward={reward:.3f}  num_agents={num_agents}  ({grid[0]}x{grid[1]} ranks, '
                      f'world {gW}x{gH}, halo ({env.geo.hx}, {env.geo.hy}))', flush=True)
    dist.barrier()
    dist.destroy_process_group()
