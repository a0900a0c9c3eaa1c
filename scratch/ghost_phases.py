"""Phase timing of the ghost-agent refresh with N ranks sharing one GPU over gloo (transport is host-staged, so the
exchange numbers are pessimistic; the bookkeeping numbers are what a real node would also pay).
usage: DIE_DIST_PROFILE=1 python -m torch.distributed.run --nproc-per-node N scratch/ghost_phases.py [M]
GHOST_OVERLAP=0: the refresh right behind its step (all phases visible); DIE_DIST_PROFILE_STAGGER=1 DIE_DIST_PROFILE_EVENTS=1: the ranks
take their phases one after the other and every phase is bracketed by HIP events behind a spin kernel — device time without host latency."""
import os, sys, time; sys.path.insert(0, '.')
os.environ['DIE_DIST_PROFILE'] = '1'
import torch, torch.distributed as dist
rank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
torch.cuda.set_device(0)
import die_amd
from die_amd import dist as D
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W = 4096
grid = {1: (1, 1), 2: (1, 2), 4: (2, 2)}[world]
gW, gH = W * grid[0], W * grid[1]
env = D.DistEnv((gW, gH), grid, die_amd.Dynamics(init_agent_ratio=0.15), probe_reach=11, device='cuda:0', seed=1,
                migrate_every=M, max_step_cells=1.6, ghosts=True, overlap=os.environ.get('GHOST_OVERLAP', '1') == '1')
agent = die_amd.PhysarumAgent(max_agents=env.capacity, seed=1, scale=1.53 / (max(gW, gH) - 1), sense_offset=10.2 / (max(gW, gH) - 1))
obs = env._get_current_obs
steps = 4 * M
for i in range(2 * M + steps):
    if i == 2 * M:
        env._prof.clear(); torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
    obs, res = env.step(agent.forward(obs))
torch.cuda.synchronize(); dist.barrier(); dt = (time.perf_counter() - t0) / steps
if rank == 0:
    n_ref = steps // M
    print(f'world {world} M {M} halo {(env.geo.hx, env.geo.hy)} agents {env.agents.N} (owned {int(env.owned_mask().sum())}): {dt*1e6:.0f} us/step; '
          f'per refresh (us): ' + ', '.join(f'{k} {v/n_ref*1e6:.0f}' for k, v in env._prof.items()), flush=True)
dist.destroy_process_group()
