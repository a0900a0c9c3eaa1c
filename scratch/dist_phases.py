"""Phase timing of DistEnv.step with N ranks sharing one GPU over gloo (transport is host-staged, so
the 'comm' numbers are pessimistic; the bookkeeping numbers are what a real node would also pay)."""
import os, sys, time; sys.path.insert(0, '.')
import torch, torch.distributed as dist
rank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
torch.cuda.set_device(0)
import die_amd
from die_amd import dist as D
W = 4096
grid = {1: (1, 1), 2: (1, 2), 4: (2, 2)}[world]
env = D.DistEnv((W * grid[0], W * grid[1]), grid, die_amd.Dynamics(init_agent_ratio=0.15), probe_reach=11, device='cuda:0', seed=1, overlap=False)
gW = W * grid[0]
agent = die_amd.PhysarumAgent(max_agents=env.capacity, seed=1 + rank, scale=1.53 / (gW - 1), sense_offset=10.2 / (gW - 1))
T = {}
def timed(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); T[name] = T.get(name, 0) + time.perf_counter() - t0; return r
orig_migrate, orig_halo = env._migrate, D.halo_exchange
env._migrate = lambda a: timed('migrate', lambda: orig_migrate(a))
D.halo_exchange = lambda p, g, c: timed('halo', lambda: orig_halo(p, g, c))
obs = env._get_current_obs
for i in range(30):
    if i == 10: T.clear(); torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
    obs, res = env.step(agent.forward(obs))
torch.cuda.synchronize(); dist.barrier(); dt = (time.perf_counter() - t0) / 20
if rank == 0:
    print(f'world {world}: {dt*1e6:.0f} us/step; per-step phases (us): ' + ', '.join(f'{k} {v/20*1e6:.0f}' for k, v in T.items()), f'agents {env.agents.N}', flush=True)
dist.destroy_process_group()
