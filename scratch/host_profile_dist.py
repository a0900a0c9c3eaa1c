"""Host-side cost of one DistEnv.step(agent.forward(obs)) (one rank, 4096² world, ghost mode + tile-binned step): cProfile.
usage: python3 scratch/host_profile_dist.py [steps]"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
import torch, torch.distributed as dist, die_amd
from die_amd.dist import DistEnv
dist.init_process_group('gloo', rank=0, world_size=1)
W = 4096
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
env = DistEnv((W, W), (1, 1), die_amd.Dynamics(init_agent_ratio=0.15), probe_reach=11, device='cuda:0', seed=1234, sort_every=8,
              migrate_every=8, max_step_cells=1.6, ghosts=True, ghost_headroom=1.3)
ag = die_amd.PhysarumAgent(max_agents=env.capacity, seed=1234, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
obs = env._get_current_obs
for _ in range(200):
    obs, *_ = env.step(ag.forward(obs))
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(n):
    obs, *_ = env.step(ag.forward(obs))
t_issue = time.perf_counter() - t
torch.cuda.synchronize()
t_all = time.perf_counter() - t
print(f'issue {1e6 * t_issue / n:.1f} us/step, with drain {1e6 * t_all / n:.1f} us/step, binned steps {env.pic_steps}')
pr = cProfile.Profile(); pr.enable()
for _ in range(n):
    obs, *_ = env.step(ag.forward(obs))
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue()[:5000])
dist.destroy_process_group()
