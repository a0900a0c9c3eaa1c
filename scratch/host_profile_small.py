"""Host side of one step of a small world (classic step): cProfile of the asynchronous loop — what the Python layer costs per
`env.step(agent.forward(obs))` when the kernels are shorter than the host's issue time.  usage: python scratch/host_profile_small.py [size] [steps]"""
import cProfile, pstats, sys, time, io
import torch
import die_amd

W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=0, device='cuda:0', max_agents='alive', sync=False)
agent = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=0, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
obs = env._get_current_obs
for _ in range(200):
    obs, *_ = env.step(agent.forward(obs))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    obs, *_ = env.step(agent.forward(obs))
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'{W}x{W}: issue {1e6 * (t1 - t0) / n:.1f} us/step, with drain {1e6 * (t2 - t0) / n:.1f} us/step')
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    obs, *_ = env.step(agent.forward(obs))
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14)
print('\n'.join(l[:150] for l in s.getvalue().splitlines()[:32]))
