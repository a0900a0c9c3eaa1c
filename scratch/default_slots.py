"""The reference's default slot layout (max_agents=None: W·H slots, 85 % never lived) on the tile-binned step:
steps/s and — under `rocprofv3 --kernel-trace --stats` — the per-kernel split (k_pic_mark, k_pic_dead beside the two kernels)."""
import sys, time
import torch
import die_amd

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
akw = dict(scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1)) if (len(sys.argv) > 3 and sys.argv[3] == 'bench') else {}
env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=0, device='cuda:0', max_agents=None, sync=False)
agent = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=0, **akw)
obs = env._get_current_obs
for _ in range(30):
    obs, *_ = env.step(agent.forward(obs))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    obs, *_ = env.step(agent.forward(obs))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
env.check()
print(f'{W}x{W}, {env.agents.N} slots, {int(env.agents.alive.sum())} alive: {n / dt:.1f} steps/s, {dt / n * 1e6:.1f} us/step, binned={env._pic is not None and env._pic.held is not None}')
