"""How much would the dead slots' pass gain from spatial order?  Default slot layout on the binned step; after `pre` steps (the
cloud has spread) time single steps, then sort the agent arrays by cell (Env.sort_agents → re-bin: the dead section keeps that
order) and time the following steps one by one."""
import sys, time
import torch
import die_amd

W = 4096
params = sys.argv[1] if len(sys.argv) > 1 else 'bench'
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 300
akw = dict(scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1)) if params == 'bench' else {}
env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=0, device='cuda:0', max_agents=None, sync=False)
agent = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=0, **akw)
obs = env._get_current_obs
for _ in range(pre):
    obs, *_ = env.step(agent.forward(obs))
def one():
    global obs
    torch.cuda.synchronize(); t0 = time.perf_counter()
    obs, *_ = env.step(agent.forward(obs))
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e6
print(params, 'before the sort:', ' '.join(f'{one():.0f}' for _ in range(5)), 'us/step')
torch.cuda.synchronize(); t0 = time.perf_counter()
env._agents_changed(); env.sort_agents()
torch.cuda.synchronize(); print(f'sort_agents: {(time.perf_counter() - t0) * 1e6:.0f} us')
print(params, 'after the sort (first step re-bins):', ' '.join(f'{one():.0f}' for _ in range(24)), 'us/step')
