"""Kernel times vs step count: how much does spatial (dis)order of the agent array cost?"""
import sys; sys.path.insert(0, '.')
import torch, die_amd, bench
W = H = 4096
env = die_amd.Env((W, H), die_amd.Dynamics(init_agent_ratio=0.15), seed=1234, max_agents='alive', sync=False)
K = env.agents.N
kw = dict(scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
agent = die_amd.PhysarumAgent(max_agents=K, seed=1234, **kw)
obs = env._get_current_obs
done = 0
for target in (2, 20, 100, 300, 1000, 3000):
    while done < target:
        obs, *_ = env.step(agent.forward(obs)); done += 1
    torch.cuda.synchronize()
    kt = bench.time_kernels(env, agent, 5)
    done += 0
    print(target, {k: round(v, 1) for k, v in kt.items()}, 'sum', round(sum(kt.values()), 1), flush=True)
