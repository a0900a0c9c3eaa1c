"""Where the host spends a step on a launch-bound grid (cProfile)."""
import sys, cProfile, pstats; sys.path.insert(0, '.')
import torch, die_amd
W = 256
env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=1, max_agents='alive', sync=False, sort_every=8)
agent = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=1, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
obs = env._get_current_obs
for _ in range(50): obs, *_ = env.step(agent.forward(obs))
def loop(n):
    global obs
    for _ in range(n): obs, *_ = env.step(agent.forward(obs))
pr = cProfile.Profile(); pr.enable(); loop(2000); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)
