"""Host side of a synchronous step (Env(sync=True), tile-binned path, 4096^2): cProfile by function."""
import cProfile, pstats, sys, time, io
import torch
import die_amd
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=0, device='cuda:0', max_agents='alive', sync=True)
agent = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=0, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
obs = env._get_current_obs
for _ in range(100):
    obs, *_ = env.step(agent.forward(obs))
t0 = time.perf_counter()
for _ in range(n):
    obs, *_ = env.step(agent.forward(obs))
dt = time.perf_counter() - t0
print(f'{W}x{W} sync=True: {1e6 * dt / n:.1f} us/step ({n / dt:.0f} steps/s)')
pr = cProfile.Profile(); pr.enable()
for _ in range(n):
    obs, *_ = env.step(agent.forward(obs))
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
print('\n'.join(l[:160] for l in s.getvalue().splitlines()[:40]))
