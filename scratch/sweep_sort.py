import sys, time; sys.path.insert(0, '.')
import torch, die_amd, bench
W = H = 4096
for se in (0, 1, 2, 4, 8, 16):
    env = die_amd.Env((W, H), die_amd.Dynamics(init_agent_ratio=0.15), seed=1234, max_agents='alive', sync=False, sort_every=se)
    K = env.agents.N
    agent = die_amd.PhysarumAgent(max_agents=K, seed=1234, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
    obs = env._get_current_obs
    for _ in range(40):
        obs, *_ = env.step(agent.forward(obs))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        obs, *_ = env.step(agent.forward(obs))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    kt = bench.time_kernels(env, agent, 5)
    print(f'sort_every={se}: {dt*1e6:.1f} us/step ({1/dt:.0f} steps/s)', {k: round(v, 1) for k, v in kt.items()}, flush=True)
