"""The same 100 world steps (198..297 of bench.py's world) through three loops: the asynchronous Python step loop (bench.py's timed
loop), Env.run (die_pic_run: one library call), and the synchronous Gym loop (Env(sync=True): float reward + info every step)."""
import sys, time; sys.path.insert(0, '.')
import torch, die_amd
W = 4096; PRE = 197; N = 100
kw = dict(scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
for name in ('python-async', 'env.run', 'sync', 'python-async', 'env.run', 'sync'):
    env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=1234, max_agents='alive', device='cuda:0', sync=(name == 'sync'))
    agent = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=1234, **kw)
    obs = env._get_current_obs
    for _ in range(PRE):
        obs, *_ = env.step(agent.forward(obs))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if name == 'env.run':
        env.run(agent, N)
    else:
        for _ in range(N):
            obs, *_ = env.step(agent.forward(obs))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{name:14s} {N / dt:8.1f} steps/s  {dt / N * 1e6:7.2f} us/step', flush=True)
    del env, agent, obs; torch.cuda.empty_cache()
