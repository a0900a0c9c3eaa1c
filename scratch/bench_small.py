"""Launch-bound grids: the step-by-step loop vs Env.run (hipGraph replay)."""
import sys, time; sys.path.insert(0, '.')
import torch, die_amd
for W in (256, 1024, 2048, 4096):
    out = []
    for mode in ('loop', 'graph'):
        env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=1234, max_agents='alive', sync=False, sort_every=8)
        K = env.agents.N
        agent = die_amd.PhysarumAgent(max_agents=K, seed=1234, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
        n = env._graph_period() * (8 if W <= 1024 else 2)
        obs = env._get_current_obs
        if mode == 'loop':
            for _ in range(40): obs, *_ = env.step(agent.forward(obs))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): obs, *_ = env.step(agent.forward(obs))
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        else:
            env.run(agent, env._graph_period() + 40, graph=True)          # captures the graph
            torch.cuda.synchronize(); t0 = time.perf_counter()
            env.run(agent, n, graph=True)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        out.append(f'{mode} {dt * 1e6:.1f} us/step ({1 / dt:.0f} steps/s)')
    print(f'{W}x{W} K={K}: ' + '; '.join(out), flush=True)
