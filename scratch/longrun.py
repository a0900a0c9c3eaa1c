"""Long free run at 4096^2: throughput and invariants over time (agents aggregate into trails)."""
import sys, time; sys.path.insert(0, '.')
import torch, die_amd
W = H = 4096
env = die_amd.Env((W, H), die_amd.Dynamics(init_agent_ratio=0.15), seed=1234, max_agents='alive', sync=False, sort_every=8)
K = env.agents.N
agent = die_amd.PhysarumAgent(max_agents=K, seed=1234, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
obs = env._get_current_obs
done = 0
for target in (200, 1000, 2000, 4000, 8000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = target - done
    for _ in range(n):
        obs, res, *_ = env.step(agent.forward(obs))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    done = target
    occ = int(env.medium.occupied().sum().item())
    rew, alive = env.read_result(res)
    chem = env.medium.chem
    print(f'step {target}: {dt*1e6:.1f} us/step, occupied cells {occ} ({occ/K:.3f} of agents), reward {rew:.1f}, alive {alive}, '
          f'chem max {float(chem.max()):.3f} mean {float(chem.mean()):.5f}, food sum {float(env.medium.food.double().sum()):.1f}, '
          f'finite {bool(torch.isfinite(chem).all())}', flush=True)
