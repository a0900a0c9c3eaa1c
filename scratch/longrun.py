"""Long free run at 4096^2 (agents aggregate into trails): throughput, invariants and tile populations over time, the
tile-binned step against the classic step bit for bit at every checkpoint."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch, die_amd
W = H = 4096
def make(pic):
    env = die_amd.Env((W, H), die_amd.Dynamics(init_agent_ratio=0.15), seed=1234, max_agents='alive', sync=False, pic=pic)
    agent = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=1234, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
    return env, agent, env._get_current_obs
runs = [list(make(True)), list(make(False))]
K = runs[0][0].agents.N
done = 0
for target in (200, 1000, 3000, 8000, 16000):
    n = target - done
    for r in runs:
        env, agent, obs = r
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            obs, res, *_ = env.step(agent.forward(obs))
        torch.cuda.synchronize(); r[2] = obs; r.append((time.perf_counter() - t0) / n); r.append(res)
    done = target
    env = runs[0][0]
    rew, alive = env.read_result(runs[0][-1])              # (raises on a bookkeeping error of the binned step)
    rew2, alive2 = runs[1][0].read_result(runs[1][-1])
    per_tile = env._pic.meta[env._pic.cur][1]
    occ = int(env.medium.occupied().sum().item())
    same = all(torch.equal(getattr(runs[0][0].medium, f), getattr(runs[1][0].medium, f)) for f in ('chem', 'food')) and \
        np.array_equal(runs[0][0].agents.to_numpy(), runs[1][0].agents.to_numpy()) and (rew, alive) == (rew2, alive2)
    chem = env.medium.chem
    print(f'step {target}: binned {runs[0][-2]*1e6:.1f} / classic {runs[1][-2]*1e6:.1f} us/step, identical state {same}, occupied cells {occ} '
          f'({occ/K:.3f} of agents), agents per 64x64 tile max {int(per_tile.max())} mean {float(per_tile.float().mean()):.0f}, '
          f'reward {rew:.1f}, alive {alive}, chem max {float(chem.max()):.3f}, finite {bool(torch.isfinite(chem).all())}', flush=True)
    for r in runs:
        del r[3:]
    assert same
