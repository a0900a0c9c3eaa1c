"""Free run with the reference's default slot layout (max_agents=None: W·H slots, 85 % dead): the tile-binned step with the dead
slots behind the segments against the classic step + dead-slot pass, bit for bit at every checkpoint (every slot's coordinates,
agent_food, the fields, reward).  usage: python scratch/longrun_dead_slots.py [size=2048] [momentum]"""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch, die_amd
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
def make(pic):
    env = die_amd.Env((W, H), die_amd.Dynamics(init_agent_ratio=0.15), seed=77, max_agents=None, sync=False, pic=pic)
    agent = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=77, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
    return [env, agent, env._get_current_obs]
runs = [make(True), make(False)]
done = 0
for target in (50, 300, 1000, 2500):
    for r in runs:
        env, agent, obs = r[:3]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(target - done):
            obs, res, *_ = env.step(agent.forward(obs))
        torch.cuda.synchronize(); r[2] = obs; r[3:] = [(time.perf_counter() - t0) / (target - done), res]
    done = target
    (rew, alive), (rew2, alive2) = runs[0][0].read_result(runs[0][4]), runs[1][0].read_result(runs[1][4])
    a0, a1 = runs[0][0].agents.to_numpy(), runs[1][0].agents.to_numpy()
    same = all(torch.equal(getattr(runs[0][0].medium, f), getattr(runs[1][0].medium, f)) for f in ('chem', 'food')) and \
        np.array_equal(a0, a1) and (rew, alive) == (rew2, alive2) and \
        np.array_equal(runs[0][1].direction_rads_numpy(), runs[1][1].direction_rads_numpy())
    binned = runs[0][0]._pic is not None and runs[0][0]._pic.held is not None
    print(f'step {target}: binned({binned}) {runs[0][3]*1e6:.1f} / classic {runs[1][3]*1e6:.1f} us/step, identical state {same}, '
          f'slots {a0.shape[1]}, alive {alive}, reward {rew:.1f} / {rew2:.1f}', flush=True)
    assert same and binned
