"""What it costs to take a step as four launches over subsets of the tiles (die_pic.sub_*: agent kernel and field kernel on the
tiles that need nothing from a neighbour, then both on the rest — the step a decomposed rank takes behind a ghost refresh)
instead of two launches over all tiles: a 4352² plane (a 4096² tile + 2 x 128 cells of halo = 68 x 68 tiles of 64 x 64), one GPU,
no decomposition — the launches are what is timed.  Also checks that both give the same bits."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np, torch
import die_amd
W = 4352
def make():
    env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=1234, max_agents='alive', sync=False)
    ag = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=1234, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
    return env, ag
def run(split, steps=200, warm=40):
    env, ag = make()
    obs = env._get_current_obs
    nt = W // 64
    ia, if_ = (3, 3, nt - 6, nt - 6), (4, 4, nt - 8, nt - 8)
    plan = [(1, (1,) + ia), (2, (1,) + if_), (1, (2,) + ia), (2, (2,) + if_)]
    def step():
        nonlocal obs
        action = ag.forward(obs)
        env._pic_plan = plan if split else None
        obs, *_ = env.step(action)
    for _ in range(warm): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    env.check()
    return dt, env.medium.chem.clone(), env.agents.to_numpy()
a = run(False); b = run(True)
print(f'two launches {a[0] * 1e6:.1f} us/step, four launches over subsets {b[0] * 1e6:.1f} us/step (+{(b[0] - a[0]) * 1e6:.1f} us); same bits: '
      f'{bool(torch.equal(a[1], b[1]) and np.array_equal(a[2], b[2]))}')
