"""Port of the reference's examples/minimal_run.py (lines 14-42) onto die_amd: same loop, same
agents and parameters; the interactive Qt plotter is replaced by an optional PNG dump of the three
frames `Env.render()` returns, so that the script runs headless on a GPU box.

    python examples/minimal_run.py [--iters 200] [--size 256] [--png out_dir]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from die_amd import BrownianAgent, Dynamics, Env, PhysarumAgent        # noqa: E402
from die_amd.agent.base import Agent                                    # noqa: E402


def run_minimal(agent: Agent, agent_ratio=0.1, field_size=(256, 256), iters=1000, png=None, seed=None):
    # Setup the environment
    dynamics = Dynamics(init_agent_ratio=agent_ratio)
    env = Env(field_size, dynamics, seed=seed)

    total_reward = 0
    obs = env._get_current_obs
    for i in range(iters):
        # Step: action & observation
        action = agent.forward(obs)
        obs, reward, _, _, stats = env.step(action)
        total_reward += reward
        if i % 50 == 0 or i == iters - 1:
            print(f'iter {i:4d}  total_reward={np.round(total_reward, 3)}  {stats}', flush=True)
    if png:
        import matplotlib
        matplotlib.use('Agg')
        from matplotlib import pyplot as plt
        os.makedirs(png, exist_ok=True)
        for name, img in zip(('medium', 'trace', 'agents'), env.render()):
            plt.imsave(os.path.join(png, f'{type(agent).__name__}_{name}.png'), np.clip(img, 0, 1))
    return total_reward, env


if __name__ == '__main__':
    p = argparse.ArgumentParser()
    p.add_argument('--iters', type=int, default=200)
    p.add_argument('--size', type=int, default=256)
    p.add_argument('--png', default=None)
    p.add_argument('--seed', type=int, default=None)
    a = p.parse_args()
    size = (a.size, a.size)

    random_agent = BrownianAgent(move_scale=0.01, seed=a.seed)
    run_minimal(random_agent, agent_ratio=0.05, field_size=size, iters=a.iters, png=a.png, seed=a.seed)

    physarum_agent = PhysarumAgent(max_agents=a.size * a.size,
                                   scale=0.006,
                                   turn_angle=30,
                                   sense_offset=0.04,
                                   seed=a.seed)
    run_minimal(physarum_agent, agent_ratio=0.15, field_size=size, iters=a.iters, png=a.png, seed=a.seed)
