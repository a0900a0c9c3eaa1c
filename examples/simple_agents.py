"""The experiments of the reference's examples/simple_agents.py (lines 45-118) on die_amd: the four hand-written agents
(const / rand / grad / physarum, same parameters) in the two worlds ('st-perlin': static Perlin food; 'dyn-pred': food
flowing in running waves, `WaveSequence.get_flow_operator`), plus the custom update cycle of its `_manual_step` (lines
14-30) written with the library's stage entry points.  The Qt plotter is replaced by an optional PNG dump.

    python examples/simple_agents.py [--agent grad] [--dynamics st-perlin] [--size 156] [--iters 300] [--manual] [--png out]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch                                                                                     # noqa: E402
from die_amd import BrownianAgent, ConstAgent, Dynamics, Env, GradientAgent, PhysarumAgent, WaveSequence      # noqa: E402
from die_amd import _lib                                                                         # noqa: E402
from die_amd.device_array import _ptr, stream_ptr                                                # noqa: E402


def manual_step(env: Env, action):
    """`_manual_step` of the reference: the sub-steps of `Env.step` one call each, for debugging and for customised update
    cycles; like there it returns no reward.  The library's stages are a little coarser than the reference's methods:

        reference                                        here
        _agent_move                                      die_agent_move_claim   (move + who-stands-where claims + the agents'
        _agent_deposit_and_layout, _agent_feed           die_agent_resolve       own gain)  /  (deposit and the field's loss)
        _agent_lifecycle                                 die_agents_lifecycle   (only with Dynamics(agents_die=True))
        _medium_resource_dynamics                        env._food_flow()
        _medium_diffuse_decay                            env._medium_diffuse_decay()
    """
    env._agents_changed()                # (a tile-binned large world has to be binned again after a hand-made step)
    act = env._as_action(action)
    env.medium.next_epoch()
    m, a, u, d = env.medium.c_struct(), env.agents.c_struct(), act.c_struct(), env._c_dynamics()
    ws, wsn, sp = _ptr(env._workspace), env._workspace.numel(), stream_ptr(env.device)
    _lib.check(_lib.lib.die_agent_move_claim(C.byref(m), C.byref(a), C.byref(u), C.byref(d), ws, wsn, sp), 'die_agent_move_claim')
    _lib.check(_lib.lib.die_agent_resolve(C.byref(m), C.byref(a), C.byref(u), C.byref(d), ws, wsn, sp), 'die_agent_resolve')
    if env.dynamics.agents_die:
        _lib.check(_lib.lib.die_agents_lifecycle(C.byref(a), sp), 'die_agents_lifecycle')
    env._food_flow()                     # identity unless the dynamics carries a flow operator
    env._medium_diffuse_decay()
    return env._get_current_obs, 0, False, False, {}


def make_agent(agent_id: str, num_agents: int, seed=None):
    if agent_id == 'const':
        return ConstAgent(delta_xy=(-0.01, 0.005), deposit=0.1)
    if agent_id == 'rand':
        return BrownianAgent(move_scale=0.01, deposit_scale=0.1, seed=seed)
    if agent_id == 'grad':
        return GradientAgent(num_agents, sense_offset=0.03, inertia=0.95, scale=0.01, deposit=4.5, noise_scale=0.025,
                             normalized_grad=True, seed=seed)
    if agent_id == 'physarum':
        return PhysarumAgent(num_agents, turn_angle=35, sense_angle=120, sense_offset=0.03, turn_tolerance=0.05, inertia=0.,
                             scale=0.0075, deposit=4.5, noise_scale=0.0, normalized_grad=True, seed=seed)
    raise ValueError(agent_id)


def run_experiment(field_size=156, agent_id='rand', dynamics_id='st-perlin', iters=1000, agent_ratio=0.15, manual=False, png=None,
                   seed=None):
    max_agents = field_size * field_size
    size = (field_size, field_size)
    wave_flow = WaveSequence(size, dt=0.01).get_flow_operator(scale=0.5, decay=0.5)
    dynamics = {
        'st-perlin': Dynamics(init_agent_ratio=agent_ratio, food_infinite=False),
        'dyn-pred': Dynamics(init_agent_ratio=agent_ratio, food_infinite=False, op_food_flow=wave_flow),
    }[dynamics_id]
    env = Env(size, dynamics, seed=seed)
    agent = make_agent(agent_id, max_agents, seed)
    total_reward = 0.
    obs = env._get_current_obs
    for i in range(iters):
        action = agent.forward(obs)
        obs, reward, _, _, stats = manual_step(env, action) if manual else env.step(action)
        total_reward += reward
        if i % 100 == 0 or i == iters - 1:
            print(f'{agent_id:9s} {dynamics_id:9s} iter {i:4d}  total_reward={np.round(total_reward, 3)}  {stats}', flush=True)
    if png:
        import matplotlib
        matplotlib.use('Agg')
        from matplotlib import pyplot as plt
        os.makedirs(png, exist_ok=True)
        frames = list(env.render()) + list(agent.render())
        for k, img in enumerate(frames):
            plt.imsave(os.path.join(png, f'{agent_id}_{dynamics_id}_{k}.png'), np.clip(np.asarray(img, dtype=np.float64), 0, 1))
    torch.cuda.synchronize()
    return total_reward, env


if __name__ == '__main__':
    p = argparse.ArgumentParser()
    p.add_argument('--agent', default='grad', choices=['const', 'rand', 'grad', 'physarum', 'all'])
    p.add_argument('--dynamics', default='st-perlin', choices=['st-perlin', 'dyn-pred'])
    p.add_argument('--size', type=int, default=156)
    p.add_argument('--iters', type=int, default=300)
    p.add_argument('--ratio', type=float, default=0.1)
    p.add_argument('--manual', action='store_true', help='the custom update cycle instead of Env.step')
    p.add_argument('--png', default=None)
    p.add_argument('--seed', type=int, default=None)
    a = p.parse_args()
    for agent_id in (['const', 'rand', 'grad', 'physarum'] if a.agent == 'all' else [a.agent]):
        run_experiment(a.size, agent_id, a.dynamics, a.iters, a.ratio, a.manual, a.png, a.seed)
