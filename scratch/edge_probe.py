"""Edge cases through the public API: what happens, vs the oracle where it is defined."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch, die_amd
from oracle import cpu_ref as R
from die_amd.device_array import from_q32, to_q32

def run(name, fn):
    try:
        print(name, '->', fn(), flush=True)
    except Exception as e:
        print(name, '-> EXC', type(e).__name__, str(e)[:200], flush=True)

def all_dead():
    W, H, N = 32, 24, 100
    medium = np.zeros((3, W, H)); medium[1] = 0.5
    agents = np.zeros((4, N)); agents[:2] = np.random.RandomState(0).rand(2, N)       # alive = 0 everywhere
    env = die_amd.Env.from_numpy(medium, agents)
    ag = die_amd.PhysarumAgent(max_agents=N, seed=1)
    obs = env._get_current_obs
    out = [env.step(ag.forward(obs))[1:] for _ in range(3)]
    return out[-1]

def one_agent():
    W, H = 16, 16
    medium = np.zeros((3, W, H)); medium[1] = 0.5
    agents = np.array([[0.5], [0.5], [1.0], [0.2]])
    env = die_amd.Env.from_numpy(medium, agents)
    ag = die_amd.PhysarumAgent(max_agents=1, seed=1)
    obs = env._get_current_obs
    for _ in range(5): obs, rew, term, trunc, info = env.step(ag.forward(obs))
    return rew, term, info

def tiny_field():
    env = die_amd.Env((2, 2), die_amd.Dynamics(init_agent_ratio=0.5), seed=3)
    ag = die_amd.BrownianAgent(max_agents=env.agents.N, seed=1)
    obs = env._get_current_obs
    for _ in range(3): obs, rew, term, trunc, info = env.step(ag.forward(obs))
    return env.medium.shape, rew, info

def ratio_zero():
    env = die_amd.Env((32, 32), die_amd.Dynamics(init_agent_ratio=0.0), seed=3)
    ag = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=1)
    obs = env._get_current_obs
    obs, rew, term, trunc, info = env.step(ag.forward(obs))
    return env.agents.N, rew, term, info

def ratio_zero_alive_only():
    env = die_amd.Env((32, 32), die_amd.Dynamics(init_agent_ratio=0.0), seed=3, max_agents='alive')
    return env.agents.N

def odd_shape():
    env = die_amd.Env((37, 53), die_amd.Dynamics(init_agent_ratio=0.2), seed=3, max_agents='alive')
    ag = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=1)
    obs = env._get_current_obs
    for _ in range(4): obs, rew, term, trunc, info = env.step(ag.forward(obs))
    return rew, info

def everyone_on_one_cell():
    W, H, N = 64, 64, 5000
    medium = np.zeros((3, W, H)); medium[1] = 0.5
    agents = np.zeros((4, N)); agents[0] = 0.3; agents[1] = 0.7; agents[2] = 1; agents[3] = 0.1
    env = die_amd.Env.from_numpy(medium, agents)
    ref = R.RefEnv(medium.copy(), agents.copy())
    act = np.zeros((3, N)); act[2] = np.arange(N) * 1e-3
    _, rew, _, _, info = env.step(act)
    _, wrew, _, _, winfo = ref.step(act)
    m = env.medium.to_numpy()
    return rew, wrew, info['num_agents'], winfo['num_agents'], float(np.abs(m[2] - ref.medium[2]).max()), int(m[0].sum())

for f in (all_dead, one_agent, tiny_field, ratio_zero, ratio_zero_alive_only, odd_shape, everyone_on_one_cell):
    run(f.__name__, f)
