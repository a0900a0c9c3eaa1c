"""Do a latency-bound kernel and a bandwidth-bound kernel of two INDEPENDENT worlds overlap on this GPU?  Two 4096² envs,
(a) stepped alternately on one stream, (b) each on its own stream.  If (b) is not clearly faster than (a), a banded
pipeline of one world's K1 / K2 / sweep over several streams cannot pay either."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import die_amd

def make(seed):
    W = 4096
    env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=seed, max_agents='alive', sync=False)
    agent = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=seed, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
    return env, agent

envs = [make(1), make(2)]
obs = [e._get_current_obs for e, _ in envs]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def loop(n, use_streams):
    for i in range(n):
        for k, (e, a) in enumerate(envs):
            if use_streams:
                with torch.cuda.stream(streams[k]):
                    obs[k], *_ = e.step(a.forward(obs[k]))
            else:
                obs[k], *_ = e.step(a.forward(obs[k]))

for mode in (False, True, False, True):
    loop(20, mode); torch.cuda.synchronize()
    t = time.perf_counter(); loop(100, mode); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print('two streams' if mode else 'one stream ', '%.1f us per pair of steps' % (dt / 100 * 1e6), flush=True)
