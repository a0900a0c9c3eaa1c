"""Steps/s of the synchronous Gym loop (Env(sync=True): float reward + info every step) at several sizes; run with
DIE_HOST_RESULT=0 for the 24-byte copy instead of the pinned result words.  usage: python3 scratch/sync_rate.py"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch, die_amd
for name, W, mk in (('Brownian 256^2 (BASELINE configs[0])', 256, lambda env: die_amd.BrownianAgent(seed=1)),
                    ('Physarum 256^2', 256, None), ('Physarum 1024^2', 1024, None), ('Physarum 2048^2', 2048, None), ('Physarum 4096^2', 4096, None)):
    env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=1234, max_agents='alive', sync=True)
    ag = mk(env) if mk else die_amd.PhysarumAgent(max_agents=env.agents.N, seed=1234, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
    obs, _ = env.reset(seed=1234) if False else (env._get_current_obs, None)
    n = 300 if W < 4096 else 200
    for _ in range(60):
        obs, rew, term, trunc, info = env.step(ag.forward(obs))
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        obs, rew, term, trunc, info = env.step(ag.forward(obs))
    dt = (time.perf_counter() - t) / n
    print(f'{name:40s} {1 / dt:9.0f} steps/s  ({dt * 1e6:.1f} us/step)', flush=True)
