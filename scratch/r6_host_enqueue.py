"""Round 6: is the host the bottleneck of the asynchronous Python loop?  Enqueue time per step (loop returns before the GPU is done)
against the GPU's time per step, 4096^2 fp32, for the Python loop and for Env.run (die_pic_run)."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch, die_amd
W = 4096
env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=1234, max_agents='alive', sync=False)
ag = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=1234, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
obs = env._get_current_obs
for _ in range(200):
    obs, *_ = env.step(ag.forward(obs))
torch.cuda.synchronize()
for n in (20, 100, 400):
    t0 = time.perf_counter()
    for _ in range(n):
        obs, *_ = env.step(ag.forward(obs))
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'python loop, {n:4d} steps: enqueue {1e6 * (t1 - t0) / n:6.1f} us/step, until the GPU is done {1e6 * (t2 - t0) / n:6.1f} us/step = {n / (t2 - t0):7.0f} steps/s')
for n in (20, 100, 400):
    t0 = time.perf_counter(); env.run(ag, n); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f'Env.run,     {n:4d} steps: enqueue {1e6 * (t1 - t0) / n:6.1f} us/step, until the GPU is done {1e6 * (t2 - t0) / n:6.1f} us/step = {n / (t2 - t0):7.0f} steps/s')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(300):
    obs, *_ = env.step(ag.forward(obs))
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
