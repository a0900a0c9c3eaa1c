"""Host-side cost of one env.step(agent.forward(obs)) at 4096² (python + ctypes), by cProfile, and the wall clock of
sync=True vs sync=False loops.  usage: python3 scratch/host_profile.py [steps]"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch, die_amd
W = 4096
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for sync in (False, True):
    env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=1234, max_agents='alive', sync=sync)
    ag = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=1234, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
    obs = env._get_current_obs
    for _ in range(200):
        obs, *_ = env.step(ag.forward(obs))
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        obs, *_ = env.step(ag.forward(obs))
    t_issue = time.perf_counter() - t
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t
    print(f'sync={sync}: issue {1e6 * t_issue / n:.1f} us/step, with drain {1e6 * t_all / n:.1f} us/step')
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        obs, *_ = env.step(ag.forward(obs))
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14)
    print(s.getvalue()[:3500])
