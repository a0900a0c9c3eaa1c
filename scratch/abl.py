"""Time the step kernels for each variant library in scratch/libs (one subprocess each)."""
import glob, os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
    import torch, die_amd, bench
    W = H = 4096
    env = die_amd.Env((W, H), die_amd.Dynamics(init_agent_ratio=0.15), seed=1234, max_agents='alive', sync=False, sort_every=int(os.environ.get('ABL_SORT', '8')))
    K = env.agents.N
    agent = die_amd.PhysarumAgent(max_agents=K, seed=1234, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
    obs = env._get_current_obs
    import time
    for _ in range(40):
        obs, *_ = env.step(agent.forward(obs))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        obs, *_ = env.step(agent.forward(obs))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    kt = bench.time_kernels(env, agent, 8)
    print(json.dumps(dict(us_per_step=round(dt * 1e6, 1), **{k: round(v, 1) for k, v in kt.items()})))
else:
    for lib in sorted(glob.glob(os.path.join(ROOT, 'scratch', 'libs', 'lib_*.so'))):
        env = dict(os.environ, DIE_AMD_LIB=lib)
        r = subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True)
        print(os.path.basename(lib), r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-800:], flush=True)
