"""Default slot layout on the tile-binned step: the rate per window of steps as the dead slots' cloud spreads from (0, 0),
and with a re-sort of the agent arrays by cell (Env.sort_agents → re-bin) every K steps."""
import sys, time
import torch
import die_amd

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 0
params = sys.argv[3] if len(sys.argv) > 3 else 'default'       # 'default': PhysarumAgent's own (41-cell step at 4096); 'bench': bench.py's (1.53-cell step, 10.2-cell probe)
pic = (sys.argv[4] if len(sys.argv) > 4 else 'binned') == 'binned'
windows = int(sys.argv[5]) if len(sys.argv) > 5 else 10
akw = dict(scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1)) if params == 'bench' else {}
env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=0, device='cuda:0', max_agents=None, sync=False, pic=pic)
agent = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=0, **akw)
obs = env._get_current_obs
step = 0
for w in range(windows):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        if K and step % K == 0 and step:
            env._agents_changed(); env.sort_agents()
        obs, *_ = env.step(agent.forward(obs))
        step += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'steps {step - 30:4d}..{step:4d}: {30 / dt:8.1f} steps/s  ({dt / 30 * 1e6:7.1f} us/step)  resort every {K}, {params} parameters, {"binned" if env._pic is not None and env._pic.held is not None else "classic"} step', flush=True)
env.check()
