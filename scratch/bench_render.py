"""Render path at 4096²: device frames (die_render_frames + downloads) vs the reference-style host renderer."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch, die_amd
from oracle.render_ref import EnvRenderer
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = die_amd.Env((W, H), die_amd.Dynamics(init_agent_ratio=0.15), seed=1, max_agents='alive', sync=False)
agent = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=1, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
obs = env._get_current_obs
for _ in range(10):
    obs, *_ = env.step(agent.forward(obs))
def t(fn, reps):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
r = env.render(); R = env._renderer
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); a.record(); R._run(env.medium, R._rgb, R._rgba, None); b.record(); torch.cuda.synchronize()
k_all = a.elapsed_time(b) * 1e3
R.rgb8(env.medium); torch.cuda.synchronize(); a.record(); R._run(env.medium, None, None, R._rgb8, update_trace=False); b.record(); torch.cuda.synchronize()
k_8 = a.elapsed_time(b) * 1e3
host = EnvRenderer((W, H))
print(f'{W}x{H}: die_render_frames kernel {k_all:.0f} us (rgb + trace + rgba), {k_8:.0f} us (rgb8 only); '
      f'Env.render() incl. downloads {t(env.render, 5):.1f} ms; Env.render_rgb8() {t(env.render_rgb8, 5):.1f} ms; '
      f'host renderer on downloaded float64 arrays {t(lambda: host.render(env.medium.to_numpy(), env.agents.to_numpy()), 2):.0f} ms', flush=True)
