"""Where does the tile-binned step start to win?  Same world stepped with the binned and the classic step, several sizes and
tile shapes.  usage: python3 scratch/pic_threshold.py"""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch, die_amd

def rate(W, H, pic, tile=None, steps=400):
    die_amd.Env.PIC_MIN_CELLS = 1 if pic else 1 << 60
    env = die_amd.Env((W, H), die_amd.Dynamics(init_agent_ratio=0.15), seed=1234, max_agents='alive', sync=False)
    if tile:
        env._pic_tile = tile
    ag = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=1234, scale=1.53 / (max(W, H) - 1), sense_offset=10.2 / (max(W, H) - 1))
    obs = env._get_current_obs
    for _ in range(100):
        obs, *_ = env.step(ag.forward(obs))
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        obs, *_ = env.step(ag.forward(obs))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    kind = 'binned' if env._pic is not None and env._pic.held is not None else 'classic'
    return dt * 1e6, kind

for W, H in ((1024, 1024), (1536, 1536), (2048, 1024), (2048, 2048), (3072, 2048), (3072, 3072), (4096, 2048)):
    out = [f'{W}x{H}:']
    for pic, tile in ((False, None), (True, None), (True, (5, 6)), (True, (4, 5))):
        try:
            us, kind = rate(W, H, pic, tile)
            out.append(f'{kind}{"" if tile is None else tile} {us:.1f}')
        except Exception as e:
            out.append(f'{tile} failed: {type(e).__name__}')
    print('  '.join(out), flush=True)
