"""Round 6, the probe: where do the reference's dead slots stand at the bench's side-measurement window — how many per 64x64 tile?
(Binning them into the tiles' segments gives every tile's workgroup its share of them.)"""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np, torch
import die_amd
W = 4096
env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=0, device='cuda:0', max_agents=None, sync=False)
ag = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=0, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
obs = env._get_current_obs
done = 0
for upto in (35, 200, 600, 2000):
    for _ in range(upto - done):
        obs, *_ = env.step(ag.forward(obs))
    done = upto
    torch.cuda.synchronize()
    A = env.agents
    dead = A.alive == 0
    x = (A.x[dead].to(torch.int64) & 0xFFFFFFFF).double() / 2.0 ** 32
    y = (A.y[dead].to(torch.int64) & 0xFFFFFFFF).double() / 2.0 ** 32
    cx = torch.clamp(torch.floor(x * (W - 1) + 0.5), 0, W - 1).long() >> 6
    cy = torch.clamp(torch.floor(y * (W - 1) + 0.5), 0, W - 1).long() >> 6
    n = torch.bincount(cx * 64 + cy, minlength=4096).cpu().numpy()
    nz = n[n > 0]
    print(f'step {upto}: {int(dead.sum())} dead slots on {len(nz)} of 4096 tiles; per occupied tile mean {nz.mean():.0f} median {np.median(nz):.0f} max {nz.max()}; '
          f'the 16 fullest tiles hold {np.sort(n)[-16:].sum() / n.sum() * 100:.0f} %  (alive agents per tile: mean 615)')
