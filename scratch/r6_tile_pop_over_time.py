"""Tile populations of the bench's world over a run: share of tiles per 8-wave-round class (what k_pic_order sorts by)."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np, torch
import die_amd
W = 4096
env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=1234, max_agents='alive', sync=False)
ag = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=1234, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
obs = env._get_current_obs
done = 0
for upto in (50, 200, 400, 700, 1000, 2000, 3000, 5000, 8000, 12000, 16000):
    for _ in range(upto - done):
        obs, *_ = env.step(ag.forward(obs))
    done = upto
    torch.cuda.synchronize()
    pic = env._pic
    n = pic.meta[pic.cur][1].cpu().numpy().astype(np.int64)
    r = -(-(-(-n // 64)) // 8)
    rim = pic.rim_cnt.cpu().numpy().astype(np.int64)
    print(f'step {upto:6d}: max {n.max():5d}; tiles by rounds 1..7+: {np.bincount(np.minimum(r, 7), minlength=8)[1:].tolist()}; rounds >= 3: {(r >= 3).mean() * 100:4.1f} %, >= 4: {(r >= 4).mean() * 100:4.1f} %, '
          f'rim lists over their 224 entries: {(rim > 224).sum()}')
