"""Tile populations of the bench's world at its timed window (world step ~205): histogram, per-XCD-band totals."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np, torch
import die_amd
W = 4096; STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 205
env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=1234, max_agents='alive', sync=False)
ag = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=1234, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
obs = env._get_current_obs
for _ in range(STEPS):
    obs, *_ = env.step(ag.forward(obs))
torch.cuda.synchronize()
pic = env._pic
n = pic.meta[pic.cur][1].cpu().numpy().astype(np.int64).reshape(64, 64)
s = pic.meta[pic.cur][2].cpu().numpy().astype(np.int64).reshape(64, 64)
print('tiles', n.size, 'agents', n.sum(), 'mean', n.mean(), 'percentiles 1/10/50/90/99/99.9/max', [int(np.percentile(n, q)) for q in (1, 10, 50, 90, 99, 99.9, 100)])
print('histogram (bins of 128):', np.bincount((n.ravel() // 128)).tolist())
print('leavers per tile: mean', (n - s).mean(), 'max', (n - s).max())
bands = n.reshape(64, 8, 8).sum(axis=(0, 2))
print('agents per XCD band of 8 tile columns:', bands.tolist(), 'max/mean', bands.max() / bands.mean())
chunks = -(-n // 64)
print('64-agent chunks per tile: mean', chunks.mean(), '; 8-wave rounds per tile (ceil(chunks/8)):', np.bincount(-(-chunks.ravel() // 8)).tolist())
