"""Phase stamps of k_pic_agents from a -DPIC_STAMPS build (DIE_AMD_LIB=scratch/libs/libdie_stamps.so): s_memtime of one
iteration (tile) of every workgroup.  Shares only — the stamped build is not the shipped kernel."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np, torch
import die_amd
W = 4096
env = die_amd.Env((W, W), die_amd.Dynamics(init_agent_ratio=0.15), seed=1234, max_agents='alive', sync=False)
ag = die_amd.PhysarumAgent(max_agents=env.agents.N, seed=1234, scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
obs = env._get_current_obs
for _ in range(40):
    obs, *_ = env.step(ag.forward(obs))
torch.cuda.synchronize()
raw = env._pic.error[2:].cpu().numpy().view(np.uint64).reshape(-1, 16).astype(np.float64)
ok = raw[:, 0] > 0
t0 = raw[:, 0]
names = {1: 'wave 3: flush of the previous tile done', 2: 'wave 0: filter + gathers issued', 3: 'wave 0: its DMAs landed',
         4: 'wave 1: candidates + words issued', 5: 'wave 2: chem window issued', 6: 'wave 2: stayers issued', 7: 'wave 2: its DMAs landed',
         8: 'compute wave 0 done', 9: 'compute wave 11 done', 10: 'barrier passed'}
print('ticks after the iteration started (wave 0 past the barrier), mean / median / p95 over', int(ok.sum()), 'tiles')
for k, n in names.items():
    m = ok & (raw[:, k] > 0)
    d = raw[m, k] - t0[m]
    print(f'  {n:44s} {d.mean():9.1f} {np.median(d):9.1f} {np.percentile(d, 95):9.1f}')
