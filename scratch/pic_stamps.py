"""Phase shares of k_pic_forward_move from a -DPIC_STAMPS build (scratch/libs/libdie_stamps.so): s_memtime at the phase
boundaries of wave 0 of every tile's workgroup.  Shares only — the stamped build is not the shipped kernel."""
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
raw = env._pic.error[2:2 + 32 * env._pic.NT].cpu().numpy().view(np.uint64).reshape(-1, 16)
st = raw[:, :6].astype(np.float64)
d = np.diff(st, axis=1)
names = ['tile loads issued, per-tile words, ranges', 'agent streams issued, tiles committed to LDS', 'filter arrivals + barrier', 'chunk loop (wave 0)', 'wave sum + final barrier']
tot = st[:, 5] - st[:, 0]
print('s_memtime ticks (shader clock) per tile workgroup, mean / median / p95')
for i, n in enumerate(names):
    print(f'  {n:46s} {d[:, i].mean():8.1f} {np.median(d[:, i]):8.1f} {np.percentile(d[:, i], 95):8.1f}   {100 * d[:, i].sum() / tot.sum():5.1f} %')
print(f'  {"total":46s} {tot.mean():8.1f} {np.median(tot):8.1f} {np.percentile(tot, 95):8.1f}')

tail = raw[:, 6].astype(np.float64)
if tail.any():
    t = tail[tail > 0]
    print(f'  tail of the previous tile (queue form): last barrier -> next tile started   {t.mean():8.1f} {np.median(t):8.1f} {np.percentile(t, 95):8.1f}   over {len(t)} tiles')
if raw[:, 8:15].any():
    st = raw[:, 8:15].astype(np.float64)
    d = np.diff(st, axis=1)
    names = ['loads issued, LUT, per-tile words + barrier', 'claims: rim decode, agent loads, LDS atomics', 'window commit + barrier', 'deposits, feeding + barrier',
             'x pass + barrier', 'y pass, stores issued']
    tot = st[:, 6] - st[:, 0]
    print('field kernel (k_pic_resolve_diffuse), wave 0 of every tile workgroup')
    for i, n in enumerate(names):
        print(f'  {n:46s} {d[:, i].mean():8.1f} {np.median(d[:, i]):8.1f} {np.percentile(d[:, i], 95):8.1f}   {100 * d[:, i].sum() / tot.sum():5.1f} %')
    print(f'  {"total":46s} {tot.mean():8.1f} {np.median(tot):8.1f} {np.percentile(tot, 95):8.1f}')
