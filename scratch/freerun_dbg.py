import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np, torch
import die_amd
from oracle import cpu_ref as R
from die_amd.device_array import from_q32, to_q32
W = H = 256
f32 = lambda v: np.asarray(v, dtype=np.float32).astype(np.float64)
medium, agents = R.synthetic_init(W, H, 0.15, seed=1234)
K = int(agents[2].sum()); agents = agents[:, :K].copy()
medium[1] = f32(medium[1]); agents[:2] = from_q32(to_q32(agents[:2])); agents[3] = f32(agents[3])
kw = dict(scale=1.53 / (W - 1), sense_offset=10.2 / (W - 1))
ref_env, ref_agent = R.RefEnv(medium, agents), R.RefPhysarumAgent(K, seed=3, **kw)
dir0 = f32(ref_agent._direction_rads); ref_agent._direction_rads = dir0.copy()
env = die_amd.Env.from_numpy(medium, agents)
ag = die_amd.PhysarumAgent(max_agents=K, seed=3, **kw); ag.set_state(dir0)
obs, robs = env._get_current_obs, ref_env.obs
shown = 0
for t in range(8):
    d_before = ref_agent._direction_rads.copy()
    dd_before = ag.direction_rads_numpy().copy() if t else dir0.copy()
    chem_ref = ref_env.medium[2].copy(); chem_dev = env.medium.to_numpy()[2]
    ax, ay = ref_env.agents[0].copy(), ref_env.agents[1].copy()
    act = ag.forward(obs); ract = ref_agent.forward(robs)
    hd_dev, hd_ref = ag.direction_rads_numpy(), ref_agent._direction_rads
    bad = np.nonzero(np.abs(np.angle(np.exp(1j * (hd_dev - hd_ref)))) > 1e-4)[0]
    print('step', t + 1, 'heading mismatches', len(bad), 'max |chem dev - ref| rel', np.abs(chem_dev - chem_ref).max() / max(chem_ref.max(), 1e-30))
    for n in bad[:5]:
        off = np.stack(R.polar2xy(kw['sense_offset'], d_before[n]))
        px, py = R.cell(ax[n] + off[0], W), R.cell(ay[n] + off[1], H)
        def grad(c):
            xm, xp = max(px - 1, 0), min(px + 1, W - 1); ym, yp = max(py - 1, 0), min(py + 1, H - 1)
            return (c[xp, py] - c[xm, py]) / (xp - xm), (c[px, yp] - c[px, ym]) / (yp - ym)
        gr, gd = grad(chem_ref), grad(chem_dev)
        drads_r = np.angle(gr[0] + 1j * gr[1]); 
        print('  slot', n, 'd_ref', repr(d_before[n]), 'd_dev', repr(dd_before[n]), 'probe', (px, py), 'grad ref', gr, 'grad dev', gd, 'drads ref', repr(drads_r),
              'delta ref', repr(R.renormalize_radians(d_before[n] - drads_r)), 'new hd dev/ref', hd_dev[n], hd_ref[n])
    obs, *_ = env.step(act); robs, *_ = ref_env.step(ract)
