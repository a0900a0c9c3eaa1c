"""Per-agent agreement of device free runs with the float64 oracle from the reference's default start (agents on cell
centres, chem = 0, sense angle 90 on the 30-degree lattice): step by step, the fraction of agents on the same cell and
with the same heading; the first step at which any agent differs, and what decided it."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np, torch
import die_amd
from oracle import cpu_ref as R
from die_amd.device_array import from_q32, to_q32
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
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
first = None
for t in range(steps):
    act = ag.forward(obs); ract = ref_agent.forward(robs)
    und = ref_agent.last_undetermined
    obs, *_ = env.step(act); robs, *_ = ref_env.step(ract)
    a = env.agents.to_numpy()
    same = (R.cell(a[0], W) == R.cell(ref_env.agents[0], W)) & (R.cell(a[1], H) == R.cell(ref_env.agents[1], H))
    hd = np.abs(np.angle(np.exp(1j * (ag.direction_rads_numpy() - ref_agent._direction_rads)))) < 1e-4
    dep = np.isclose(act.to_numpy()[2], ract[2], rtol=1e-4, atol=1e-7)
    if first is None and not (same & hd & dep).all():
        first = t
    print(f'step {t + 1:3d}: same cell {same.mean():.5f}  same heading {hd.mean():.5f}  same deposit {dep.mean():.5f}  und(oracle) {und.mean():.3f}')
print('first step with any difference:', None if first is None else first + 1)
