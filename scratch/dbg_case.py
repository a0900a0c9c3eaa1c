import sys; sys.path.insert(0, '.')
import numpy as np, torch, die_amd
from oracle import cpu_ref as R
from tests.test_gpu_parity import random_state, f32
# reproduce fuzz_forward seed 2 case 96
import os
rs = np.random.RandomState(int(os.environ.get('FUZZ_SEED', '2')))
for case in range(int(os.environ.get('FUZZ_CASE', '96')) + 1):
    W = int(rs.choice([2, 5, 16, 33, 64, 200])); H = int(rs.choice([2, 7, 12, 64, 130, 256]))
    N = int(rs.choice([1, 50, 3000, 20000]))
    medium, agents = random_state(W, H, N, int(0.7 * N), rs)
    phys = rs.rand() < 0.6
    kw = dict(scale=float(rs.choice([0.001, 0.01, 0.05])), deposit=float(rs.choice([1.0, 4.0, 4.5])),
              sense_offset=float(rs.choice([0.0, 0.01, 0.04, 0.3])), normalized_grad=bool(rs.rand() < 0.8),
              grad_clip=None if rs.rand() < 0.2 else float(rs.choice([1e-5, 1e-3])))
    if phys:
        kw.update(turn_angle=int(rs.choice([20, 30, 35, 45])), sense_angle=int(rs.choice([60, 100, 120])), turn_tolerance=float(rs.choice([0.05, 0.1, 0.2])),
                  inertia=float(rs.choice([0.0, 0.0, 0.5])), noise_scale=float(rs.choice([0.0, 0.0, 0.02])))
    else:
        kw.update(inertia=float(rs.choice([0.0, 0.9])), noise_scale=float(rs.choice([0.0, 0.025])))
    prev = f32(rs.normal(0, .4, (2, N)))
print(case, W, H, N, phys, kw)
ref = R.RefPhysarumAgent(N, seed=case, **kw); dev = die_amd.PhysarumAgent(max_agents=N, seed=case, **kw)
ref._prev_grad = prev.copy()
dir0 = f32(ref._direction_rads); ref._direction_rads = dir0.copy()
want = ref.forward((agents, medium))
env = die_amd.Env.from_numpy(medium, agents)
dev.set_state(dir0, None)
got = dev.forward(env._get_current_obs).to_numpy()
bad = ~(np.isclose(got[0], want[0], rtol=1e-5, atol=1e-6*kw['scale']+1e-9) & np.isclose(got[1], want[1], rtol=1e-5, atol=1e-6*kw['scale']+1e-9) & np.isclose(got[2], want[2], rtol=1e-5, atol=1e-8))
print(bad.sum())
g = np.stack(np.gradient(medium[2]))
off = np.stack(R.polar2xy(kw['sense_offset'], dir0))
px, py = R.cell(agents[0] + off[0], W), R.cell(agents[1] + off[1], H)
gg = g[:, px, py]; norm = np.hypot(gg[0], gg[1])
for n in np.nonzero(bad)[0][:8]:
    print(n, 'got', got[:, n], 'want', want[:, n], 'norm', norm[n], 'g', gg[:, n], 'dir0', np.degrees(dir0[n]), 'wantdir', np.degrees(ref._direction_rads[n]), 'gotdir', np.degrees(dev.direction_rads_numpy()[n]), 'und', ref.last_undetermined[n])
