"""Randomised forward() parity sweep (GPU vs oracle): Physarum and Gradient agents, random parameters."""
import sys, os; sys.path.insert(0, '.')
import numpy as np, torch
import die_amd
from oracle import cpu_ref as R
from tests.test_gpu_parity import random_state, f32
rs = np.random.RandomState(int(os.environ.get('FUZZ_SEED', '0')))
n_cases = int(os.environ.get('FUZZ_CASES', '200'))
fails = 0; worst = 0.0
for case in range(n_cases):
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
        ref = R.RefPhysarumAgent(N, seed=case, **kw); dev = die_amd.PhysarumAgent(max_agents=N, seed=case, **kw)
    else:
        kw.update(inertia=float(rs.choice([0.0, 0.9])), noise_scale=float(rs.choice([0.0, 0.025])))
        ref = R.RefGradientAgent(N, seed=case, **kw); dev = die_amd.GradientAgent(max_agents=N, seed=case, **kw)
    prev = f32(rs.normal(0, .4, (2, N)))
    ref._prev_grad = prev.copy()
    dir0 = f32(ref._direction_rads); ref._direction_rads = dir0.copy()
    want = ref.forward((agents, medium))
    env = die_amd.Env.from_numpy(medium, agents)
    dev.set_state(dir0, prev if kw['inertia'] else None)
    got = dev.forward(env._get_current_obs).to_numpy()
    atol = 1e-6 * kw['scale'] + 1e-9
    bad = ~(np.isclose(got[0], want[0], rtol=1e-5, atol=atol) & np.isclose(got[1], want[1], rtol=1e-5, atol=atol) & np.isclose(got[2], want[2], rtol=1e-5, atol=1e-8))
    frac = bad.mean(); worst = max(worst, frac if N >= 3000 else 0)
    if bad.sum() > max(3, 3e-3 * N):
        fails += 1
        print(f'CASE {case} W={W} H={H} N={N} phys={phys} kw={kw}: {bad.sum()} bad of {N}', flush=True)
print(f'fuzz forward: {n_cases} cases, {fails} failures, worst mismatch fraction at N>=3000: {worst:.2e}', flush=True)
