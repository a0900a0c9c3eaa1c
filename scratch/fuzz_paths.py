"""Randomised check that every switchable device path gives identical bits: fused step / staged step /
store+repair claim / lazy-fused forward / eager forward / re-sorted arrays."""
import sys, os; sys.path.insert(0, '.')
import numpy as np, torch, die_amd
from tests.test_gpu_parity import random_state, f32
rs = np.random.RandomState(int(os.environ.get('FUZZ_SEED', '0')))
fails = 0
for case in range(int(os.environ.get('FUZZ_CASES', '60'))):
    W = int(rs.choice([4, 16, 33, 64, 128])); H = int(rs.choice([4, 8, 12, 64, 244, 252, 256]))
    N = int(rs.choice([5, 300, 4000])); K = int(rs.randint(0, N + 1))
    medium, agents = random_state(W, H, N, K, rs, collide=float(rs.choice([0.0, 0.5])))
    dyn = dict(boundary=die_amd.BoundaryCondition(rs.choice(['wrap', 'limit'])), agents_die=bool(rs.rand() < 0.3), food_infinite=bool(rs.rand() < 0.2),
               diffuse_sigma=float(rs.choice([0.5, 0.8])))
    phys = rs.rand() < 0.7
    turn = np.radians(30); dir0 = f32(np.floor(rs.uniform(-np.pi, np.pi, N) / turn) * turn); prev = f32(rs.normal(0, .4, (2, N)))
    outs = []
    for variant in ('default', 'DIE_NO_FUSED_STEP', 'DIE_STORE_CLAIM', 'eager', 'sorted'):
        if variant.startswith('DIE_'): os.environ[variant] = '1'
        env = die_amd.Env.from_numpy(medium, agents, die_amd.Dynamics(**dyn), sort_every=1 if variant == 'sorted' else 0)
        if phys: ag = die_amd.PhysarumAgent(max_agents=N, seed=3, scale=2.0 / max(W, H), sense_offset=0.05)
        else: ag = die_amd.GradientAgent(max_agents=N, seed=3, scale=0.01, sense_offset=0.03, inertia=0.8, noise_scale=0.02)
        ag.set_state(dir0, None if phys else prev)
        ag.lazy = variant != 'eager'
        obs = env._get_current_obs
        for _ in range(4):
            obs, *_ = env.step(ag.forward(obs))
        outs.append((env.medium.to_numpy(), env.agents.to_numpy(), ag.direction_rads_numpy()))
        if variant.startswith('DIE_'): del os.environ[variant]
    for v, o in zip(('staged', 'store-claim', 'eager', 'sorted'), outs[1:]):
        for a, b in zip(outs[0], o):
            if not np.array_equal(a, b):
                fails += 1; print(f'CASE {case} variant {v} differs: W={W} H={H} N={N} K={K} phys={phys} dyn={dyn}', flush=True); break
print(f'fuzz paths: {fails} failures', flush=True)
