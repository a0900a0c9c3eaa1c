import sys, os; sys.path.insert(0, '.')
import numpy as np, torch, die_amd
from oracle import cpu_ref as R
rs = np.random.RandomState(int(os.environ.get('FUZZ_SEED', '0')))
fails = 0
for case in range(int(os.environ.get('FUZZ_CASES', '60'))):
    W = int(rs.choice([2, 3, 17, 64, 100, 255, 256, 513])); H = int(rs.choice([2, 5, 12, 64, 127, 128, 300, 1024]))
    ratio = float(rs.choice([0.0, 0.001, 0.05, 0.15, 0.5, 1.0])); seed = int(rs.randint(0, 2 ** 31)) * int(rs.choice([1, 2 ** 20 + 7]))
    try:
        want_m, want_a = R.synthetic_init(W, H, ratio, seed)
        K = int(want_m[0].sum())
        if K == 0:
            continue
        env = die_amd.Env((W, H), die_amd.Dynamics(init_agent_ratio=ratio), seed=seed)
        m, a = env.medium.to_numpy(), env.agents.to_numpy()
        assert np.array_equal(m[0], want_m[0]), 'seeding'
        assert env._num_seeded == K
        assert np.abs(a[:2] - want_a[:2]).max() <= 2.0 ** -32, 'xy'
        assert np.array_equal(a[2], want_a[2]) and np.allclose(a[3], want_a[3], rtol=1e-6), 'alive/food'
        d = np.abs(m[1] - want_m[1]); assert (d > 1e-6).mean() <= 1e-4 and d.max() <= 1.001e-3, 'food field'
        ag = die_amd.PhysarumAgent(max_agents=W * H, seed=seed); ag._alloc_state('cuda:0')
        ref = R.RefPhysarumAgent(W * H, seed=seed)
        assert np.mean(np.abs(ag.direction_rads_numpy() - ref._direction_rads) > 1e-6) <= 2e-4, 'heading'
    except Exception as e:
        fails += 1; print(f'CASE {case} W={W} H={H} ratio={ratio} seed={seed}: {type(e).__name__} {e}', flush=True)
print(f'fuzz init: {fails} failures', flush=True)
