"""Randomised single-step parity sweep (GPU vs oracle) over shapes, parameters and occupancy — a
one-off shake-out run, not part of the test suite."""
import sys, os, traceback; sys.path.insert(0, '.')
import numpy as np, torch
import die_amd
from oracle import cpu_ref as R
from tests.test_gpu_parity import random_state, quantised_action, ref_dyn, f32

rs = np.random.RandomState(int(os.environ.get('FUZZ_SEED', '0')))
n_cases = int(os.environ.get('FUZZ_CASES', '300'))
fails = 0
skipped = 0
for case in range(n_cases):
    W = int(rs.choice([2, 3, 4, 5, 7, 8, 12, 16, 31, 32, 33, 64, 100, 128, 250, 256]))
    H = int(rs.choice([2, 3, 4, 8, 12, 16, 20, 36, 60, 64, 128, 244, 248, 252, 256, 260, 500, 512]))
    N = int(rs.choice([1, 2, 5, 64, 257, 1000, 4096, 20000]))
    K = int(rs.randint(0, N + 1))
    sigma = float(rs.choice([0.3, 0.5, 0.8, 1.0, 1.2]))
    dyn = die_amd.Dynamics(boundary=die_amd.BoundaryCondition(rs.choice(['wrap', 'limit'])), food_infinite=bool(rs.rand() < 0.2),
                           agents_die=bool(rs.rand() < 0.2), op_action_cost=die_amd.zero_cost if rs.rand() < 0.2 else die_amd.linear_action_cost,
                           diffuse_sigma=sigma, rate_feed=float(rs.choice([0.1, 0.5])), rate_decay_chem=float(rs.choice([0.0, 0.1, 0.3])))
    sort_every = int(rs.choice([0, 1]))
    try:
        medium, agents = random_state(W, H, N, K, rs, collide=float(rs.choice([0.0, 0.3, 0.9])))
        action = quantised_action(N, rs, float(rs.choice([0.5 / max(W, H), 3.0 / max(W, H), 0.4])))
        rd = ref_dyn(dyn)
        for f in ('rate_feed', 'rate_decay_chem', 'diffuse_sigma'):
            setattr(rd, f, float(np.float32(getattr(rd, f))))
        ref = R.RefEnv(medium, agents, rd)
        env = die_amd.Env.from_numpy(medium, agents, dyn, sort_every=sort_every)
        for step in range(2):
            _, want_r, want_t, _, want_i = ref.step(action)
            _, r, t, _, i = env.step(action)
        ga, gm = env.agents.to_numpy(), env.medium.to_numpy()
        tol_xy = 2.0 ** -32 if dyn.boundary.value == 'limit' else 0.0
        assert np.abs(ga[:2] - ref.agents[:2]).max() <= tol_xy, 'xy'
        assert np.array_equal(ga[2], ref.agents[2]), 'alive'
        assert np.array_equal(gm[0], ref.medium[0]), 'agents channel'
        assert i['num_agents'] == want_i['num_agents'] and t == want_t, 'info'
        assert np.allclose(ga[3], ref.agents[3], rtol=1e-5, atol=2e-7), 'agent_food'
        assert np.allclose(gm[1], ref.medium[1], rtol=1e-5, atol=1e-8), 'food'
        assert np.allclose(gm[2], ref.medium[2], rtol=2e-5, atol=2e-7), 'chem'
        assert abs(r - want_r) <= 1e-5 * np.abs(ref.last_gained).sum() + 1e-9, 'reward'
    except NotImplementedError as e:
        skipped += 1
        continue
    except Exception as e:
        fails += 1
        print(f'CASE {case} FAILED W={W} H={H} N={N} K={K} sigma={sigma} dyn={dyn} sort={sort_every}: {type(e).__name__} {e}', flush=True)
        if fails > 10:
            break
print(f'fuzz: {n_cases} cases, {skipped} unsupported, {fails} failures', flush=True)
