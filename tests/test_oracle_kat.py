"""Analytic known-answer tests of the oracle's Env.step / forward glue (SURVEY §8c list):
the part of the path no reference-made vector pins."""
import numpy as np

from oracle import cpu_ref as R


def _env(W=6, H=6, agents_xy=(), alive=None, food=0.0, chem=0.0, N=None, **dyn):
    medium = np.zeros((3, W, H))
    medium[R.M_FOOD] = food
    medium[R.M_CHEM] = chem
    K = len(agents_xy)
    N = N or max(K, 1)
    agents = np.zeros((4, N))
    for k, (x, y) in enumerate(agents_xy):
        agents[:, k] = [x, y, 1. if alive is None else alive[k], 0.5]
    return R.RefEnv(medium, agents, R.RefDynamics(**dyn))


def test_cell_table():
    assert list(R.cell([0., 0.1, 0.1 + 1e-12, 0.5, 1., -3., 9.], 6)) == [0, 1, 1, 3, 5, 0, 5]
    assert list(R.cell([0.0999], 6)) == [0]          # 0.0999*5+.5 = .9995
    assert list(R.cell([0.3], 6)) == [2]             # tie 0.3*5 = 1.5 → larger index


def test_move_wraps_all_slots_including_dead():
    e = _env(agents_xy=[(0.9, 0.1), (0.2, 0.95)], alive=[1, 0])
    act = np.array([[0.2, -0.3], [-0.2, 0.1], [0., 0.]])
    e.agent_move(act)
    assert np.allclose(e.agents[:2], [[0.1, 0.9], [0.9, 0.05]])
    e2 = _env(agents_xy=[(0.9, 0.1)], boundary='limit')
    e2.agent_move(np.array([[0.2], [-0.2], [0.]]))
    assert np.allclose(e2.agents[:2, 0], [1.0, 0.0])


def test_last_writer_wins_on_collision_and_layout():
    # slots 0,1,3 share cell (1,1); slot 2 is elsewhere; slot 4 is dead on the shared cell
    xy = [(0.2, 0.2), (0.21, 0.19), (0.8, 0.6), (0.2, 0.21), (0.2, 0.2)]
    e = _env(agents_xy=xy, alive=[1, 1, 1, 1, 0], chem=1.0)
    act = np.zeros((3, 5))
    act[R.U_DEP] = [10., 20., 30., 40., 50.]
    e.agent_deposit_and_layout(act)
    chem = e.medium[R.M_CHEM]
    assert chem[1, 1] == 1.0 + 40.          # highest ALIVE slot wins, no accumulation, dead ignored
    assert chem[4, 3] == 1.0 + 30.
    assert chem.sum() == 36 + 70.
    a = e.medium[R.M_AGENTS]
    assert a.sum() == 2 and a[1, 1] == 1 and a[4, 3] == 1


def test_feed_duplicates_on_shared_cells_and_dead_slots_consume():
    xy = [(0.2, 0.2), (0.2, 0.2), (0.8, 0.6), (0.2, 0.2), (0.6, 0.0)]
    e = _env(agents_xy=xy, alive=[1, 1, 1, 0, 0], food=0.5)
    act = np.zeros((3, 5))
    act[:, 0] = [0.003, 0.004, 2.0]
    e.agent_deposit_and_layout(act)
    gained = e.agent_feed(act)
    burned0 = 0.02 * 2.0 + 0.01 * 0.005
    # both co-located alive agents AND the dead slot on the occupied cell get the full .05
    assert np.allclose(gained, [0.05 - burned0, 0.05, 0.05, 0.05, 0.0])
    food = e.medium[R.M_FOOD]
    assert np.isclose(food[1, 1], 0.45) and np.isclose(food[4, 3], 0.45) and np.isclose(food[3, 0], 0.5)
    assert np.isclose(food.sum(), 36 * 0.5 - 2 * 0.05)     # the field loses it once per cell
    e2 = _env(agents_xy=xy[:1], food=0.5, food_infinite=True)
    e2.agent_deposit_and_layout(act[:, :1])
    e2.agent_feed(act[:, :1])
    assert (e2.medium[R.M_FOOD] == 0.5).all()


def test_gaussian_impulse_on_torus_corner():
    chem = np.zeros((6, 6))
    chem[0, 0] = 1.0
    out = R.diffuse_decay(chem, 0.5, 0.1)
    w = R.gaussian_weights(0.5)
    k = np.zeros(6)
    for i, o in enumerate(range(-2, 3)):
        k[o % 6] += w[i]
    assert np.allclose(out, 0.9 * np.outer(k, k), rtol=1e-13)
    assert np.isclose(out.sum(), 0.9)                        # mass × (1 − decay)


def test_step_reward_counts_every_slot_and_info():
    xy = [(0.2, 0.2), (0.8, 0.6)]
    e = _env(agents_xy=xy, food=0.5, N=5)                    # 3 dead slots at (0,0), unoccupied cell
    act = np.zeros((3, 5))
    act[R.U_DX] = 0.02                                       # dead slots move and burn too
    obs, reward, term, trunc, info = e.step(act)
    want = 2 * 0.05 - 5 * 0.01 * 0.02
    assert np.isclose(reward, want)
    assert info == {'num_agents': 2, 'reward': np.round(want, 3), 'mean_reward': np.round(want / 2, 5)}
    assert term is False and trunc is False
    assert np.allclose(e.agents[R.A_X], [0.22, 0.82, 0.02, 0.02, 0.02])
    assert obs[1].shape == (3, 6, 6) and obs[1] is not e.medium


def test_physarum_forward_probe_clamps_and_deposit_mask():
    W = H = 8
    N = 3
    medium = np.zeros((3, W, H))
    medium[R.M_CHEM] = np.add.outer(np.arange(W) ** 2, np.zeros(H)) * 0.1     # gradient along +x
    medium[R.M_FOOD] = 0.25
    agents = np.zeros((4, N))
    agents[:, 0] = [0.5, 0.5, 1, 1]
    agents[:, 1] = [0.98, 0.5, 1, 1]      # probe beyond the edge → clamped to the last row (one-sided grad)
    agents[:, 2] = [0.5, 0.5, 0, 0]       # dead slot still acts
    a = R.RefPhysarumAgent(N, scale=0.01, sense_offset=0.2, init_noise=np.array([[1., 1., 1.], [1e-3, 1e-3, -1.]]))
    assert np.allclose(a._direction_rads, [0., 0., -np.pi / 6 * 2])           # floor(angle/30°)·30°
    act = a.forward((agents, medium), turn_sign=np.array([1., -1., 1.]))
    # gradient direction is exactly +x → drads == 0 → "undetermined gradient": random turn, mask .1
    assert np.allclose(a._direction_rads, [np.pi / 6, -np.pi / 6, -np.pi / 6], atol=1e-12)
    assert np.allclose(act[R.U_DEP], 4.0 * 0.25 * 0.1)
    assert np.allclose(np.hypot(act[0], act[1]), 0.01)


def test_physarum_turn_truth_table():
    N = 6
    a = R.RefPhysarumAgent(N, init_noise=np.ones((2, N)))
    a._direction_rads = np.zeros(N)
    atol = np.radians(30) * 0.1
    # gradient angle = -delta (delta = dir - drads); cases: inside tol, just outside, left, right, unseen, zero grad
    drads = np.array([-0.5 * atol, -(atol / 0.99 + 1e-6), -0.5, 0.5, 2.0, 0.0])
    g = np.stack([np.cos(drads), np.sin(drads)])
    g[:, 5] = 0
    sign = np.array([1., 1., 1., 1., -1., -1.])
    out = a._process_gradient(g, sign)
    new_dir = np.arctan2(out[1], out[0])
    t = np.pi / 6
    assert np.allclose(new_dir, [t, -t, -t, t, -t, -t])
    assert list(a._deposit_mask) == [False, True, True, True, True, False]
    assert list(a.last_undetermined) == [True, False, False, False, True, True]


def test_brownian_formula_and_alive_mask():
    agents = np.zeros((4, 4))
    agents[R.A_ALIVE] = [1, 1, 0, 1]
    b = R.RefBrownianAgent(move_scale=0.01, deposit_scale=0.5)
    units = (np.array([0, 1000, 500, 250]), np.array([500, 500, 500, 500]), np.array([0, 1000, 1, 2]))
    act = b.forward((agents, None), units=units)
    assert np.allclose(act[0], [-0.01, 0.01, 0, -0.005])
    assert np.allclose(act[1], 0) and np.allclose(act[2], [0, 0.5, 0, 0.001])


def test_agents_from_medium_row_major_and_synthetic_init():
    medium, agents = R.synthetic_init(16, 12, 0.15, seed=1234)
    ix, iy = medium[R.M_AGENTS].nonzero()
    K = len(ix)
    assert 10 < K < 60 and agents.shape == (4, 16 * 12)
    assert np.allclose(agents[R.A_X, :K], ix / 15.) and np.allclose(agents[R.A_Y, :K], iy / 11.)
    assert (agents[R.A_ALIVE, :K] == 1).all() and (agents[:, K:] == 0).all()
    assert (agents[R.A_FOOD, :K] >= 0.1).all() and (agents[R.A_FOOD, :K] <= 1.0).all()
    f = medium[R.M_FOOD]
    assert f.min() == 0 and 0.3 < f.max() <= 0.5 and 0.3 < (f == 0).mean() < 0.7
    assert np.array_equal(f, np.round(f, 3))
    _, a2 = R.synthetic_init(16, 12, 0.15, seed=1234, max_agents=K)
    assert a2.shape == (4, K)


def test_perlin_noise_properties():
    """oracle.perlin2 (the stand-in for the un-vendored perlin_noise package of core/data_init.py:190-196): gradient noise
    is exactly 0 on lattice points, bounded by sqrt(1/2), continuous, and fixed by the seed."""
    xs = np.arange(0, 9, dtype=np.float64)
    g = R.perlin2(5, xs[:, None] * np.ones((1, 9)), np.ones((9, 1)) * xs[None, :])
    assert np.array_equal(g, np.zeros((9, 9)))
    t = np.linspace(0, 8, 4001)
    f = R.perlin2(5, t[:, None] * np.ones((1, 4)), np.array([[0.25, 1.5, 3.75, 7.1]]) * np.ones((4001, 1)))
    assert np.abs(f).max() <= np.sqrt(0.5) + 1e-12 and np.abs(f).max() > 0.2
    assert np.abs(np.diff(f, axis=0)).max() < 0.01                      # smooth at 1/500 of a lattice cell
    assert np.array_equal(f, R.perlin2(5, t[:, None] * np.ones((1, 4)), np.array([[0.25, 1.5, 3.75, 7.1]]) * np.ones((4001, 1))))
    assert not np.array_equal(f, R.perlin2(6, t[:, None] * np.ones((1, 4)), np.array([[0.25, 1.5, 3.75, 7.1]]) * np.ones((4001, 1))))
    food = R.perlin_field(64, 48, 8, 1234, threshold=1.0)
    assert food.min() >= 0 and 0.3 < (food > 0).mean() < 0.7 and np.array_equal(food, np.round(food, 3))
    assert (R.perlin_field(64, 48, 8, 1234, threshold=0.1) <= 0.1).all()


def test_perlin3_noise_properties_and_sequence():
    """oracle.perlin3 (PerlinNoiseSequence, core/data_init.py:55-69): 0 on lattice points, bounded, continuous, fixed by the
    seed; the sequence cycles over arange(*t_bounds, dt) and its flow operator is scale·field + (1 − decay)·current."""
    xs = np.arange(0, 5, dtype=np.float64)
    assert np.array_equal(R.perlin3(5, xs, xs[::-1], xs), np.zeros(5))
    t = np.linspace(0, 6, 3001)
    f = R.perlin3(5, t, 0.37 + 0 * t, 1.2 + 0 * t)
    assert 0.2 < np.abs(f).max() <= 1.0 and np.abs(np.diff(f)).max() < 0.01
    assert np.array_equal(f, R.perlin3(5, t, 0.37 + 0 * t, 1.2 + 0 * t)) and not np.array_equal(f, R.perlin3(6, t, 0.37 + 0 * t, 1.2 + 0 * t))
    # a slice at integer t·octaves is a different 2-D noise for every lattice plane, continuous in t
    a, b = R.perlin3_field(16, 12, 0.25, 8, 3), R.perlin3_field(16, 12, 0.2501, 8, 3)
    assert np.array_equal(a, np.round(a, 3)) and np.abs(a - b).max() <= 0.003 and np.abs(a).max() > 0.1
    seq = R.RefPerlinNoiseSequence((16, 12), dt=0.25, t_bounds=(0, 0.75), octaves=8, seed=3)
    flow = seq.get_flow_operator(scale=0.5, decay=0.25)
    cur = np.full((16, 12), 0.4)
    for k in range(4):                                                   # the fourth call wraps around to t = 0
        want = 0.5 * R.perlin3_field(16, 12, [0.0, 0.25, 0.5, 0.0][k], 8, 3) + 0.75 * cur
        cur = flow(cur)
        assert np.array_equal(cur, want)


def test_step_substep_order_is_observable_through_the_food_flow():
    """core/env.py:101-131: move → deposit + layout → feed → lifecycle → food flow → diffuse/decay.  With a food-flow
    operator that is not the identity the order shows: feeding sees the food BEFORE the flow, the flow acts on the
    food AFTER consumption, the deposit lands on the cell the agent MOVED to and is diffused in the same step."""
    W = H = 6
    medium = np.zeros((3, W, H))
    medium[1] = 0.4
    agents = np.zeros((4, 2))
    agents[:, 0] = [2 / 5, 3 / 5, 1., 0.5]                      # on cell (2, 3)
    action = np.array([[1 / 5, 0.], [0., 0.], [2.0, 0.]])       # moves to (3, 3), deposits 2.0; slot 1 is dead and idle
    flow = lambda food: 2.0 * food + 0.1
    env = R.RefEnv(medium, agents, R.RefDynamics(op_food_flow=flow))
    _, reward, term, _, info = env.step(action)
    eaten = 0.1 * 0.4
    food = np.full((W, H), 0.4)
    food[3, 3] -= eaten                                         # consumption at the NEW cell, before the flow
    assert np.allclose(env.medium[1], flow(food), rtol=0, atol=1e-15)
    burned = 0.02 * 2.0 + 0.01 * (1 / 5)
    assert np.isclose(env.agents[3, 0], 0.5 + eaten - burned, rtol=0, atol=1e-15) and np.isclose(reward, eaten - burned)
    chem = np.zeros((W, H))
    chem[3, 3] = 2.0
    assert np.allclose(env.medium[2], R.diffuse_decay_explicit(chem, 0.5, 0.1), rtol=1e-12, atol=1e-15)
    assert env.medium[0, 3, 3] == 1 and env.medium[0].sum() == 1 and info['num_agents'] == 1 and not term
