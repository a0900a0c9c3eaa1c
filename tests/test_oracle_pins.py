"""Pins for the CPU oracle: third-party primitives the reference calls
(pandas nearest lookup, scipy gaussian filter, numpy gradient) and golden vectors made by
the reference's own pure-numpy function bodies (tests/golden/make_ref_helper_vectors.py)."""
import os

import numpy as np
import pandas as pd
import pytest
import scipy.ndimage

from oracle import cpu_ref as R


@pytest.fixture(scope='module')
def G(golden_dir):
    return np.load(os.path.join(golden_dir, 'ref_helpers.npz'))


@pytest.mark.parametrize('n', [2, 3, 12, 16, 256, 1024, 4096])
def test_cell_equals_pandas_nearest(n):
    """core/utils.py:53 `field.sel(..., method='nearest')` → pandas get_indexer(nearest)."""
    rs = np.random.RandomState(n)
    labels = np.linspace(0, 1, n)
    v = np.concatenate([rs.uniform(-0.3, 1.3, 200000), labels, (labels[:-1] + labels[1:]) / 2,
                        [0., 1., -5., 7., 1 - 1e-16, 1e-300]])
    want = pd.Index(labels).get_indexer(v, method='nearest')
    got = R.cell(v, n)
    bad = (want != got)
    # float midpoints between two labels are ties up to rounding of |v - label|: there (and
    # only there) the two may pick different neighbours of the same pair
    mids = slice(200000 + n, 200000 + 2 * n - 1)
    assert not bad[:200000 + n].any() and not bad[200000 + 2 * n - 1:].any()
    assert (np.abs(got[mids][bad[mids]] - want[mids][bad[mids]]) == 1).all()


def test_cell_on_fixed_point_grid_matches_integer_formula():
    """Device coordinates are Q0.32 fixed point; the integer formula the kernels use must
    equal the oracle's float formula on those coordinates (bit-exact index work)."""
    rs = np.random.RandomState(1)
    for n in (2, 7, 256, 4096, 16384):
        X = rs.randint(0, 2 ** 32, size=300000, dtype=np.uint64)
        X[:4] = [0, 2 ** 32 - 1, 2 ** 31, 2 ** 31 - 1]
        k = np.arange(n, dtype=np.uint64)
        X[4:4 + n] = ((k * np.uint64(2 ** 32)) // np.uint64(n - 1)).clip(0, 2 ** 32 - 1)[:len(X) - 4][:n]
        integer = ((X * np.uint64(n - 1) + np.uint64(2 ** 31)) >> np.uint64(32)).astype(np.int64)
        assert (integer == R.cell(X.astype(np.float64) / 2.0 ** 32, n)).all()


@pytest.mark.parametrize('sigma', [0.5, 0.8, 1.0])
def test_diffuse_matches_scipy_and_explicit_weights(sigma):
    rs = np.random.RandomState(3)
    chem = rs.rand(37, 23)
    a = R.diffuse_decay(chem, sigma, 0.1)
    b = R.diffuse_decay_explicit(chem, sigma, 0.1)
    c = scipy.ndimage.gaussian_filter(chem, sigma, mode='wrap') * 0.9
    assert np.allclose(a, b, rtol=1e-13, atol=1e-15)
    assert np.array_equal(a, c)


def test_gaussian_weights_values():
    """σ=.5 → radius 2 and these 5 taps; σ=.8 → radius 3."""
    w = R.gaussian_weights(0.5)
    assert np.allclose(w, [2.63865083e-4, 0.106450772, 0.786570726, 0.106450772, 2.63865083e-4], rtol=1e-8)
    assert len(R.gaussian_weights(0.8)) == 7
    from scipy.ndimage._filters import _gaussian_kernel1d
    assert np.allclose(w, _gaussian_kernel1d(0.5, 0, 2))


def test_gradient_field_is_np_gradient_normalised():
    rs = np.random.RandomState(4)
    chem = rs.rand(9, 11)
    chem[3:6, 3:6] = 0.25                       # flat patch → zero gradient → masked, 0/0 → 0
    g = R.gradient_field(chem)
    gx, gy = np.gradient(chem)
    assert g.shape == (2, 9, 11)
    assert (g[:, 4, 4] == 0).all()
    n = np.hypot(gx, gy)
    ok = n >= 1e-5
    assert np.allclose(g[0][ok], (gx / n)[ok]) and np.allclose(g[1][ok], (gy / n)[ok])
    # one-sided edges (not periodic)
    assert np.isclose(gx[0, 2], chem[1, 2] - chem[0, 2]) and np.isclose(gx[-1, 2], chem[-1, 2] - chem[-2, 2])


# ---------------------------------------------------------------- reference-made vectors
def test_ref_renormalize_discretize_polar(G):
    assert np.array_equal(R.renormalize_radians(G['renorm_in']), G['renorm_out'])
    assert np.array_equal(R.discretize(G['disc_in'], float(G['disc_step'])), G['disc_out'])
    r, th = R.xy2polar(G['xy_in'][0], G['xy_in'][1])
    assert np.array_equal(r, G['xy2polar_r']) and np.array_equal(th, G['xy2polar_theta'])
    assert np.array_equal(R.get_radians(G['xy_in']), G['get_radians_out'])
    px, py = R.polar2xy(0.03, th)
    assert np.array_equal(px, G['polar2xy_x']) and np.array_equal(py, G['polar2xy_y'])


@pytest.mark.parametrize('ci', [0, 1, 2])
def test_ref_physarum_turn_logic(G, ci):
    """PhysarumAgent._discrete_turn/_choose_turn/_process_deposit (core/agent/gradient.py:168-214)
    run from the reference file vs the oracle class, same random 0/1 draws."""
    turn_angle, sense_angle, rtol, normalized = G[f'turn{ci}_params']
    N = G[f'turn{ci}_dir0'].shape[0]
    a = R.RefPhysarumAgent(N, turn_angle=turn_angle, sense_angle=sense_angle, turn_tolerance=rtol,
                           normalized_grad=bool(normalized), init_noise=np.ones((2, N)))
    a._direction_rads = G[f'turn{ci}_dir0'].copy()
    sign = (G[f'turn{ci}_rand01'] - 0.5) * 2
    out = a._process_gradient(G[f'turn{ci}_grad_in'], sign)
    assert np.array_equal(out, G[f'turn{ci}_grad_out'])
    assert np.array_equal(a._deposit_mask, G[f'turn{ci}_deposit_mask'])
    assert np.array_equal(a._process_deposit(G[f'turn{ci}_food']), G[f'turn{ci}_deposit_out'])


def test_ref_momentum(G):
    a = R.RefGradientAgent(300, inertia=0.9, noise_scale=0.025, init_noise=G['mom_prev'])
    out = a._process_momentum(G['mom_in'].copy(), G['mom_noise'])
    assert np.array_equal(out, G['mom_out'])
    assert np.array_equal(a._prev_grad, G['mom_out'])


def test_ref_boundary_mask_random_wave(G):
    for b in ('wrap', 'limit'):
        assert np.array_equal(R.move_handle_boundary(G['boundary_in'], b), G[f'boundary_{b}_out'])
    assert np.array_equal(R.mask_range(G['mask_in'], mask_above=0.15), G['mask_out_ratio015'])
    assert np.array_equal(R.agents_channel_from_uniform(G['mask_in'], 0.15), G['agents_ch_ratio015'])
    # get_random(size, .1, 1.) == (1-.1)*u.round(3)+.1  — the agent_food formula of agents_from_medium
    assert np.array_equal(0.9 * G['get_random_raw'].round(3) + 0.1, G['get_random_out_01_1'])
    assert np.allclose(R.wave_field(9, 6, 0.25), G['wave_9x6_t025'], rtol=0, atol=0)
    assert np.array_equal(G['meshgrid_5x7'].shape, (2, 5, 7))


def test_ref_data_initializer_builder_chain(G):
    """The builder chain of core/data_init.py:171-253 — __init__ → with_noise ×2 → with_agents → build_numpy, the static mask
    of build / build_agents, _add_masked, and BrownianAgent's chain over (N,) with the alive mask (core/agent/static.py:40-50)
    — outputs of the reference's OWN method bodies on seeded numpy draws (tests/golden/make_ref_helper_vectors.py); the
    oracle's restatement, fed the same uniforms, must reproduce them bit for bit (A17 / A18's
    `(b − a)·u.round(3) + a` × mask composition pinned by reference output)."""
    raw = iter(G['builder_raw'])
    b = R.RefDataInitializer((7, 5), ('agents', 'env_food', 'chem1'), draw=lambda size: next(raw))
    b.with_noise('env_food', 0.1, 0.6).with_noise('chem1', -2, 3).with_agents(0.3)
    assert np.array_equal(b.build_numpy(), G['builder_numpy'])
    assert set(np.unique(G['builder_numpy'][0])) <= {0., 1.} and 0 < G['builder_numpy'][0].sum() < 35
    assert np.array_equal(b.build_numpy() * G['builder_mask'], G['builder_masked'])
    b._static_mask = G['builder_mask']
    assert np.array_equal(b.build(), G['builder_masked'])
    b._add_masked('env_food', np.full((7, 5), 0.25))
    assert np.array_equal(b.build_numpy()[1], G['builder_add_masked'])
    raw_a = iter(G['builder_action_raw'])
    ba = R.RefDataInitializer(40, ('dx', 'dy', 'deposit1'), mask=G['builder_action_alive'], draw=lambda size: next(raw_a))
    ba.with_noise('dx', -0.01, 0.01).with_noise('dy', -0.01, 0.01).with_noise('deposit1', 0, 0.5)
    assert np.array_equal(ba.build(), G['builder_action_out'])
    assert (G['builder_action_out'][:, G['builder_action_alive'] == 0] == 0).all()
    # the formula the device kernels and oracle/rng.py use for one with_noise call (orng.builder_noise)
    assert np.array_equal(0.5 * G['builder_raw'][0].round(3) + 0.1, G['builder_numpy'][1])


def test_ref_flow_operator_sequence_and_field_trace(G):
    """FieldSequence.get_flow_operator + __iter__ (core/data_init.py:29-38, cycling over the time axis) and
    FieldTrace.update (core/render.py:29-30), outputs of the reference's own function bodies."""
    flow = R.RefWaveSequence((7, 5), dt=0.25, t_bounds=(0, 0.75)).get_flow_operator(scale=0.5, decay=0.25)
    cur = G['flow_in']
    for k in range(4):
        cur = flow(cur)
        assert np.array_equal(cur, G[f'flow_out{k}']), k
    from oracle.render_ref import FieldTrace
    tr = FieldTrace((6, 4), trace_steps=8)
    for k in range(3):
        tr.update(G['trace_fields'][k])
        assert np.array_equal(tr.trace, G[f'trace_out{k}'])


def test_ref_convolution_model(G):
    """ConvolutionModel.forward (core/agent/evo.py:45-118), outputs of the reference's own class body run with torch
    (float32): the oracle's circular convolution stack + tanh, evaluated in float64, agrees within 1e-5."""
    for ci in range(4):
        x = G[f'nca{ci}_in'].astype(np.float64)
        ws = [G[f'nca{ci}_w{li}'].astype(np.float64) for li in range(int(G[f'nca{ci}_nw']))]
        y = x
        for w in ws:
            y = R.conv2d_circular(y, w)
        assert np.allclose(np.tanh(y), G[f'nca{ci}_out'], rtol=1e-5, atol=1e-5), ci
        if x.shape[0] == 3:
            assert np.allclose(R.nca_sense(x, ws, True), G[f'nca{ci}_out'], rtol=1e-5, atol=1e-5)
