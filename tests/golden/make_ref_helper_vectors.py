"""Generates tests/golden/ref_helpers.npz — golden vectors produced by the REFERENCE'S OWN
function bodies, run in the build container (where /root/reference is mounted).

The reference package cannot be imported (its modules import xarray / skimage /
gymnasium / perlin_noise / evotorch at top level, none installed), but several functions
on the hot path are pure numpy.  This script parses the reference files with `ast`,
compiles ONLY those function definitions (nothing is copied into this repo — the source
text is read from /root/reference at generation time), runs them on seeded inputs and
stores inputs + outputs.  tests/test_oracle_pins.py then checks oracle/cpu_ref.py
against the stored outputs, on CPU, with no access to the reference.

Run:  python tests/golden/make_ref_helper_vectors.py   (only where /root/reference exists)
"""
import ast
import os
import sys
from enum import Enum

import numpy as np

REF = os.environ.get('DIE_REFERENCE', '/root/reference')
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ref_helpers.npz')


def _functions(path, names, cls=None):
    """Compile the named top-level functions (or methods of `cls`) of a reference file."""
    with open(os.path.join(REF, path)) as f:
        tree = ast.parse(f.read())
    body = tree.body
    if cls is not None:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    picked = []
    for n in body:
        if isinstance(n, ast.FunctionDef) and n.name in names:
            n.decorator_list = []          # staticmethod/property decorators are re-applied by hand
            n.returns = None
            for a in n.args.args + n.args.kwonlyargs:
                a.annotation = None
            picked.append(n)
    assert {n.name for n in picked} == set(names), (path, names)
    return ast.Module(body=picked, type_ignores=[])


def _exec(module, ns):
    ast.fix_missing_locations(module)
    exec(compile(module, '<reference>', 'exec'), ns)
    return ns


def main():
    rs = np.random.RandomState(20240611)
    out = {}

    # ---- core/utils.py helpers -------------------------------------------------------
    u = _exec(_functions('core/utils.py', ['polar2z', 'z2polar', 'polar2xy', 'xy2polar', 'get_radians',
                                           'renormalize_radians', 'discretize', 'get_meshgrid']),
              {'np': np, 'Sequence': None})
    rads = np.concatenate([rs.uniform(-12, 12, 500), [0., np.pi, -np.pi, 2 * np.pi, -2 * np.pi, 3.5, -3.5,
                                                       np.pi / 6, -np.pi / 6, 1e-9, -1e-9]])
    out['renorm_in'] = rads
    out['renorm_out'] = u['renormalize_radians'](rads)
    step = np.radians(30)
    out['disc_in'] = rads
    out['disc_step'] = step
    out['disc_out'] = u['discretize'](rads, step)
    xy = rs.normal(0, .4, size=(2, 600))
    xy[:, :4] = [[0, 1, -1, 0], [0, 0, 0, -1]]
    out['xy_in'] = xy
    r_, th_ = u['xy2polar'](xy[0], xy[1])
    out['xy2polar_r'], out['xy2polar_theta'] = r_, th_
    out['get_radians_out'] = u['get_radians'](xy)
    px, py = u['polar2xy'](0.03, th_)
    out['polar2xy_x'], out['polar2xy_y'] = px, py
    out['meshgrid_5x7'] = u['get_meshgrid']((5, 7))

    # ---- PhysarumAgent turn logic (core/agent/gradient.py) ---------------------------
    ns = {'np': np, 'xy2polar': u['xy2polar'], 'polar2xy': u['polar2xy'], 'get_radians': u['get_radians'],
          'renormalize_radians': u['renormalize_radians'], 'discretize': u['discretize']}
    phys = _exec(_functions('core/agent/gradient.py',
                            ['_choose_turn', '_discrete_turn', '_discretize_grad', '_process_deposit'],
                            cls='PhysarumAgent'), dict(ns))
    grad_m = _exec(_functions('core/agent/gradient.py', ['_process_momentum'], cls='GradientAgent'), dict(ns))

    class Stub:
        pass
    for name in ('_choose_turn', '_discrete_turn', '_discretize_grad', '_process_deposit'):
        setattr(Stub, name, phys[name])
    Stub._process_momentum = grad_m['_process_momentum']

    cases = [(30, 90, 0.1, True), (35, 120, 0.05, True), (45, 60, 0.2, False)]
    for ci, (turn_angle, sense_angle, rtol, normalized) in enumerate(cases):
        N = 4000
        s = Stub()
        s._turn_radians = np.radians(turn_angle)
        s._sense_radians = np.radians(sense_angle)
        s._rtol = rtol
        s._normalized = normalized
        prev = rs.normal(0, .4, size=(2, N))
        s._direction_rads = s._discretize_grad(prev)
        out[f'turn{ci}_params'] = np.array([turn_angle, sense_angle, rtol, float(normalized)])
        out[f'turn{ci}_dir0'] = s._direction_rads.copy()
        # sampled gradients: unit vectors, zeros (masked-out gradient), +x axis (drads == 0),
        # directions placed right at the atol / sense thresholds relative to the heading
        ang = rs.uniform(-np.pi, np.pi, N)
        g = np.stack([np.cos(ang), np.sin(ang)])
        g[:, :400] = 0.
        g[:, 400:500] = [[1.], [0.]]
        atol = s._turn_radians * rtol
        k = np.arange(500, 1100)
        offs = np.tile([atol, -atol, atol / 0.99, -atol / 0.99, s._sense_radians, -s._sense_radians], 100) \
            + rs.normal(0, 1e-3, 600) * (rs.rand(600) < 0.7)
        tgt = s._direction_rads[k] - offs
        g[:, k] = [np.cos(tgt), np.sin(tgt)]
        if not normalized:
            g *= rs.uniform(0.2, 3.0, N)
        out[f'turn{ci}_grad_in'] = g
        seed = 777 + ci
        np.random.seed(seed)
        draws = np.random.randint(0, 2, N)
        out[f'turn{ci}_rand01'] = draws
        np.random.seed(seed)
        g_out = s._discrete_turn(g)
        out[f'turn{ci}_grad_out'] = np.asarray(g_out)
        out[f'turn{ci}_deposit_mask'] = np.asarray(s._deposit_mask)
        food = rs.uniform(0, .5, N)
        s._deposit = 4.0
        out[f'turn{ci}_food'] = food
        out[f'turn{ci}_deposit_out'] = np.asarray(s._process_deposit(None, food))

    # momentum: grad = (1-inertia)*grad + inertia*prev + noise_scale*noise
    s = Stub()
    s._inertia, s._noise_scale = 0.9, 0.025
    s._prev_grad = rs.normal(0, .4, size=(2, 300))
    noise = rs.normal(0, .4, size=(2, 300))
    s._get_some_noise = lambda: noise
    gin = rs.normal(0, 1, size=(2, 300))
    out['mom_prev'], out['mom_noise'], out['mom_in'] = s._prev_grad.copy(), noise, gin
    out['mom_out'] = np.asarray(s._process_momentum(gin.copy()))

    # ---- Env._agent_move_handle_boundary (core/env.py) -------------------------------
    with open(os.path.join(REF, 'core/env.py')) as f:       # the enum the method compares against
        enum_def = next(n for n in ast.parse(f.read()).body
                        if isinstance(n, ast.ClassDef) and n.name == 'BoundaryCondition')
    BoundaryCondition = _exec(ast.Module(body=[enum_def], type_ignores=[]), {'Enum': Enum})['BoundaryCondition']
    import logging
    env_ns = _exec(_functions('core/env.py', ['_agent_move_handle_boundary'], cls='Env'),
                   {'np': np, 'BoundaryCondition': BoundaryCondition, 'logging': logging})
    coords = np.concatenate([rs.uniform(-1.5, 2.5, 400), [0., 1., -0., 1. + 1e-12, -1e-12, 2., -1.]])
    for bname in ('wrap', 'limit'):
        e = Stub()
        e.dynamics = Stub()
        e.dynamics.boundary = BoundaryCondition[bname]
        out[f'boundary_{bname}_out'] = env_ns['_agent_move_handle_boundary'](e, coords.copy())
    out['boundary_in'] = coords

    # ---- DataInitializer._mask / get_random (core/data_init.py) ----------------------
    di = _exec(_functions('core/data_init.py', ['_mask', 'get_random'], cls='DataInitializer'), {'np': np})
    samp = np.round(rs.uniform(-0.2, 1.2, 1000), 3)
    out['mask_in'] = samp
    out['mask_out_ratio015'] = di['_mask'](None, samp, mask_above=0.15)
    out['agents_ch_ratio015'] = np.ceil(di['_mask'](None, samp, mask_above=0.15))
    np.random.seed(99)
    raw = np.random.random_sample(1000)
    np.random.seed(99)
    out['get_random_raw'] = raw
    out['get_random_out_01_1'] = di['get_random'](1000, 0.1, 1.0)

    # ---- WaveSequence.__getitem__ (core/data_init.py) --------------------------------
    wv = _exec(_functions('core/data_init.py', ['__getitem__'], cls='WaveSequence'), {'np': np})
    w = Stub()
    w._grid = u['get_meshgrid']((9, 6))
    out['wave_9x6_t025'] = wv['__getitem__'](w, 0.25)

    # ---- FieldSequence.get_flow_operator / __iter__ (core/data_init.py:29-38): four calls over a 3-point time axis, so
    # that the cycle wraps; WaveSequence.__getitem__ supplies the fields
    from itertools import cycle
    fs = _exec(_functions('core/data_init.py', ['get_flow_operator', '__iter__'], cls='FieldSequence'), {'np': np, 'cycle': cycle})

    class Seq:
        pass
    Seq.__iter__ = fs['__iter__']
    Seq.get_flow_operator = fs['get_flow_operator']
    Seq.__getitem__ = wv['__getitem__']
    sq = Seq()
    sq._grid = u['get_meshgrid']((7, 5))
    sq._ts = np.arange(0, 0.75, 0.25)
    flow = sq.get_flow_operator(scale=0.5, decay=0.25)
    cur = rs.uniform(0, 1, (7, 5))
    out['flow_in'] = cur.copy()
    for k in range(4):
        cur = flow(cur)
        out[f'flow_out{k}'] = np.asarray(cur).copy()

    # ---- FieldTrace (core/render.py:9-30): three updates
    ft = _exec(_functions('core/render.py', ['__init__', 'update'], cls='FieldTrace'), {'np': np})

    class Trace:
        pass
    Trace.__init__ = ft['__init__']
    Trace.update = ft['update']
    tr = Trace((6, 4), trace_steps=8)
    fields = (rs.rand(3, 6, 4) < 0.3).astype(np.float64)
    out['trace_fields'] = fields
    for k in range(3):
        tr.update(fields[k])
        out[f'trace_out{k}'] = tr._trace_field.copy()

    # ---- ConvolutionModel (core/agent/evo.py:45-118): the reference's class body compiled as a whole (its zero-argument
    # super() needs the class cell) against torch, which IS installed; weights set from seeded numpy arrays
    import inspect
    from copy import deepcopy
    import torch as th
    from torch import nn
    sa = _exec(_functions('core/utils.py', ['save_args']), {'inspect': inspect, 'deepcopy': deepcopy})
    with open(os.path.join(REF, 'core/agent/evo.py')) as f:
        tree = ast.parse(f.read())
    cm = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'ConvolutionModel')
    for n in ast.walk(cm):
        if isinstance(n, ast.FunctionDef):
            n.returns = None
            for a in n.args.args + n.args.kwonlyargs:
                a.annotation = None
    ns = _exec(ast.Module(body=[cm], type_ignores=[]), {'nn': nn, 'th': th, 'save_args': sa['save_args'],
                                                        'xavier_uniform': nn.init.xavier_uniform_})
    Model = ns['ConvolutionModel']
    for ci, (nobs, ksz, shape) in enumerate([(3, (3,), (9, 7)), (3, (3, 3), (8, 12)), (2, (5, 3), (11, 10)), (3, (1, 7), (13, 9))]):
        model = Model(num_obs_channels=nobs, num_act_channels=3, kernel_sizes=ksz, requires_grad=False)
        convs = [k for k in model.kernels if hasattr(k, 'weight')]
        ws = [rs.uniform(-0.7, 0.7, tuple(k.weight.shape)).astype(np.float32) for k in convs]
        for k, w in zip(convs, ws):
            k.weight.copy_(th.from_numpy(w))
        x = rs.rand(1, nobs, *shape).astype(np.float32)
        x[0, 0] = (x[0, 0] < 0.3)                          # an 'agents'-like 0/1 channel
        y = model(th.from_numpy(x.copy())).numpy()
        out[f'nca{ci}_in'] = x[0]
        out[f'nca{ci}_out'] = y[0]
        out[f'nca{ci}_nw'] = np.array(len(ws))
        for li, w in enumerate(ws):
            out[f'nca{ci}_w{li}'] = w

    # ---- DataInitializer builder chain (core/data_init.py:171-253), round 3: __init__ → with_noise → with_agents → build_numpy
    # and the static mask of build / build_agents (:241-253), the reference's own method bodies on seeded numpy draws.  (with_const
    # is left out: it uses np.float, gone from numpy 1.24 on — core/data_init.py:215 — and is a plain np.full.)  Appended LAST so
    # that the arrays above keep their values.
    names = ['__init__', '_mask', '_get_random', 'get_random', 'with_noise', 'with_agents', 'build_numpy', '_add_masked']
    bd = _exec(_functions('core/data_init.py', names, cls='DataInitializer'), {'np': np})

    class Builder:
        pass
    for n in names:
        setattr(Builder, n, staticmethod(bd[n]) if n == 'get_random' else bd[n])
    size = (7, 5)
    np.random.seed(4242)
    raw = np.random.random_sample((3,) + size)             # the three draws the chain below makes, in order
    np.random.seed(4242)
    b = Builder(size, ('agents', 'env_food', 'chem1'))
    b.with_noise('env_food', 0.1, 0.6).with_noise('chem1', -2, 3).with_agents(0.3)
    out['builder_raw'] = raw
    out['builder_numpy'] = b.build_numpy()
    mask = (np.arange(35).reshape(size) % 3 != 0).astype(np.float64)
    out['builder_mask'] = mask
    out['builder_masked'] = Builder(size, ('agents', 'env_food', 'chem1'), mask=mask).build_numpy() * 0 + b.build_numpy() * mask
    b._add_masked('env_food', np.full(size, 0.25))          # adds where the channel is > 0 (:206-209)
    out['builder_add_masked'] = b.build_numpy()[1]
    # BrownianAgent's chain (core/agent/static.py:40-50): a builder over (N,) with the alive mask
    N = 40
    alive = (np.arange(N) % 4 != 1).astype(np.float64)
    np.random.seed(777)
    raw_a = np.random.random_sample((3, N))
    np.random.seed(777)
    ba = Builder(N, ('dx', 'dy', 'deposit1'), mask=alive)
    ba.with_noise('dx', -0.01, 0.01).with_noise('dy', -0.01, 0.01).with_noise('deposit1', 0, 0.5)
    out['builder_action_raw'], out['builder_action_alive'] = raw_a, alive
    out['builder_action_out'] = ba.build_numpy() * ba._static_mask       # what build_agents wraps (:248-253)

    np.savez_compressed(OUT, **out)
    print('wrote', OUT, len(out), 'arrays')


if __name__ == '__main__':
    if not os.path.isdir(REF):
        sys.exit(f'{REF} not present: golden vectors can only be regenerated in the build container')
    main()
