"""CPU-only checks of the boundary and the host logic: the C-ABI library loads and exports
every symbol include/die_hip.h declares, struct layouts agree, host helpers agree with the
oracle.  No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built_lib():
    import importlib.util
    spec = importlib.util.spec_from_file_location('die_build', os.path.join(ROOT, 'die_amd', 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build()


def _declared_functions():
    text = open(os.path.join(ROOT, 'include', 'die_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(die_[a-z_0-9]+)\s*\(', text)))


def test_header_symbols_exported(built_lib):
    lib = C.CDLL(built_lib)
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/die_hip.h but not exported'
    from die_amd import _lib
    assert sorted(_lib.EXPORTS) == names
    assert _lib.lib.die_abi_version() == _lib.ABI_VERSION


def test_struct_layouts_match_header(built_lib):
    from die_amd import _lib
    # sizes follow from the field lists in include/die_hip.h under the x86-64 SysV ABI
    assert C.sizeof(_lib.Medium) == 4 * 4 + 4 * 8 + 4 * 4 + 4 * 4 + 8
    assert C.sizeof(_lib.Agents) == 8 + 5 * 8
    assert C.sizeof(_lib.Action) == 8 + 3 * 8
    assert C.sizeof(_lib.Dynamics) == 12 * 4
    assert C.sizeof(_lib.GradientAgent) == 8 * 4 + 3 * 8 + 5 * 8 + 8 + 4 + 4 + 8
    assert C.sizeof(_lib.FoodSpec) == 16 + 4 * 8 * 8
    assert C.sizeof(_lib.PicLayout) == 10 * 8
    assert C.sizeof(_lib.PicSide) == 6 * 4 + 8 + 4 * 8
    # die_pic: tile shape 2 x i32, N i64, two layouts, dep / dep_plane / part_gain / error, k1_threads + stages, rim / rim_code /
    # rim_cnt / status_out, turn_bits / turn_slots / turn_ready + reserved / reserved pointer / sub_mode + rectangle + reserved / n_alive / occ /
    # prev_grad[2][2]
    assert C.sizeof(_lib.Pic) == 8 + 8 + 2 * 10 * 8 + 4 * 8 + 8 + 4 * 8 + 8 + 8 + 8 + 8 + 6 * 4 + 8 + 8 + 4 * 8
    assert C.sizeof(_lib.Batch) == 8 + 8 + 8 + 8 + 64 * 8


def test_tile_binned_geometry_helpers_without_gpu(built_lib):
    """die_pic_tiles / die_pic_rim_cap are pure host arithmetic: compiled tile shapes, entries per rim list (nine lists of
    that many code bytes are one 4-byte load per thread of the field kernel: 4 x 512 / 9 and 4 x 256 / 9, rounded down to
    whole words)."""
    from die_amd import _lib
    from die_amd.pic import pick_tile
    L = _lib.lib
    assert L.die_pic_tiles(4096, 4096, 6, 6) == 4096 and L.die_pic_tiles(192, 256, 4, 5) == 12 * 8
    assert L.die_pic_tiles(4096, 4096, 3, 3) == -1 and L.die_pic_tiles(0, 64, 6, 6) == -1
    assert L.die_pic_rim_cap(6, 6) == 224 and L.die_pic_rim_cap(5, 7) == 224 and L.die_pic_rim_cap(5, 6) == 112 and L.die_pic_rim_cap(4, 5) == 112
    assert L.die_pic_rim_cap(7, 7) == -1
    for xs, ys in ((6, 6), (5, 7), (5, 6), (4, 5)):
        assert L.die_pic_rim_cap(xs, ys) % 4 == 0 and 9 * L.die_pic_rim_cap(xs, ys) <= 4 * (512 if (1 << xs) * (1 << ys) >= 4096 else 256)
    assert pick_tile(4096, 4096, 1.53) == (6, 6) and pick_tile(128, 96, 1.53) == (4, 5) and pick_tile(100, 100, 1.53) is None
    assert pick_tile(192, 192, 40.0) == (6, 6) and pick_tile(192, 192, 63.5) is None


def test_argument_validation_without_gpu(built_lib):
    """Bad arguments are rejected on the host before anything touches a device."""
    from die_amd import _lib
    assert _lib.lib.die_workspace_bytes(0, 4, 4) == -1
    assert _lib.lib.die_workspace_bytes(4096, 4096, 4096 * 4096) >= 4096 * 4096 * 4
    rc = _lib.lib.die_diffuse_decay(None, None, 8, 8, 0, 0.5, 0.1, None)
    assert rc == -1 and b'non-null' in _lib.lib.die_last_error()
    m = _lib.Medium(1, 1, 0, 1, None, None, None, None, 0, 0, 0, 0, 0, 0, 0, 0, None)
    a = _lib.Agents(0, None, None, None, None, None)
    g = _lib.GradientAgent()
    u = _lib.Action(0, None, None, None)
    assert _lib.lib.die_gradient_forward(C.byref(m), C.byref(a), C.byref(g), C.byref(u), None) == -1
    with pytest.raises(_lib.DieError):
        _lib.check(-1, 'x')
    # the ghost refresh by tiles: planes that are no tile of a decomposed world / owned cells off the tile borders / a band
    # outside the interior are refused before any launch
    p = _lib.Pic(6, 6, 1, (_lib.PicLayout * 2)(), None, None, None, None, 0, 0, None, None, None, None)
    summary = (C.c_int64 * _lib.PIC_GHOST_SUMMARY_WORDS)()
    plain = _lib.Medium(256, 256, 0, 1, None, None, None, None, 0, 0, 0, 0, 0, 0, 0, 0, None)
    assert _lib.lib.die_pic_ghost_pack(C.byref(plain), C.byref(p), 0, 0, None, summary, None) == -1
    assert b'decomposed' in _lib.lib.die_last_error()
    off_grid = _lib.Medium(256, 256, 0, 1, None, None, None, None, 512, 512, 0, 0, 48, 64, 208, 192, None)
    assert _lib.lib.die_pic_ghost_merge(C.byref(off_grid), C.byref(p), 0, 0, None, 16, summary, None) == -1
    assert b'whole 64x64 tiles' in _lib.lib.die_last_error()


def test_q32_roundtrip_and_cell_formula():
    from die_amd.device_array import from_q32, to_q32
    from oracle import cpu_ref as R
    v = np.array([0., 0.25, 1 / 3, 0.999999999, 1.0])
    q = to_q32(v)
    assert q.dtype == np.uint32 and q[0] == 0 and q[1] == 2 ** 30 and q[-1] == 2 ** 32 - 1
    assert np.abs(from_q32(q) - v).max() <= 2.0 ** -32
    n = 4096
    cells = ((q.astype(np.uint64) * np.uint64(n - 1) + np.uint64(2 ** 31)) >> np.uint64(32)).astype(np.int64)
    assert (cells == R.cell(from_q32(q), n)).all()


def test_host_food_spec_matches_oracle():
    from die_amd.data_init import food_spec_from_seed
    from oracle import cpu_ref as R
    for seed in (0, 1234, 2 ** 40 + 17):
        s = food_spec_from_seed(seed)
        o = R.FoodSpec.from_seed(seed)
        assert s.n_waves == len(o.fx)
        for i in range(s.n_waves):
            assert s.fx[i] == o.fx[i] and s.fy[i] == o.fy[i]
            assert np.isclose(s.phase[i], o.phase[i], rtol=0, atol=1e-15) and np.isclose(s.amp[i], o.amp[i], rtol=1e-15)


def test_env_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    import die_amd
    with pytest.raises(RuntimeError, match='no CPU path'):
        die_amd.Env((8, 8))


def test_agent_save_load_roundtrip(tmp_path):
    import die_amd
    a = die_amd.PhysarumAgent(max_agents=64, scale=0.006, turn_angle=30, sense_offset=0.04, seed=5)
    f = tmp_path / 'agent.json'
    a.save(f)
    b = die_amd.PhysarumAgent.load(f)
    assert b.init_params == a.init_params and b.init_params['sense_offset'] == 0.04
    c = die_amd.BrownianAgent(move_scale=0.02)
    assert c.init_params == {'move_scale': 0.02, 'deposit_scale': 0.5, 'seed': None}


def test_two_launch_rule_is_the_librarys(built_lib):
    """die_pic_two_launch: the ONE statement of which form the tile-binned step takes (die_amd/pic.py restated it);
    pinned here against the rule in words — an agent changes cell by at most floor(reach) + 1 per axis, + 1 across the world's
    seam, and must stay R cells clear of the far border of the tile it walks onto."""
    import numpy as np
    from die_amd import _lib
    L = _lib.lib
    for worldmax, (xs, ys), scale, sigma, mode in ((4096, (6, 6), 1.53 / 4095, 0.5, 0), (4096, (6, 6), 0.006, 0.5, 0), (128, (4, 5), 1.53 / 127, 0.5, 0),
                                                    (128, (4, 5), 11.5 / 127, 0.5, 0), (4096, (6, 6), 1.53 / 4095, 0.5, 1), (4096, (6, 6), 1.53 / 4095, 1.2, 0),
                                                    (4096, (6, 6), 59.5 / 4095, 0.5, 0), (256, (5, 6), -9.9 / 255, 0.8, 0)):
        reach = float(np.float32(abs(scale)) * np.float32(worldmax - 1))
        R = int(4.0 * float(np.float32(sigma)) + 0.5)
        want = mode == 0 and 1 <= R <= 4 and int(reach) + 2 + R <= min(1 << xs, 1 << ys)
        assert L.die_pic_two_launch(worldmax, xs, ys, scale, sigma, mode) == int(want), (worldmax, xs, ys, scale, sigma, mode)
    assert L.die_pic_two_launch(4096, 3, 3, 0.001, 0.5, 0) == -1


def test_step_bound_of_a_gradient_agent_with_momentum(built_lib):
    """die_pic_step_bound: per axis, |u'| <= (1 - i)·1 + i·|u| + ns·n_max for a normalised gradient (core/agent/gradient.py:82-91),
    n_max = the largest 0.4-sigma Box-Muller normal the library draws from a 32-bit uniform: a bound that holds for the initial
    _prev_grad (the same normals) and is a fixed point of the recurrence."""
    import math
    from die_amd import _lib
    B = _lib.lib.die_pic_step_bound
    n_max = 0.4 * math.sqrt(-2.0 * math.log(2.0 ** -32))
    assert 2.66 < n_max < 2.67
    assert B(0.0, 0.0) == 1.0
    assert abs(B(0.0, 0.05) - (1 + 0.05 * 2.67)) < 1e-6
    assert abs(B(0.9, 0.025) - 2.67) < 1e-6                        # the reference's defaults: the initial noise dominates
    assert abs(B(0.9, 0.5) - (1 + 0.5 * 2.67 / 0.1)) < 1e-4
    for i, ns in ((0.9, 0.025), (0.5, 0.2), (0.99, 0.01), (0.3, 0.0)):
        b = B(i, ns)
        assert b >= n_max and (1 - i) + i * b + ns * n_max <= b * (1 + 1e-6)
    assert math.isinf(B(1.0, 0.0)) and math.isinf(B(1.5, 0.1))
