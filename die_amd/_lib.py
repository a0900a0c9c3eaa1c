"""ctypes binding of libdie_hip.so (include/die_hip.h).  There is no CPU fallback: if the
library is missing or does not load, importing this module raises."""
import ctypes as C
import os

# torch first: it brings its own copy of the HIP runtime (soname libamdhip64.so.7).  Loaded
# before libdie_hip.so, that copy also satisfies our NEEDED entry, so kernels launched here
# and torch's allocations share one runtime; the other order puts two runtimes in the process.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DIE_AMD_LIB') or os.path.join(_HERE, 'libdie_hip.so')   # override: kernel experiments

DIE_OK = 0
DIE_F32, DIE_F16 = 0, 1
DIE_BOUNDARY_WRAP, DIE_BOUNDARY_LIMIT, DIE_BOUNDARY_NONE = 0, 1, 2
DIE_COST_LINEAR, DIE_COST_ZERO = 0, 1
DIE_AGENT_GRADIENT, DIE_AGENT_PHYSARUM = 0, 1
OWNER_EPOCH_SHIFT, OWNER_EPOCH_MAX, OWNER_SLOT_MASK = 27, 31, 0x07FFFFFF
DIFFUSE_MODES = {'wrap': 0, 'nearest': 1, 'reflect': 2, 'mirror': 3, 'constant': 4}
ABI_VERSION = 23


class Medium(C.Structure):
    _fields_ = [('W', C.c_int32), ('H', C.c_int32), ('dtype', C.c_int32), ('epoch', C.c_int32),
                ('owner', C.c_void_p), ('food', C.c_void_p), ('chem', C.c_void_p), ('chem_next', C.c_void_p),
                ('gW', C.c_int32), ('gH', C.c_int32), ('ox', C.c_int32), ('oy', C.c_int32),
                ('own_x0', C.c_int32), ('own_y0', C.c_int32), ('own_x1', C.c_int32), ('own_y1', C.c_int32),
                ('sense_mask', C.c_void_p)]


class Agents(C.Structure):
    _fields_ = [('N', C.c_int64), ('x', C.c_void_p), ('y', C.c_void_p), ('alive', C.c_void_p),
                ('agent_food', C.c_void_p), ('slot', C.c_void_p)]


class Action(C.Structure):
    _fields_ = [('N', C.c_int64), ('dx', C.c_void_p), ('dy', C.c_void_p), ('deposit', C.c_void_p)]


class Dynamics(C.Structure):
    _fields_ = [('rate_feed', C.c_float), ('rate_decay_chem', C.c_float), ('diffuse_sigma', C.c_float),
                ('boundary', C.c_int32), ('cost', C.c_int32), ('cost_w_deposit', C.c_float),
                ('cost_w_dist', C.c_float), ('food_infinite', C.c_int32), ('agents_die', C.c_int32),
                ('has_dead_slots', C.c_int32), ('diffuse_mode', C.c_int32), ('staged', C.c_int32)]


class GradientAgent(C.Structure):
    _fields_ = [('kind', C.c_int32), ('normalized_grad', C.c_int32), ('scale', C.c_float), ('deposit', C.c_float),
                ('inertia', C.c_float), ('sense_offset', C.c_float), ('noise_scale', C.c_float),
                ('grad_clip', C.c_float), ('turn_radians', C.c_double), ('sense_radians', C.c_double),
                ('turn_tolerance', C.c_double), ('heading_hi', C.c_void_p), ('heading_lo', C.c_void_p),
                ('prev_gx', C.c_void_p), ('prev_gy', C.c_void_p), ('turn_sign', C.c_void_p),
                ('seed', C.c_uint64), ('step', C.c_uint32), ('reserved2', C.c_uint32), ('step_base', C.c_void_p)]


class ConvPlane(C.Structure):
    _fields_ = [('data', C.c_void_p), ('kind', C.c_int32), ('reserved', C.c_int32)]


DIE_PLANE_F32, DIE_PLANE_F16, DIE_PLANE_AGENTS = 0, 1, 2
PAD_MODES = {'circular': 0, 'zeros': 1, 'reflect': 2, 'replicate': 3}        # torch.nn.Conv2d padding_mode → die_pad_mode
DIE_FIELD_CONST, DIE_FIELD_NOISE, DIE_FIELD_AGENTS, DIE_FIELD_PERLIN = 0, 1, 2, 3


class PicLayout(C.Structure):
    _fields_ = [('x', C.c_void_p), ('y', C.c_void_p), ('agent_food', C.c_void_p), ('slot', C.c_void_p), ('heading_hi', C.c_void_p),
                ('heading_lo', C.c_void_p),
                ('off', C.c_void_p), ('n', C.c_void_p), ('s', C.c_void_p), ('inc', C.c_void_p)]


class Pic(C.Structure):
    _fields_ = [('tile_xs', C.c_int32), ('tile_ys', C.c_int32), ('N', C.c_int64), ('layout', PicLayout * 2),
                ('dep', C.c_void_p), ('dep_plane', C.c_void_p), ('part_gain', C.c_void_p), ('error', C.c_void_p),
                ('k1_threads', C.c_int32), ('stages', C.c_int32),
                ('rim', C.c_void_p), ('rim_code', C.c_void_p), ('rim_cnt', C.c_void_p), ('status_out', C.c_void_p),
                ('turn_bits', C.c_void_p), ('turn_slots', C.c_int64), ('turn_ready', C.c_int32), ('order_ready', C.c_int32),
                ('order', C.c_void_p), ('sub_mode', C.c_int32), ('sub_tx0', C.c_int32), ('sub_ty0', C.c_int32), ('sub_ntx', C.c_int32),
                ('sub_nty', C.c_int32), ('halo_fresh', C.c_int32), ('n_alive', C.c_int64), ('occ', C.c_void_p),
                ('prev_grad', (C.c_void_p * 2) * 2)]


class PicSide(C.Structure):          # die_pic_side: one neighbour of the ghost refresh by tiles
    _fields_ = [('tx0', C.c_int32), ('ty0', C.c_int32), ('ntx', C.c_int32), ('nty', C.c_int32), ('hx0', C.c_int32), ('hy0', C.c_int32),
                ('cap', C.c_int64), ('send_counts', C.c_void_p), ('send_rec', C.c_void_p), ('recv_counts', C.c_void_p), ('recv_rec', C.c_void_p)]


PIC_GHOST_SUMMARY_WORDS = 19
PIC_GHOST_FLAGS = {1: 'a tile holds another number of agents than its per-tile words say', 2: 'a band holds more agents than a message',
                   4: 'a received count is impossible', 8: 'the agents do not fit the local arrays'}


class Batch(C.Structure):
    _fields_ = [('replicas', C.c_int32), ('reserved', C.c_int32), ('plane_stride', C.c_int64), ('agent_stride', C.c_int64),
                ('seed_stride', C.c_uint64), ('n', C.c_int64 * 64)]


class Rect(C.Structure):
    _fields_ = [('plane', C.c_void_p), ('pitch', C.c_int32), ('r0', C.c_int32), ('r1', C.c_int32), ('c0', C.c_int32),
                ('c1', C.c_int32), ('elem_bytes', C.c_int32), ('buf_offset', C.c_int64)]


class FoodSpec(C.Structure):
    _fields_ = [('n_waves', C.c_int32), ('scale', C.c_float), ('perlin_octaves', C.c_int32), ('threshold', C.c_float), ('fx', C.c_double * 8), ('fy', C.c_double * 8),
                ('phase', C.c_double * 8), ('amp', C.c_double * 8)]


_P = C.POINTER
_SIGNATURES = {
    'die_abi_version': (C.c_int, []),
    'die_last_error': (C.c_char_p, []),
    'die_workspace_bytes': (C.c_int64, [C.c_int32, C.c_int32, C.c_int64]),
    'die_gradient_forward': (C.c_int, [_P(Medium), _P(Agents), _P(GradientAgent), _P(Action), C.c_void_p]),
    'die_brownian_forward': (C.c_int, [_P(Agents), C.c_float, C.c_float, C.c_uint64, C.c_uint32, _P(Action),
                                       C.c_void_p]),
    'die_const_forward': (C.c_int, [C.c_int64, C.c_float, C.c_float, C.c_float, _P(Action), C.c_void_p]),
    'die_env_step': (C.c_int, [_P(Medium), _P(Agents), _P(Action), _P(Dynamics), C.c_void_p, C.c_void_p, C.c_int64,
                               C.c_void_p]),
    'die_forward_env_step': (C.c_int, [_P(Medium), _P(Agents), _P(GradientAgent), _P(Action), _P(Dynamics), C.c_void_p,
                                       C.c_void_p, C.c_int64, C.c_void_p]),
    'die_batch_workspace_bytes': (C.c_int64, [C.c_int32]),
    'die_forward_env_step_batch': (C.c_int, [_P(Medium), _P(Agents), _P(GradientAgent), _P(Action), _P(Dynamics), _P(Batch), C.c_void_p,
                                             C.c_void_p, C.c_int64, C.c_void_p]),
    'die_forward_move_claim_tile': (C.c_int, [_P(Medium), _P(Agents), _P(GradientAgent), _P(Action), _P(Dynamics), C.c_void_p,
                                              C.c_int64, C.c_void_p]),
    'die_env_step_finish': (C.c_int, [_P(Medium), _P(Agents), _P(Action), _P(Dynamics), C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_void_p]),
    'die_forward_move_claim': (C.c_int, [_P(Medium), _P(Agents), _P(GradientAgent), _P(Action), _P(Dynamics), C.c_void_p,
                                         C.c_int64, C.c_void_p]),
    'die_agent_move_claim': (C.c_int, [_P(Medium), _P(Agents), _P(Action), _P(Dynamics), C.c_void_p, C.c_int64,
                                       C.c_void_p]),
    'die_agent_move': (C.c_int, [_P(Medium), _P(Agents), _P(Action), _P(Dynamics), C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                 C.c_void_p]),
    'die_forward_move': (C.c_int, [_P(Medium), _P(Agents), _P(GradientAgent), _P(Action), _P(Dynamics), C.c_int32, C.c_int32,
                                   C.c_int32, C.c_void_p, C.c_void_p]),
    'die_step_reduce_ex': (C.c_int, [_P(Agents), C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p]),
    'die_agents_lifecycle': (C.c_int, [_P(Agents), C.c_void_p]),
    'die_gradient_render': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    'die_agent_claim_feed': (C.c_int, [_P(Medium), _P(Agents), _P(Action), _P(Dynamics), C.c_void_p, C.c_int64,
                                       C.c_void_p]),
    'die_diffuse_decay_tile': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                         C.c_void_p]),
    'die_agent_resolve': (C.c_int, [_P(Medium), _P(Agents), _P(Action), _P(Dynamics), C.c_void_p, C.c_int64,
                                    C.c_void_p]),
    'die_step_reduce': (C.c_int, [_P(Agents), _P(Dynamics), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'die_medium_deposit_feed_diffuse': (C.c_int, [_P(Medium), _P(Dynamics), C.c_void_p]),
    'die_medium_deposit_feed_diffuse_tile': (C.c_int, [_P(Medium), _P(Dynamics), C.c_int32, C.c_void_p]),
    'die_tile_sweep_reduce': (C.c_int, [_P(Medium), _P(Agents), _P(Dynamics), C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_void_p]),
    'die_agent_dead_slots': (C.c_int, [_P(Medium), _P(Agents), _P(Action), _P(Dynamics), C.c_void_p, C.c_int64,
                                       C.c_void_p]),
    'die_diffuse_decay_mode': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32,
                                         C.c_void_p]),
    'die_diffuse_decay': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                    C.c_void_p]),
    'die_init_medium': (C.c_int, [_P(Medium), C.c_double, C.c_uint64, _P(FoodSpec), C.c_void_p]),
    'die_init_agents': (C.c_int, [_P(Medium), _P(Agents), C.c_uint64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'die_init_heading': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_uint64, C.c_void_p]),
    'die_field_fill': (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_uint64, C.c_uint32,
                                 C.c_uint32, C.c_void_p]),
    'die_medium_from_fields': (C.c_int, [_P(Medium), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    'die_food_flow_wave': (C.c_int, [_P(Medium), C.c_double, C.c_double, C.c_double, C.c_void_p]),
    'die_food_flow_perlin': (C.c_int, [_P(Medium), C.c_double, C.c_int32, C.c_double, C.c_double, C.c_uint64, C.c_void_p]),
    'die_sense_mask': (C.c_int, [_P(Medium), C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    'die_render_frames': (C.c_int, [_P(Medium), C.c_void_p, C.c_float, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    'die_rects_pack': (C.c_int, [_P(Rect), C.c_int32, C.c_void_p, C.c_void_p]),
    'die_pic_two_launch': (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32]),
    'die_pic_step_bound': (C.c_float, [C.c_float, C.c_float]),
    'die_stream_copy': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'die_host_device_pointer': (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    'die_rects_unpack': (C.c_int, [_P(Rect), C.c_int32, C.c_void_p, C.c_void_p]),
    'die_rects_unpack_max': (C.c_int, [_P(Rect), C.c_int32, C.c_void_p, C.c_void_p]),
    'die_records_gather': (C.c_int, [_P(C.c_void_p), _P(C.c_int32), C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'die_records_scatter': (C.c_int, [_P(C.c_void_p), _P(C.c_int32), C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'die_ghost_workspace_bytes': (C.c_int64, [C.c_int64]),
    'die_ghost_plan': (C.c_int, [_P(Medium), _P(Agents), C.c_int32, _P(C.c_int8), _P(C.c_void_p), _P(C.c_int64), C.c_void_p,
                                 C.c_void_p, C.c_int64, C.c_void_p]),
    'die_records_gather_dev': (C.c_int, [_P(C.c_void_p), _P(C.c_int32), C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    'die_records_scatter_at': (C.c_int, [_P(C.c_void_p), _P(C.c_int32), C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                         C.c_void_p]),
    'die_ghost_pack': (C.c_int, [_P(C.c_void_p), _P(C.c_int32), C.c_int32, C.c_int32, _P(C.c_void_p), C.c_void_p, _P(C.c_int64),
                                 _P(C.c_int64), _P(C.c_int64), C.c_void_p, C.c_void_p]),
    'die_ghost_apply': (C.c_int, [_P(C.c_void_p), _P(C.c_int32), C.c_int32, C.c_int32, C.c_void_p, _P(C.c_int64), _P(C.c_int64),
                                  _P(C.c_int64), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    'die_pic_tiles': (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'die_pic_rim_cap': (C.c_int64, [C.c_int32, C.c_int32]),
    'die_pic_bin': (C.c_int, [_P(Medium), _P(Agents), C.c_void_p, C.c_void_p, _P(Pic), C.c_int32, C.c_void_p]),
    'die_pic_bin_momentum': (C.c_int, [_P(Medium), _P(Agents), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _P(Pic), C.c_int32, C.c_void_p]),
    'die_pic_forward_env_step': (C.c_int, [_P(Medium), _P(Pic), C.c_int32, _P(GradientAgent), _P(Action), _P(Dynamics), C.c_void_p,
                                           C.c_void_p]),
    'die_pic_run': (C.c_int, [_P(Medium), _P(Pic), C.c_int32, _P(GradientAgent), _P(Dynamics), C.c_int32, C.c_void_p, C.c_void_p]),
    'die_pic_run_completed': (C.c_int32, []),
    'die_agents_mark_owner': (C.c_int, [_P(Medium), _P(Agents), C.c_void_p]),
    'die_pic_action_physarum': (C.c_int, [_P(Pic), C.c_int32, _P(GradientAgent), _P(Action), C.c_void_p]),
    'die_pic_ghost_pack': (C.c_int, [_P(Medium), _P(Pic), C.c_int32, C.c_int32, _P(PicSide), C.c_void_p, C.c_void_p]),
    'die_pic_ghost_merge': (C.c_int, [_P(Medium), _P(Pic), C.c_int32, C.c_int32, _P(PicSide), C.c_int64, C.c_void_p, C.c_void_p]),
    'die_pic_ghost_merge_phase': (C.c_int, [_P(Medium), _P(Pic), C.c_int32, C.c_int32, _P(PicSide), C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]),
    'die_pic_ghost_inplace': (C.c_int, [_P(Medium), _P(Pic), C.c_int32, C.c_int32, _P(PicSide), C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    'die_conv2d_circular': (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P(ConvPlane), C.c_int32, C.c_int32, _P(C.c_void_p), C.c_int32,
                                      C.c_void_p, C.c_int32, C.c_void_p]),
    'die_conv2d': (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P(ConvPlane), C.c_int32, C.c_int32, _P(C.c_void_p), C.c_int32,
                             C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    'die_gather_scale': (C.c_int, [_P(Medium), _P(Agents), _P(C.c_void_p), _P(C.c_float), _P(Action), C.c_void_p]),
    'die_sort_workspace_bytes': (C.c_int64, [C.c_int32, C.c_int32, C.c_int64]),
    'die_agents_sort': (C.c_int, [_P(Medium), _P(Agents), _P(Agents), C.c_int32, _P(C.c_void_p), _P(C.c_void_p), C.c_void_p,
                                  C.c_int64, C.c_void_p]),
}
EXPORTS = tuple(_SIGNATURES)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f'{LIB_PATH} is missing: build it with `python die_amd/build.py` '
                          '(hipcc, --offload-arch=gfx950). die_amd has no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError here = the .so does not match include/die_hip.h
        fn.restype = res
        fn.argtypes = args
    if lib.die_abi_version() != ABI_VERSION:
        raise ImportError(f'libdie_hip.so ABI {lib.die_abi_version()} != binding ABI {ABI_VERSION}: rebuild')
    return lib


lib = _load()


class DieError(RuntimeError):
    pass


def check(rc: int, what: str = ''):
    if rc != DIE_OK:
        msg = lib.die_last_error().decode(errors='replace')
        if rc == -3:
            raise NotImplementedError(f'{what}: {msg}')
        raise DieError(f'{what}: status {rc}: {msg}')
