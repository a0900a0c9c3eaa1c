"""TEST INFRASTRUCTURE ONLY — the parity oracle.  Nothing under die_amd/ may import this.

A float64 numpy/scipy CPU restatement of the reference's grid-update hot path
(gkirgizov/die): `Env.step` field dynamics, `PhysarumAgent` / `GradientAgent` /
`BrownianAgent` / `ConstAgent` `.forward`, and the `DataInitializer` allocation.
Each function cites the reference lines it follows (paths relative to the
reference checkout).  xarray label lookups are replaced by the index arithmetic
they resolve to; every random draw is an explicit argument (see oracle/rng.py).

PARITY STATUS: *partially pinned*.  The reference has no test, golden vector or
fixture on this path (test/unit/agent.py covers core/agent/evo.py only) and cannot be
imported here (xarray / skimage / gymnasium / perlin_noise / evotorch are absent —
ordinary ModuleNotFoundError).  What pins this file:
  * `cell()`                 == pandas `Index.get_indexer(method='nearest')`, the call
                                xarray's `.sel(method='nearest')` resolves to (tests/test_oracle_pins.py)
  * `diffuse_decay()`        calls scipy.ndimage.gaussian_filter — the function
                                skimage.filters.gaussian wraps — and is cross-checked with explicit weights
  * pure-numpy helper bodies (`renormalize_radians`, `discretize`, `polar2xy`, `xy2polar`,
    `PhysarumAgent._choose_turn/_discrete_turn`, `GradientAgent._process_momentum`,
    `DataInitializer._mask/get_random`, `Env._agent_move_handle_boundary`,
    `WaveSequence.__getitem__`) were executed from the reference's own files in the build
    container and their outputs committed as tests/golden/ref_helpers.npz
    (script: tests/golden/make_ref_helper_vectors.py); the oracle is checked against them.
The xarray-dependent glue of `Env.step` / `forward` itself stays unpinned by reference
output; it is pinned by analytic known-answer tests only (tests/test_oracle_kat.py).
"""
from dataclasses import dataclass, field as dc_field
from typing import Callable, Optional, Tuple

import numpy as np
import scipy.ndimage

from . import rng as orng

# core/base_types.py:31-36
MEDIUM_CHANNELS = ('agents', 'env_food', 'chem1')
AGENT_CHANNELS = ('x', 'y', 'alive', 'agent_food')
ACTION_CHANNELS = ('dx', 'dy', 'deposit1')
M_AGENTS, M_FOOD, M_CHEM = 0, 1, 2
A_X, A_Y, A_ALIVE, A_FOOD = 0, 1, 2, 3
U_DX, U_DY, U_DEP = 0, 1, 2


# --------------------------------------------------------------------------------------
# helpers: core/utils.py:154-183
# --------------------------------------------------------------------------------------
def polar2xy(r, theta):
    """core/utils.py:154-165 (r·e^{iθ} → real, imag)."""
    z = r * np.exp(1j * np.asarray(theta))
    return np.real(z), np.imag(z)


def xy2polar(x, y):
    """core/utils.py:158-169."""
    z = np.asarray(x) + 1j * np.asarray(y)
    return np.abs(z), np.angle(z)


def get_radians(coords):
    """core/utils.py:172-175."""
    return xy2polar(coords[0], coords[1])[1]


def renormalize_radians(rads):
    """core/utils.py:178-180: into (-pi, pi]."""
    return (rads - np.pi) % (-2 * np.pi) + np.pi


def discretize(value, step):
    """core/utils.py:183-184."""
    return (value // step) * step


def cell(v, n: int) -> np.ndarray:
    """Nearest grid index of coordinate v on labels linspace(0, 1, n): what
    `field.sel(x=v, method='nearest')` resolves to (core/utils.py:39-54 → pandas
    `get_indexer(method='nearest')`): ties go to the larger index, out-of-range clamps."""
    return np.clip(np.floor(np.asarray(v, dtype=np.float64) * (n - 1) + 0.5), 0, n - 1).astype(np.int64)


def gaussian_weights(sigma: float, truncate: float = 4.0) -> np.ndarray:
    """scipy.ndimage._gaussian_kernel1d (order 0): radius int(truncate*sigma + .5)."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    w = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return w / w.sum()


# --------------------------------------------------------------------------------------
# Env: core/env.py
# --------------------------------------------------------------------------------------
def linear_action_cost(action: np.ndarray, weights=(0.02, 0.01)) -> np.ndarray:
    """core/env.py:29-35."""
    dist = np.linalg.norm(action[[U_DX, U_DY]], axis=0)
    return weights[0] * np.abs(action[U_DEP]) + weights[1] * dist


def zero_cost(action: np.ndarray) -> np.ndarray:
    """core/env.py:38-39."""
    return np.zeros(action.shape[1:])


@dataclass
class RefDynamics:
    """core/env.py:42-61."""
    op_action_cost: Callable = linear_action_cost
    op_food_flow: Callable = dc_field(default=lambda x: x)
    rate_feed: float = 0.1
    rate_decay_chem: float = 0.1
    boundary: str = 'wrap'           # BoundaryCondition value: 'wrap' | 'limit' | anything else = pass-through
    diffuse_mode: str = 'wrap'
    diffuse_sigma: float = .5
    apply_sense_mask: bool = False
    strict_cost: bool = True
    food_infinite: bool = False
    agents_die: bool = False
    agents_born: bool = False
    init_agent_ratio: float = 0.1
    compat: str = 'intended'         # 'reference': keep the AgentIndexer's stale array after _agent_lifecycle (see RefEnv)


def move_handle_boundary(coords: np.ndarray, boundary: str) -> np.ndarray:
    """core/env.py:152-161."""
    if boundary == 'wrap':
        return coords % 1.
    if boundary == 'limit':
        return coords.clip(0., 1.)
    return coords


def diffuse_decay(chem: np.ndarray, sigma: float, decay: float, mode: str = 'wrap') -> np.ndarray:
    """core/env.py:136-145: skimage.filters.gaussian(preserve_range=True) is
    scipy.ndimage.gaussian_filter(truncate=4.0) on a float image; then ×(1−decay)."""
    out = scipy.ndimage.gaussian_filter(np.asarray(chem, dtype=np.float64), sigma=sigma, mode=mode, truncate=4.0)
    out *= (1. - decay)
    return out


def diffuse_decay_explicit(chem: np.ndarray, sigma: float, decay: float) -> np.ndarray:
    """Same as diffuse_decay(mode='wrap') with the separable weights written out:
    axis 0 then axis 1, periodic with period W / H."""
    w = gaussian_weights(sigma)
    r = len(w) // 2
    out = np.asarray(chem, dtype=np.float64)
    for axis in (0, 1):
        acc = np.zeros_like(out)
        for k in range(-r, r + 1):
            acc += w[k + r] * np.roll(out, -k, axis=axis)
        out = acc
    return out * (1. - decay)


class RefEnv:
    """core/env.py:64-311 on plain arrays: medium (3, W, H) f64, agents (4, N) f64."""

    def __init__(self, medium: np.ndarray, agents: np.ndarray, dynamics: Optional[RefDynamics] = None):
        self.medium = np.array(medium, dtype=np.float64)
        self.agents = np.array(agents, dtype=np.float64)
        self.dynamics = dynamics or RefDynamics()
        assert self.medium.ndim == 3 and self.medium.shape[0] == 3
        assert self.agents.ndim == 2 and self.agents.shape[0] == 4
        self.last_gained = None
        # AgentIndexer(field_size, self.agents) of Env._init_data (core/env.py:83) holds THIS array object
        # (core/utils.py:22); see agent_lifecycle
        self._idx_agents = self.agents

    @property
    def field_size(self) -> Tuple[int, int]:
        return self.medium.shape[1], self.medium.shape[2]

    # core/utils.py:26-54
    def cells_of(self, xy: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        W, H = self.field_size
        return cell(xy[0], W), cell(xy[1], H)

    def alive_index(self) -> np.ndarray:
        """core/utils.py:67-75 (the indexer's array)."""
        return (self._idx_agents[A_ALIVE] > 0).nonzero()[0]

    @property
    def num_alive(self) -> int:
        return int(self.alive_index().shape[0])

    def agent_move(self, action: np.ndarray):
        """core/env.py:163-172 — every slot, dead ones too."""
        self.agents[[A_X, A_Y]] = move_handle_boundary(self.agents[[A_X, A_Y]] + action[[U_DX, U_DY]],
                                                       self.dynamics.boundary)

    def agent_deposit_and_layout(self, action: np.ndarray):
        """core/env.py:204-215.  `.loc[cells] += deposit` is get–add–set with fancy
        indices: on a shared cell the LAST alive slot in index order wins."""
        idx = self.alive_index()
        ix, iy = self.cells_of(self._idx_agents[[A_X, A_Y]][:, idx])
        deposit = action[U_DEP, idx]
        chem = self.medium[M_CHEM]
        chem[ix, iy] = chem[ix, iy] + deposit
        self.medium[M_AGENTS] = 0
        self.medium[M_AGENTS][ix, iy] = 1.

    def agent_feed(self, action: np.ndarray) -> np.ndarray:
        """core/env.py:220-243."""
        d = self.dynamics
        food = self.medium[M_FOOD]
        consumed_field = d.rate_feed * food * (self.medium[M_AGENTS] > 0)
        ix, iy = self.cells_of(self._idx_agents[[A_X, A_Y]])     # all N slots (only_alive=False), the indexer's array
        consumed = consumed_field[ix, iy]
        if not d.food_infinite:
            self.medium[M_FOOD] = food - consumed_field
        burned = d.op_action_cost(action)
        gained = consumed - burned
        self.agents[A_FOOD] += gained
        return gained

    def agent_lifecycle(self):
        """core/env.py:245-261: `self.agents = self.agents.where(have_food, 0)` REBINDS the attribute to a new array
        while the AgentIndexer keeps the one it was built with (core/utils.py:22).  compat='reference' reproduces
        that: from now on deposits, layout, feeding look-ups and num_alive read the old array (positions after this
        step's move, alive flags never cleared), while moves, agent_food and the zeroing go to the new one.
        compat='intended': the indexer follows the new array (what the code evidently means)."""
        if self.dynamics.agents_die:
            have_food = self.agents[A_FOOD] > 1e-4
            self.agents = np.where(have_food, self.agents, 0.)
            if self.dynamics.compat != 'reference':
                self._idx_agents = self.agents

    def medium_resource_dynamics(self):
        """core/env.py:147-150."""
        self.medium[M_FOOD] = self.dynamics.op_food_flow(self.medium[M_FOOD])

    def medium_diffuse_decay(self):
        d = self.dynamics
        self.medium[M_CHEM] = diffuse_decay(self.medium[M_CHEM], d.diffuse_sigma, d.rate_decay_chem, d.diffuse_mode)

    def sense_mask(self) -> np.ndarray:
        """core/env.py:276-290 (skimage default mode for gaussian is 'nearest')."""
        a = self.medium[M_AGENTS]
        if self.dynamics.apply_sense_mask:
            return np.ceil(scipy.ndimage.gaussian_filter(a, sigma=2.0, mode='nearest', truncate=4.0).round(3))
        return np.ones_like(a)

    @property
    def obs(self):
        """core/env.py:292-298."""
        return self.agents, np.where(self.sense_mask().astype(bool), self.medium, 0.)

    def step(self, action: np.ndarray):
        """core/env.py:101-131."""
        action = np.asarray(action, dtype=np.float64)
        self.agent_move(action)
        self.agent_deposit_and_layout(action)
        gained = self.agent_feed(action)
        self.agent_lifecycle()
        self.medium_resource_dynamics()
        self.medium_diffuse_decay()
        self.last_gained = gained
        num_agents = self.num_alive
        reward = float(gained.sum())
        mean_gain = reward / num_agents if num_agents > 0 else 0.
        info = {'num_agents': num_agents, 'reward': np.round(reward, 3), 'mean_reward': np.round(mean_gain, 5)}
        return self.obs, reward, num_agents == 0, False, info


# --------------------------------------------------------------------------------------
# Agents: core/agent/gradient.py, core/agent/static.py
# --------------------------------------------------------------------------------------
def gradient_field(chem: np.ndarray, normalized: bool = True, grad_clip: Optional[float] = 1e-5) -> np.ndarray:
    """core/agent/gradient.py:55-71: np.gradient (central, one-sided at the edges, NOT
    periodic), optional normalisation with 0/0 → 0, mask of norms below grad_clip."""
    grad = np.stack(np.gradient(np.asarray(chem, dtype=np.float64)))
    norm = np.sqrt(grad[0] ** 2 + grad[1] ** 2)
    if normalized:
        with np.errstate(divide='ignore', invalid='ignore'):
            grad = np.nan_to_num(np.true_divide(grad, norm))
    if grad_clip is not None:
        grad = grad * (norm >= grad_clip)
    return grad


class RefGradientAgent:
    """core/agent/gradient.py:13-135.  `init_noise` (2, N) replaces the unseeded
    `default_rng().normal(0, .4)` of :50-53; `noise` per forward likewise."""

    def __init__(self, max_agents: int, scale=0.01, deposit=4.0, inertia=0.9, sense_offset=0.,
                 noise_scale=0.025, normalized_grad=True, grad_clip=1e-5, init_noise: Optional[np.ndarray] = None,
                 seed: int = 0):
        self._size = max_agents
        self._seed = seed
        self._calls = 0
        self._noise_scale = noise_scale
        self._scale = scale
        self._deposit = deposit
        self._inertia = inertia
        self._sense_offset_scale = sense_offset
        self._normalized = normalized_grad
        self._grad_clip = grad_clip
        if init_noise is None:
            init_noise = orng.normals2(seed, 0, max_agents, orng.STREAM_INIT_HEADING)
        self._prev_grad = np.array(init_noise, dtype=np.float64)
        self._direction_rads = get_radians(self._prev_grad)
        self._render_grad = None

    def _process_gradient(self, grad, turn_sign):
        return grad

    def _process_momentum(self, grad, noise):
        """:82-91."""
        grad = (1 - self._inertia) * grad + self._inertia * self._prev_grad
        grad = grad + self._noise_scale * noise
        self._prev_grad = grad
        return grad

    def _process_deposit(self, sensed_food):
        return self._deposit * sensed_food

    def forward(self, obs, turn_sign: Optional[np.ndarray] = None, noise: Optional[np.ndarray] = None) -> np.ndarray:
        """:96-124.  No alive masking: every slot senses and acts."""
        agents, medium = obs
        N = agents.shape[1]
        W, H = medium.shape[1:]
        step = self._calls
        self._calls += 1
        if turn_sign is None:
            turn_sign = orng.turn_signs(self._seed, step, N)
        if noise is None:
            noise = orng.normals2(self._seed, step, N, orng.STREAM_NOISE) if self._noise_scale != 0 \
                else np.zeros((2, N))
        action = np.zeros((3, N))
        grad_field = gradient_field(medium[M_CHEM], self._normalized, self._grad_clip)
        off = np.stack(polar2xy(self._sense_offset_scale, self._direction_rads))          # :73-76
        px, py = cell(agents[A_X] + off[0], W), cell(agents[A_Y] + off[1], H)             # clamped, not wrapped
        g = grad_field[:, px, py]
        g = self._process_gradient(g, turn_sign)
        g = self._process_momentum(g, noise)
        self._direction_rads = get_radians(g)
        self._render_grad = grad_field
        sensed_food = medium[M_FOOD][cell(agents[A_X], W), cell(agents[A_Y], H)]
        action[[U_DX, U_DY]] = g * self._scale
        action[U_DEP] = self._process_deposit(sensed_food)
        return action


class RefPhysarumAgent(RefGradientAgent):
    """core/agent/gradient.py:138-219."""

    def __init__(self, max_agents: int, scale=0.005, deposit=4.0, inertia=0.0, sense_offset=0.03,
                 noise_scale=0.0, normalized_grad=True, grad_clip=1e-5, turn_angle=30, sense_angle=90,
                 turn_tolerance=0.1, init_noise=None, seed: int = 0):
        super().__init__(max_agents, scale, deposit, inertia, sense_offset, noise_scale, normalized_grad,
                         grad_clip, init_noise, seed)
        self._turn_radians = np.radians(turn_angle)
        self._sense_radians = np.radians(sense_angle)
        self._rtol = turn_tolerance
        self._direction_rads = discretize(get_radians(self._prev_grad), self._turn_radians)
        self._deposit_mask = 1.
        self.last_undetermined = None

    def _choose_turn(self, drads, turn_sign):
        """:168-193."""
        dir_delta = renormalize_radians(self._direction_rads - drads)
        atol = self._turn_radians * self._rtol
        undetermined_grad = np.isclose(0, drads, rtol=1e-5)
        undetermined_turn = np.isclose(0, dir_delta, rtol=1e-2, atol=atol)
        unseen_grad = np.abs(dir_delta) > self._sense_radians
        undetermined = undetermined_grad | undetermined_turn | unseen_grad
        dir_delta = dir_delta * np.logical_not(undetermined)
        turn = np.array(turn_sign, dtype=np.float64)
        turn[dir_delta > atol] = -1
        turn[dir_delta < -atol] = 1
        turn *= self._turn_radians
        self._deposit_mask = np.logical_not(undetermined_grad | undetermined_turn)
        self.last_undetermined = undetermined
        return turn

    def _process_gradient(self, grad, turn_sign):
        """:195-208,216-219."""
        dr, drads = xy2polar(grad[0], grad[1])
        turn = self._choose_turn(drads, turn_sign)
        directions = renormalize_radians(self._direction_rads + turn)
        dr = 1. if self._normalized else dr
        return np.stack(polar2xy(dr, directions))

    def _process_deposit(self, sensed_food):
        """:210-214."""
        mask = np.clip(self._deposit_mask, 0.1, 1.0)
        return self._deposit * sensed_food * mask


class RefBrownianAgent:
    """core/agent/static.py:31-50 with DataInitializer.action_for/with_noise/build_agents
    (core/data_init.py:159-169,218-220,248-253): `(b-a)*u.round(3)+a`, times alive."""

    def __init__(self, move_scale=0.01, deposit_scale=0.5, seed: int = 0):
        self._scale = move_scale
        self._dep_scale = deposit_scale
        self._seed = seed
        self._calls = 0

    def forward(self, obs, units=None) -> np.ndarray:
        agents, _ = obs
        N = agents.shape[1]
        if units is None:
            units = orng.brownian_units(self._seed, self._calls, N)
        self._calls += 1
        s = self._scale
        u = [np.asarray(q) / 1000.0 for q in units]
        data = np.stack([(s - -s) * u[0] + -s, (s - -s) * u[1] + -s, (self._dep_scale - 0.) * u[2] + 0.])
        return data * agents[A_ALIVE]


class RefConstAgent:
    """core/agent/static.py:9-28."""

    def __init__(self, delta_xy, deposit=0.):
        self._data = (delta_xy[0], delta_xy[1], deposit)

    def forward(self, obs) -> np.ndarray:
        agents, _ = obs
        action = np.zeros((3, agents.shape[1]))
        for c in range(3):
            action[c] = self._data[c]
        return action


# --------------------------------------------------------------------------------------
# Allocation: core/data_init.py
# --------------------------------------------------------------------------------------
def mask_range(sampled: np.ndarray, mask_below=0.0, mask_above=1.0) -> np.ndarray:
    """core/data_init.py:181-185."""
    return sampled * ((mask_below <= sampled) & (sampled <= mask_above))


def agents_channel_from_uniform(u_round3: np.ndarray, ratio: float) -> np.ndarray:
    """core/data_init.py:222-226: ceil(u·[0 ≤ u ≤ ratio])."""
    return np.ceil(mask_range(u_round3, mask_above=ratio))


class RefDataInitializer:
    """core/data_init.py:171-253, the builder: `__init__` (zeros per channel, static mask), `with_const` (:214-216),
    `with_noise` (:218-220 → `get_random`, :168-169: (b − a)·u.round(3) + a), `with_agents` (:222-226: ceil of the uniform
    draw masked to [0, ratio]), `_add_masked` (:206-209), `build_numpy` (:238-239) and the static mask `build` /
    `build_agents` multiply in (:241-253).  The reference draws from numpy's global generator; here every `_get_random`
    takes its uniforms from `draw(shape)` (tests pass the reference's own draws, or the Philox streams the device uses)."""

    def __init__(self, field_size, channels=(), mask=1., draw=None):
        self._size = field_size
        self._channels = {c: np.zeros(field_size) for c in channels}
        self._static_mask = mask
        self._draw = draw

    def _get_random(self, a=0., b=1.):
        return (b - a) * np.asarray(self._draw(self._size), dtype=np.float64).round(3) + a

    def with_const(self, channel, value=0.):
        self._channels[channel] = np.full(self._size, value, dtype=np.float64)
        return self

    def with_noise(self, channel, a=0, b=1):
        self._channels[channel] = self._get_random(a, b)
        return self

    def with_agents(self, ratio):
        self._channels['agents'] = np.ceil(mask_range(self._get_random(), mask_above=ratio))
        return self

    def _add_masked(self, channel, data):
        self._channels[channel] += data * (self._channels[channel] > 0.)
        return self

    def build_numpy(self):
        return np.stack(list(self._channels.values()))

    def build(self):
        return self.build_numpy() * self._static_mask


def agents_from_medium(medium: np.ndarray, food_u_round3: np.ndarray, max_agents: Optional[int] = None,
                       food_ratio: float = 1.0) -> np.ndarray:
    """core/data_init.py:133-150 + core/utils.py:140-151: occupied cells in row-major
    order fill slots [0, K): x = linspace(0,1,W)[ix], y likewise, alive = 1,
    agent_food = (food_ratio − 0.1)·u.round(3) + 0.1; the rest is zero."""
    W, H = medium.shape[1:]
    ix, iy = (medium[M_AGENTS] > 0).nonzero()
    K = ix.shape[0]
    xs, ys = np.linspace(0, 1, W), np.linspace(0, 1, H)
    if not max_agents:
        max_agents = W * H
    agents = np.zeros((4, max_agents))
    agents[A_X, :K] = xs[ix]
    agents[A_Y, :K] = ys[iy]
    agents[A_ALIVE, :K] = 1.
    agents[A_FOOD, :K] = (food_ratio - 0.1) * np.asarray(food_u_round3)[:K] + 0.1
    return agents


@dataclass
class FoodSpec:
    """Synthetic stand-in for the un-vendored `perlin_noise` food field
    (core/data_init.py:190-196,228-231): a sum of torus-periodic sinusoids with the
    positive half kept (≈50 % zeros, max ≈ amp), rounded to 3 decimals like the reference."""
    fx: np.ndarray
    fy: np.ndarray
    phase: np.ndarray
    amp: np.ndarray
    scale: float = 0.5

    @staticmethod
    def from_seed(seed: int, n_waves: int = 6, max_freq: int = 5, scale: float = 0.5) -> 'FoodSpec':
        r = orng._draw(seed, 0, np.arange(n_waves, dtype=np.uint64), orng.STREAM_INIT_FOOD)
        fx = (r[0] % np.uint32(2 * max_freq + 1)).astype(np.int64) - max_freq
        fy = (r[1] % np.uint32(max_freq)).astype(np.int64) + 1
        phase = r[2].astype(np.float64) * (2 * np.pi / 4294967296.0)
        amp = 0.5 + r[3].astype(np.float64) / 4294967296.0
        amp = amp / amp.sum()
        return FoodSpec(fx.astype(np.float64), fy.astype(np.float64), phase, amp, scale)

    def field(self, W: int, H: int) -> np.ndarray:
        x = (np.arange(W, dtype=np.float64) / W)[:, None]
        y = (np.arange(H, dtype=np.float64) / H)[None, :]
        s = np.zeros((W, H))
        for fx, fy, ph, a in zip(self.fx, self.fy, self.phase, self.amp):
            s += a * np.sin(2 * np.pi * (fx * x + fy * y) + ph)
        # the waves sum to at most 1 in magnitude; stretch so typical peaks reach `scale`
        s = np.clip(2.0 * self.scale * s, 0., self.scale)
        return np.round(s, 3)


def perlin2(seed: int, x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """2-D gradient noise (Perlin 1985 / 2002) — what the un-vendored `perlin_noise.PerlinNoise(octaves)` of
    core/data_init.py:190-196 is built from: unit gradients on the integer lattice, corner dot products blended with the
    quintic fade; the caller scales the coordinates by `octaves`.  Lattice gradient angles come from Philox(seed, lattice
    point) — the twin of die_perlin2 in die_amd/csrc/die_rng.h."""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    fx0, fy0 = np.floor(x), np.floor(y)
    i, j = fx0.astype(np.int64), fy0.astype(np.int64)
    tx, ty = x - fx0, y - fy0

    def dot(ii, jj, dx, dy):
        key = ((ii & 0xFFFFF).astype(np.uint64) | ((jj & 0xFFFFF).astype(np.uint64) << np.uint64(20))).ravel()
        r0 = orng._draw(seed, 0, key, orng.STREAM_INIT_FOOD)[0].reshape(dx.shape)
        th = 6.283185307179586476925 * (r0.astype(np.float64) * (1.0 / 4294967296.0))
        return np.cos(th) * dx + np.sin(th) * dy

    d00, d10 = dot(i, j, tx, ty), dot(i + 1, j, tx - 1.0, ty)
    d01, d11 = dot(i, j + 1, tx, ty - 1.0), dot(i + 1, j + 1, tx - 1.0, ty - 1.0)
    u = tx * tx * tx * (tx * (tx * 6.0 - 15.0) + 10.0)
    v = ty * ty * ty * (ty * (ty * 6.0 - 15.0) + 10.0)
    a, b = d00 + u * (d10 - d00), d01 + u * (d11 - d01)
    return a + v * (b - a)


def perlin_field(W: int, H: int, octaves: int, seed: int, threshold: float = 1.0) -> np.ndarray:
    """with_food_perlin / _get_perlin (core/data_init.py:190-196,228-231): noise at the linspace(0, 1, n) labels times
    `octaves`, `.round(3)`, values outside [0, threshold] masked to 0."""
    xs = (np.arange(W, dtype=np.float64) / (W - 1) if W > 1 else np.zeros(1))[:, None] * np.ones((1, H))
    ys = np.ones((W, 1)) * (np.arange(H, dtype=np.float64) / (H - 1) if H > 1 else np.zeros(1))[None, :]
    return mask_range(np.round(perlin2(seed, xs * octaves, ys * octaves), 3), mask_above=threshold)


def synthetic_init(W: int, H: int, ratio: float, seed: int, max_agents: Optional[int] = None, food: str = 'perlin'):
    """Env._init_data (core/env.py:74-86): Perlin food (threshold 1.0, 8 octaves), seeded agents; returns (medium,
    agents).  `food='waves'`: the sinusoid mix of round 1."""
    C = W * H
    u = orng.uniform_round3(seed, 0, C, orng.STREAM_INIT_AGENTS).reshape(W, H)
    medium = np.zeros((3, W, H))
    medium[M_AGENTS] = agents_channel_from_uniform(u, ratio)
    medium[M_FOOD] = perlin_field(W, H, 8, seed) if food == 'perlin' else FoodSpec.from_seed(seed).field(W, H)
    K = int((medium[M_AGENTS] > 0).sum())
    fu = orng.uniform_round3(seed, 0, K, orng.STREAM_INIT_AGENT_FOOD)
    agents = agents_from_medium(medium, fu, max_agents)
    return medium, agents


def wave_field(W: int, H: int, t: float) -> np.ndarray:
    """core/data_init.py:71-89 WaveSequence.__getitem__ (grid from core/utils.py:113-118:
    meshgrid of the reversed sizes, so grid[0] varies along the last axis)."""
    xcs = [np.linspace(0., 1., num=size) for size in reversed((W, H))]
    grid = np.stack(np.meshgrid(*xcs))
    pi = np.pi
    x, y = (grid - 0.5) * 2
    r = np.linalg.norm((x, y), axis=0)
    rwave = r + np.cos(pi * x) + np.sin(0.4 * pi * y)
    z_waves = np.cos(1 * pi * (rwave + t))
    z_islands = (np.sin(pi * x * 3 + t) + np.cos(pi * y * 3 + t))
    mix = 0.25
    return (1 - mix) * z_waves + mix * z_islands


# --------------------------------------------------------------------------------------
# NeuralAutomataAgent sensing: core/agent/evo.py:45-118,150-174
# --------------------------------------------------------------------------------------
_NP_PAD = {'circular': 'wrap', 'zeros': 'constant', 'reflect': 'reflect', 'replicate': 'edge'}


def conv2d_same(x: np.ndarray, w: np.ndarray, mode: str = 'circular') -> np.ndarray:
    """One bias-free Conv2d with padding='same', padding_mode=`mode` (core/agent/evo.py:51,81-92: `boundary`) on a
    (cin, W, H) array: torch's convolution is a cross-correlation, out[o, x, y] = Σ_i Σ_a Σ_b w[o, i, a, b] · in[i, x + a − r,
    y + b − r] over the field padded by r = k // 2 cells per side as torch.nn.functional.pad does ('circular' wraps,
    'zeros' pads with 0, 'reflect' mirrors without repeating the edge, 'replicate' repeats it)."""
    cout, cin, k, k2 = w.shape
    assert k == k2 and k % 2 == 1 and x.shape[0] == cin
    r = k // 2
    W, H = x.shape[1:]
    xp = np.pad(x, ((0, 0), (r, r), (r, r)), mode=_NP_PAD[mode]) if r else x
    out = np.zeros((cout, W, H))
    for a in range(k):
        for b in range(k):
            out += np.einsum('oi,ixy->oxy', w[:, :, a, b], xp[:, a:a + W, b:b + H])
    return out


def conv2d_circular(x: np.ndarray, w: np.ndarray) -> np.ndarray:
    return conv2d_same(x, w, 'circular')


def nca_sense(medium: np.ndarray, weights, with_agent_channel: bool = True, boundary: str = 'circular') -> np.ndarray:
    """ConvolutionModel.forward (core/agent/evo.py:110-118, dropout p = 0): the kernels one after the other, no
    activation in between (:81-99), Tanh at the end."""
    x = np.asarray(medium if with_agent_channel else medium[1:], dtype=np.float64)
    for w in weights:
        x = conv2d_same(x, np.asarray(w, dtype=np.float64), boundary)
    return np.tanh(x)


def nca_forward(obs, weights, scale=0.1, deposit=1.0, with_agent_channel=True, boundary: str = 'circular') -> np.ndarray:
    """NeuralAutomataAgent.forward (core/agent/evo.py:150-174): the transformed medium read at every slot's nearest
    cell (core/utils.py:56-65, only_alive=False), times (scale, scale, deposit)."""
    agents, medium = obs
    W, H = medium.shape[1:]
    sense = nca_sense(medium, weights, with_agent_channel, boundary)
    ix, iy = cell(agents[A_X], W), cell(agents[A_Y], H)
    return sense[:, ix, iy] * np.array([scale, scale, deposit])[:, None]


def perlin3(seed: int, x, y, z) -> np.ndarray:
    """3-D gradient noise, the twin of die_perlin3 (die_amd/csrc/die_rng.h): what `PerlinNoise(octaves)((x, y, t))` of
    core/data_init.py:55-69 is built from.  Lattice gradients uniform on the sphere (z = 2·u1 − 1, azimuth 2π·u2) from
    Philox(seed, step word 1, lattice point); eight corner dot products, quintic fade."""
    x, y, z = np.broadcast_arrays(np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64), np.asarray(z, dtype=np.float64))
    fx0, fy0, fz0 = np.floor(x), np.floor(y), np.floor(z)
    i, j, k = fx0.astype(np.int64), fy0.astype(np.int64), fz0.astype(np.int64)
    tx, ty, tz = x - fx0, y - fy0, z - fz0

    def dot(ii, jj, kk, dx, dy, dz):
        key = ((ii & 0xFFFFF).astype(np.uint64) | ((jj & 0xFFFFF).astype(np.uint64) << np.uint64(20)) |
               ((kk & 0xFFFFF).astype(np.uint64) << np.uint64(40))).ravel()
        r = orng._draw(seed, 1, key, orng.STREAM_INIT_FOOD)
        gz = 2.0 * (r[0].astype(np.float64) * (1.0 / 4294967296.0)).reshape(dx.shape) - 1.0
        gr = np.sqrt(np.maximum(1.0 - gz * gz, 0.0))
        th = 6.283185307179586476925 * (r[1].astype(np.float64) * (1.0 / 4294967296.0)).reshape(dx.shape)
        return gr * np.cos(th) * dx + gr * np.sin(th) * dy + gz * dz

    fade = lambda t: t * t * t * (t * (t * 6.0 - 15.0) + 10.0)
    u, v, w = fade(tx), fade(ty), fade(tz)
    planes = []
    for c in (0, 1):
        dz = tz - float(c)
        d00, d10 = dot(i, j, k + c, tx, ty, dz), dot(i + 1, j, k + c, tx - 1.0, ty, dz)
        d01, d11 = dot(i, j + 1, k + c, tx, ty - 1.0, dz), dot(i + 1, j + 1, k + c, tx - 1.0, ty - 1.0, dz)
        a, b = d00 + u * (d10 - d00), d01 + u * (d11 - d01)
        planes.append(a + v * (b - a))
    return planes[0] + w * (planes[1] - planes[0])


def perlin3_field(W: int, H: int, t: float, octaves: int, seed: int) -> np.ndarray:
    """PerlinNoiseSequence.__getitem__ (core/data_init.py:64-69): noise((x, y, t)) on the linspace(0, 1, n) labels, `.round(3)`."""
    xs = np.linspace(0, 1, W)[:, None] * np.ones((1, H))
    ys = np.ones((W, 1)) * np.linspace(0, 1, H)[None, :]
    return np.round(perlin3(seed, xs * octaves, ys * octaves, np.full((W, H), t * octaves)), 3)


class RefFieldSequence:
    """FieldSequence (core/data_init.py:16-52): `__iter__` cycles over arange(*t_bounds, dt) yielding self[t];
    `get_flow_operator` (:29-38) returns food_flow(current) = scale·next(it) + (1 − decay)·current."""

    def __init__(self, field_size, dt: float = 0.01, t_bounds=(0, 10)):
        self._size = tuple(field_size)
        self._ts = np.arange(*t_bounds, dt)

    def __getitem__(self, t):
        raise NotImplementedError

    def __iter__(self):
        from itertools import cycle
        for t in cycle(self._ts):
            yield self[t]

    def get_flow_operator(self, scale: float = 1.0, decay: float = 0.0):
        it = iter(self)

        def food_flow(current):
            return scale * next(it) + (1 - decay) * current
        return food_flow


class RefPerlinNoiseSequence(RefFieldSequence):
    """core/data_init.py:55-69 (t_bounds (0, 1), octaves 8) with the seeded noise above."""

    def __init__(self, field_size, dt: float = 0.01, t_bounds=(0, 1), octaves: int = 8, seed: int = 0):
        super().__init__(field_size, dt, t_bounds)
        self._octaves, self._seed = int(octaves), int(seed)

    def __getitem__(self, t):
        return perlin3_field(self._size[0], self._size[1], float(t), self._octaves, self._seed)


class RefWaveSequence:
    """FieldSequence / WaveSequence (core/data_init.py:15-89): `__iter__` cycles over arange(*t_bounds, dt) yielding
    wave_field(t); `get_flow_operator` (:29-38) returns food_flow(current) = scale·next(it) + (1 − decay)·current."""

    def __init__(self, field_size, dt: float = 0.01, t_bounds=(0, 10)):
        self._size = tuple(field_size)
        self._ts = np.arange(*t_bounds, dt)

    def __iter__(self):
        from itertools import cycle
        for t in cycle(self._ts):
            yield wave_field(self._size[0], self._size[1], t)

    def get_flow_operator(self, scale: float = 1.0, decay: float = 0.0):
        it = iter(self)

        def food_flow(current):
            return scale * next(it) + (1 - decay) * current
        return food_flow
