"""TEST INFRASTRUCTURE ONLY — never imported by the product path (die_amd/).

Counter-based random streams shared (bit for bit) by the CPU oracle and the HIP kernels.

The reference draws from *unseeded global* numpy generators in four places
(core/agent/gradient.py:33,50-53 `default_rng().normal`, :183 `np.random.randint`,
core/data_init.py:168-169 `np.random.random_sample`), so its runs are not reproducible.
Both sides of the parity tests therefore take their random numbers from one
stateless generator keyed by (seed, step, slot, stream): Philox4x32-10
(Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11).  The same
function is written in HIP in die_amd/csrc/die_rng.h; `tests/test_rng.py` pins this
numpy version with the published Random123 known-answer vectors.

Streams (the 4th counter word):
  STREAM_TURN      per-slot random turn sign of PhysarumAgent._choose_turn (one block of 128 bits per 128 slots: turn_bits)
  STREAM_BROWNIAN  three rounded uniforms of BrownianAgent.forward
  STREAM_NOISE     two normals N(0, .4) of GradientAgent._get_some_noise
  STREAM_INIT_*    synthetic initial fields / agents / headings (data_init)
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)

STREAM_TURN = 0
STREAM_BROWNIAN = 1
STREAM_NOISE = 2
STREAM_INIT_AGENTS = 3
STREAM_INIT_FOOD = 4
STREAM_INIT_HEADING = 5
STREAM_INIT_AGENT_FOOD = 6
STREAM_BUILDER = 7          # DataInitializer.with_noise (core/data_init.py:218-220): call k uses word k % 4 of step k // 4


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  All inputs broadcastable unsigned 32-bit values.
    Returns four uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & MASK32 for c in (c0, c1, c2, c3)]
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0)
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def _draw(seed, step, slot, stream):
    slot = np.asarray(slot, dtype=np.uint64)
    return philox4x32_10(slot & MASK32, slot >> np.uint64(32), np.uint64(step & 0xFFFFFFFF),
                         np.uint64(stream), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def turn_signs(seed: int, step: int, n: int, slots=None) -> np.ndarray:
    """±1 per slot; stands in for `(np.random.randint(0, 2, n) - 0.5) * 2`
    (core/agent/gradient.py:183)."""
    slots = np.arange(n, dtype=np.uint64) if slots is None else np.asarray(slots, dtype=np.uint64)
    return np.where(turn_bits(seed, step, slots), 1.0, -1.0)


def turn_bits(seed: int, step: int, slots: np.ndarray) -> np.ndarray:
    """The random turn bit of each slot id.  One Philox block serves 128 slots (die_amd/csrc/die_rng.h die_turn_word): word
    w = slot >> 5 of the step's bit table is word (w & 3) of Philox(counter = (w >> 2, 0, step, STREAM_TURN), key = seed);
    the slot's bit is bit (slot & 31) of it."""
    slots = np.asarray(slots, dtype=np.uint64)
    w = slots >> np.uint64(5)
    r = philox4x32_10(w >> np.uint64(2), 0, np.uint64(step & 0xFFFFFFFF), np.uint64(STREAM_TURN), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    words = np.choose((w & np.uint64(3)).astype(np.int64), [r[0], r[1], r[2], r[3]])
    return ((words >> (slots & np.uint64(31)).astype(np.uint32)) & np.uint32(1)).astype(bool)


def round3_units(bits: np.ndarray) -> np.ndarray:
    """Integer r in [0, 1000] distributed as round(U*1000) for U = bits/2^32, i.e. the
    numerator of `np.random.random_sample().round(3)` (core/data_init.py:168-169);
    integer arithmetic so that host and device agree exactly."""
    b = bits.astype(np.uint64)
    return ((b * np.uint64(1000) + np.uint64(1 << 31)) >> np.uint64(32)).astype(np.int64)


def uniform_round3(seed: int, step: int, n: int, stream: int, word: int = 0, slots=None) -> np.ndarray:
    slots = np.arange(n, dtype=np.uint64) if slots is None else slots
    return round3_units(_draw(seed, step, slots, stream)[word]) / 1000.0


def brownian_units(seed: int, step: int, n: int):
    """Three round3 numerators (dx, dy, deposit1) per slot."""
    r = _draw(seed, step, np.arange(n, dtype=np.uint64), STREAM_BROWNIAN)
    return round3_units(r[0]), round3_units(r[1]), round3_units(r[2])


def normals2(seed: int, step: int, n: int, stream: int = STREAM_NOISE, scale: float = 0.4):
    """Two independent N(0, scale) per slot via Box–Muller on words 0..1 (shape (2, n))."""
    r = _draw(seed, step, np.arange(n, dtype=np.uint64), stream)
    u1 = (r[0].astype(np.float64) + 1.0) * (1.0 / 4294967296.0)   # (0, 1]
    u2 = r[1].astype(np.float64) * (1.0 / 4294967296.0)          # [0, 1)
    rad = np.sqrt(-2.0 * np.log(u1))
    return scale * np.stack([rad * np.cos(2 * np.pi * u2), rad * np.sin(2 * np.pi * u2)])


def builder_noise(seed: int, call: int, n: int, a: float, b: float) -> np.ndarray:
    """The `call`-th with_noise of a DataInitializer (core/data_init.py:168-169,218-220): (b − a)·u.round(3) + a."""
    return (b - a) * uniform_round3(seed, call // 4, n, STREAM_BUILDER, word=call % 4) + a
