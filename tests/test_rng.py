"""Philox4x32-10 (oracle/rng.py) against the published Random123 known-answer vectors."""
import numpy as np

from oracle import rng


def test_philox_kat():
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        got = rng.philox4x32_10(*[np.array([c]) for c in ctr], key[0], key[1])
        assert tuple(int(g[0]) for g in got) == want


def test_round3_units_distribution_and_exactness():
    bits = np.array([0, 1, 2147483, 2147484, 0xFFFFFFFF, 0x80000000], dtype=np.uint32)
    r = rng.round3_units(bits)
    want = np.floor(bits.astype(np.float64) / 2 ** 32 * 1000 + 0.5).astype(np.int64)
    assert (r == want).all()
    assert r.min() >= 0 and r.max() <= 1000
    big = rng.round3_units(rng._draw(5, 0, np.arange(200000, dtype=np.uint64), rng.STREAM_BROWNIAN)[0])
    assert abs(big.mean() - 500) < 3


def test_turn_signs_balanced_and_stateless():
    a = rng.turn_signs(7, 3, 100000)
    assert set(np.unique(a)) == {-1.0, 1.0}
    assert abs(a.mean()) < 0.02
    b = rng.turn_signs(7, 3, 100000)
    assert (a == b).all()
    sub = rng.turn_signs(7, 3, 0, slots=np.array([5, 99999], dtype=np.uint64))
    assert (sub == a[[5, 99999]]).all()
    assert (rng.turn_signs(7, 4, 1000) != a[:1000]).any()
