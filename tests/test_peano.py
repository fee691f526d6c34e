"""Peano-Hilbert keys: oracle and library against the reference's known answers."""
import numpy as np


def test_oracle_peano_kats(O, kats):
    for (x, y, z, bits), want in kats["peano_hilbert_key"]:
        assert O.peano_key(x, y, z, bits) == want


def test_library_peano_kats(pkg, have_lib, kats):
    for (x, y, z, bits), want in kats["peano_hilbert_key"]:
        assert pkg.peano_hilbert_key(x, y, z, bits) == want


def test_library_matches_oracle_random(pkg, have_lib, O):
    rng = np.random.default_rng(1)
    for bits in (1, 2, 5, 10, 18, 21):
        xyz = rng.integers(0, 1 << bits, size=(500, 3))
        for x, y, z in xyz:
            assert pkg.peano_hilbert_key(x, y, z, bits) == O.peano_key(x, y, z, bits)


def test_prefix_property(pkg, have_lib):
    """key(21 bits) >> 9 == key(18 bits of the truncated coordinates): what lets the engine sort on 63-bit keys"""
    rng = np.random.default_rng(2)
    for x, y, z in rng.integers(0, 1 << 21, size=(300, 3)):
        assert pkg.peano_hilbert_key(x, y, z, 21) >> 9 == pkg.peano_hilbert_key(x >> 3, y >> 3, z >> 3, 18)


def test_curve_is_a_bijection_and_continuous(O):
    bits = 3
    n = 1 << bits
    cells = {}
    for x in range(n):
        for y in range(n):
            for z in range(n):
                cells[O.peano_key(x, y, z, bits)] = (x, y, z)
    assert sorted(cells) == list(range(n ** 3))
    for k in range(n ** 3 - 1):
        a, b = cells[k], cells[k + 1]
        assert sum(abs(p - q) for p, q in zip(a, b)) == 1   # Hilbert: consecutive cells share a face
