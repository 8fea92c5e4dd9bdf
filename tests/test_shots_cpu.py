"""CPU tests of the shot-sampling oracle (``oracle/sampler.py``): Philox4x32-10 pinned by the
Random123 known-answer vectors, the sampler by the statistical properties the reference
tests (``tests/test_jaqsi.py:1230-1382``)."""
import numpy as np
import pytest

from oracle import sampler as S

KAT = [  # Random123 kat_vectors, philox4x32 10: counter[4] key[2] -> out[4]
    ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
    ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
    ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
     [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
]


@pytest.mark.parametrize("ctr,key,want", KAT)
def test_philox_known_answers(ctr, key, want):
    out = S.philox4x32_10(*[np.array([c], dtype=np.uint64) for c in ctr], *key)
    assert [int(x[0]) for x in out] == want


def test_uniforms_are_open_interval_and_reproducible():
    u = S.uniforms(10001, seed=0xDEADBEEFCAFE, row=3)
    assert u.shape == (10001,) and u.min() > 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.02 and abs(u.var() - 1 / 12) < 0.01
    assert np.array_equal(u, S.uniforms(10001, 0xDEADBEEFCAFE, 3))
    assert not np.array_equal(u, S.uniforms(10001, 0xDEADBEEFCAFE, 4))
    # odd shot counts drop only the last half-pair
    assert np.array_equal(S.uniforms(7, 5, 0), S.uniforms(8, 5, 0)[:7])


def test_sampler_properties():
    """Bell-like probabilities: counts sum to shots, zero-probability states are never drawn,
    estimates converge (test_jaqsi.py:1232-1258), different keys differ (:1293-1309)."""
    probs = np.array([0.5, 0.0, 0.0, 0.5], dtype=np.float32)
    c = S.sample_counts(probs, 4096, seed=42)
    assert c.sum() == 4096 and c[1] == 0 and c[2] == 0
    est = S.sample_shots(probs, "probs", [], 100000, seed=123)
    assert np.allclose(est, probs, atol=0.02) and np.isclose(est.sum(), 1.0)
    z0 = np.array([1, 1, -1, -1.0])
    ev = S.sample_shots(probs, "expval", [z0], 100, seed=7)
    assert -1.0 <= ev[0] <= 1.0
    assert not np.array_equal(S.sample_counts(probs, 100, 0), S.sample_counts(probs, 100, 1))
    with pytest.raises(ValueError, match="only supported for 'probs' and 'expval'"):
        S.sample_shots(probs, "state", [], 10, 0)


def test_sampler_chi_square():
    rng = np.random.default_rng(5)
    p = rng.random(64).astype(np.float32)
    p /= p.sum()
    shots = 200000
    c = S.sample_counts(p, shots, seed=99, row=17)
    chi2 = np.sum((c - shots * p) ** 2 / (shots * p))
    assert chi2 < 63 + 5 * np.sqrt(2 * 63)  # 5 sigma
