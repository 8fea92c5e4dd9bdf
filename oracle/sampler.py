"""NumPy restatement of the shot sampler (oracle).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

Follows ``qml_essentials/simulation.py:320-377`` (``sample_shots``): draw ``shots`` basis
states from the exact probabilities, histogram them, return ``counts / shots`` or
``diag(O) . counts / shots`` per observable.  The draw itself is ``jax.random.choice(key,
dim, (shots,), p=probs)`` in the reference -- inverse-CDF sampling (``r = cumsum[-1] * u``,
``searchsorted``) on threefry bits, which cannot be reproduced without JAX (SURVEY.md 8-c
"Not pinnable").  This restatement keeps the algorithm and replaces the bit source by the
published Philox4x32-10 counter RNG (Salmon et al., SC'11; pinned in
``tests/test_shots_cpu.py`` by the Random123 known-answer vectors), with the counter layout
the engine documents in ``include/qmle_sv.h``.  Parity status: algorithm pinned by the reference's own statistical
tests (``tests/test_jaqsi.py:1230-1382``); the random stream is this build's own.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Ten Philox rounds on uint32 counter arrays; returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        c1, c3 = p1 & MASK, p0 & MASK
        c0, c2 = n0 & MASK, n2 & MASK
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def u53(hi, lo):
    """(hi:lo >> 11 + 0.5) / 2^53, uniform in (0, 1)."""
    bits = ((hi << np.uint64(32)) | lo) >> np.uint64(11)
    return (bits.astype(np.float64) + 0.5) / 9007199254740992.0


def uniforms(shots, seed, row):
    """The ``shots`` uniforms of batch row ``row``: pair p -> shots 2p, 2p+1."""
    pairs = np.arange((shots + 1) // 2, dtype=np.uint64)
    x0, x1, x2, x3 = philox4x32_10(pairs & MASK, pairs >> np.uint64(32), row & 0xFFFFFFFF,
                                   row >> 32, seed & 0xFFFFFFFF, seed >> 32)
    u = np.empty(2 * pairs.size, dtype=np.float64)
    u[0::2] = u53(x0, x1)
    u[1::2] = u53(x2, x3)
    return u[:shots]


def sample_counts(probs, shots, seed, row=0):
    """Histogram of ``shots`` inverse-CDF draws from one probability row (float32)."""
    cdf = np.cumsum(np.asarray(probs, dtype=np.float32).astype(np.float64))
    idx = np.searchsorted(cdf, cdf[-1] * uniforms(shots, seed, row), side="left")
    return np.bincount(idx, minlength=cdf.size).astype(np.int32)


def sample_shots(probs, type, obs_diags, shots, seed, row=0):
    """simulation.py:350-377.  ``obs_diags``: lifted diagonals (2^n,) per observable."""
    est = sample_counts(probs, shots, seed, row) / shots
    if type == "probs":
        return est
    if type == "expval":
        return np.array([np.real(np.dot(d, est)) for d in obs_diags])
    raise ValueError(f"Shot simulation is only supported for 'probs' and 'expval', got {type!r}.")
