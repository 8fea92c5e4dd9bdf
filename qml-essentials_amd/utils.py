"""Splittable PRNG keys (host side).

The reference draws parameters with ``jax.random`` (threefry keys,
``model.py:687-693``; ``utils.py:9-13`` ``safe_random_split``).  Bit-identical
samples are impossible without JAX (SURVEY.md 8-c "Not pinnable"), so this module
only reproduces the *interface*: ``key(seed)``, ``split``, ``uniform`` -- on NumPy's
Philox counter RNG, which like threefry is keyed and splittable.
"""
from __future__ import annotations

import os
from typing import Optional, Sequence, Union

import numpy as np


# ---------------------------------------------------------------------------------------------
# complex128 mode.  The reference selects its dtype through the global JAX switch
# ``jax.config.update("jax_enable_x64", True)`` (``operations.py:12-16``; its own tests switch it
# on: tests/test_coefficients.py:19, test_entanglement.py:13, test_ansaetze.py:18).  Same here:
# ``enable_x64()`` makes every noise-free, shot-free execution run on the complex128 engine
# (``qmle_run_batch_f64``): float64 angles, complex128 states, float64 results.  ``Model(...,
# x64=True)`` scopes it to one model.  Default off (complex64, like JAX's default).
# ---------------------------------------------------------------------------------------------
_X64 = False


def enable_x64(on: bool = True) -> None:
    global _X64
    _X64 = bool(on)


def x64_enabled() -> bool:
    return _X64


class x64_scope:
    """``with x64_scope(flag):`` -- temporarily force the mode (``None`` leaves it alone)."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        global _X64
        self.prev = _X64
        if self.on is not None:
            _X64 = bool(self.on)
        return self

    def __exit__(self, *exc):
        global _X64
        _X64 = self.prev
        return False


class PRNGKey:
    """Immutable key; ``split`` derives independent children."""

    __slots__ = ("_seq",)

    def __init__(self, seed: Union[int, np.random.SeedSequence] = 0):
        self._seq = seed if isinstance(seed, np.random.SeedSequence) else np.random.SeedSequence(int(seed))

    def split(self, num: int = 2):
        return [PRNGKey(s) for s in self._seq.spawn(num)]

    def generator(self) -> np.random.Generator:
        return np.random.Generator(np.random.Philox(self._seq))

    def __repr__(self) -> str:
        return f"PRNGKey(entropy={self._seq.entropy}, spawn_key={self._seq.spawn_key})"


def key(seed: int) -> PRNGKey:
    return PRNGKey(seed)


def as_key(k) -> Optional[PRNGKey]:
    if k is None or isinstance(k, PRNGKey):
        return k
    if isinstance(k, (int, np.integer)):
        return PRNGKey(int(k))
    raise TypeError(f"random_key must be an int seed or a PRNGKey, got {type(k)}")


def key_to_seed(k) -> int:
    """64-bit seed of a key (what the device sampler's Philox stream is keyed with)."""
    lo, hi = as_key(k)._seq.generate_state(2, dtype=np.uint32)
    return (int(hi) << 32) | int(lo)


def safe_random_split(random_key: Optional[PRNGKey], *args, num: int = 2, **kwargs):
    """``split`` that tolerates ``None`` (``qml_essentials/utils.py:9-13``)."""
    if random_key is None:
        return [None] * num if num != 2 else (None, None)
    parts = as_key(random_key).split(num)
    return tuple(parts) if num == 2 else parts


def _philox_words(k: PRNGKey) -> np.ndarray:
    """The two 64-bit key words numpy's Philox takes from the key's SeedSequence (cached per key:
    the hash behind ``generate_state`` costs ~10 us, an analysis loop re-seeded with the same key
    every call should not pay it each time)."""
    ent = k._seq.entropy  # (an int, or a sequence of ints: SeedSequence([1, 2, 3]))
    if isinstance(ent, (list, tuple, np.ndarray)):
        ent = tuple(int(e) for e in np.asarray(ent).ravel())
    ident = (ent, tuple(k._seq.spawn_key), k._seq.pool_size)
    try:
        hash(ident)
    except TypeError:  # an entropy of some other unhashable kind: no caching
        return k._seq.generate_state(2, np.uint64)
    w = _WORDS.get(ident)
    if w is None:
        if len(_WORDS) > 4096:
            _WORDS.clear()
        w = _WORDS[ident] = k._seq.generate_state(2, np.uint64)
    return w


_WORDS: dict = {}
DEVICE_SAMPLER_MIN = 16384  # draws of at least this many values are written by the GPU


def device_sampling(n: int) -> bool:
    """True when a draw of ``n`` values comes from the GPU sampler (``uniform`` copies it back,
    ``uniform_device`` leaves it in HBM)."""
    return (n >= DEVICE_SAMPLER_MIN and not os.environ.get("QMLE_HOST_SAMPLER")
            and not os.environ.get("QMLE_NUMPY_SAMPLER") and _gpu_present())


def uniform_device(random_key: PRNGKey, shape: Sequence[int], minval: float = 0.0,
                   maxval: float = 1.0):
    """:func:`uniform` left where the GPU wrote it: a float32 CUDA tensor of ``shape`` holding
    numpy's Philox stream bit for bit (``qmle_philox_uniform_f32_device``).  What
    ``jax.random.uniform`` hands the reference (``model.py:687-693``) is a device array too."""
    from . import _native as N

    shape = tuple(int(d) for d in shape)
    n = int(np.prod(shape)) if shape else 1
    return N.philox_uniform_device(_philox_words(as_key(random_key)), n, minval, maxval).reshape(shape)


def uniform(random_key: PRNGKey, shape: Sequence[int], minval: float = 0.0,
            maxval: float = 1.0) -> np.ndarray:
    # numpy's Philox4x64-10 stream under the key's SeedSequence, produced by the library's host-side
    # generator (csrc/qmle_rng.cpp: the same floats, bit for bit -- tests/test_abi_cpu.py -- at a
    # fraction of numpy's ~9 ns per value; QMLE_NUMPY_SAMPLER=1 takes numpy's own loop)
    k = as_key(random_key)
    shape = tuple(int(d) for d in shape)
    if os.environ.get("QMLE_NUMPY_SAMPLER"):
        return k.generator().uniform(minval, maxval, size=shape).astype(np.float32)
    from . import _native as N

    n = int(np.prod(shape)) if shape else 1
    # large draws on a GPU box: the same stream written by the GPU (one work item per Philox block)
    # and copied back -- the host loop costs 0.14 ms per 73 728 values in a hot loop and 0.4 ms inside
    # an analysis loop that has just waited for the GPU; QMLE_HOST_SAMPLER=1 keeps the host loop
    if device_sampling(n):
        return uniform_device(k, shape, minval, maxval).cpu().numpy()
    return N.philox_uniform(_philox_words(k), n, minval, maxval).reshape(shape)


def _gpu_present() -> bool:
    try:
        import torch

        return bool(torch.cuda.is_available())
    except Exception:  # pragma: no cover
        return False


class random:  # namespace so that ``from ...utils import random; random.key(0)`` reads like jax
    key = staticmethod(key)
    PRNGKey = staticmethod(key)
    split = staticmethod(lambda k, num=2: as_key(k).split(num))
    uniform = staticmethod(uniform)
