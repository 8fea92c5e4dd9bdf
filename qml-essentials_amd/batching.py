"""Batch tracer: what ``jax.vmap`` tracing gives the reference, without a compiler.

The reference records the circuit ONCE with abstract per-sample values and lets
``jax.vmap`` replay the traced program over the batch
(``qml_essentials/script.py:272-329``).  Here the circuit function is also run
once, but every batched argument is wrapped in a :class:`Batched` value that
*looks* un-batched to circuit code (``shape``, indexing, arithmetic all ignore
the hidden leading batch axis).  A gate built from such a value stores a whole
column of per-sample angles, so one recorded tape + one ``[B, n_slots]`` angle
table describes the full batch -- exactly the input the HIP engine wants.
"""
from __future__ import annotations

import numbers
from typing import Any, Tuple

import numpy as np


class Batched:
    """Array with a hidden leading batch axis (``data.shape == (B, *shape)``)."""

    __array_priority__ = 1000.0
    __slots__ = ("data",)

    def __init__(self, data: np.ndarray):
        self.data = np.asarray(data)

    # --- what circuit code may ask -------------------------------------------------
    @property
    def batch(self) -> int:
        return self.data.shape[0]

    @property
    def shape(self) -> Tuple[int, ...]:
        return self.data.shape[1:]

    @property
    def ndim(self) -> int:
        return self.data.ndim - 1

    @property
    def dtype(self):
        return self.data.dtype

    def __len__(self) -> int:
        if self.ndim == 0:
            raise TypeError("len() of a 0-d batched value")
        return self.shape[0]

    def __getitem__(self, idx) -> "Batched":
        if not isinstance(idx, tuple):
            idx = (idx,)
        return Batched(self.data[(slice(None),) + idx])

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def reshape(self, *shape) -> "Batched":
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        return Batched(self.data.reshape((self.batch,) + tuple(shape)))

    def squeeze(self) -> "Batched":
        keep = [self.batch] + [s for s in self.shape if s != 1]
        return Batched(self.data.reshape(keep))

    def any(self) -> bool:
        return bool(self.data.any())

    def mean(self, axis=None) -> "Batched":
        if axis is None:
            axis = tuple(range(1, self.data.ndim))
        else:
            axis = axis + 1 if axis >= 0 else axis
        return Batched(self.data.mean(axis=axis))

    def astype(self, dt) -> "Batched":
        return Batched(self.data.astype(dt))

    # --- arithmetic (right-aligned broadcasting on the visible shape) --------------
    def _align(self, other: Any):
        a = self.data
        if isinstance(other, Batched):
            b = other.data
            if a.shape[0] != b.shape[0]:
                raise ValueError(f"batch mismatch: {a.shape[0]} vs {b.shape[0]}")
            nd = max(a.ndim, b.ndim)
            a = a.reshape((a.shape[0],) + (1,) * (nd - a.ndim) + a.shape[1:])
            b = b.reshape((b.shape[0],) + (1,) * (nd - b.ndim) + b.shape[1:])
            return a, b
        b = np.asarray(other)
        if b.ndim > self.ndim:
            a = a.reshape((a.shape[0],) + (1,) * (b.ndim - self.ndim) + a.shape[1:])
        return a, b

    def _bin(self, other, fn, swap=False):
        a, b = self._align(other)
        return Batched(fn(b, a) if swap else fn(a, b))

    def __add__(self, o): return self._bin(o, np.add)
    def __radd__(self, o): return self._bin(o, np.add, True)
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return self._bin(o, np.subtract, True)
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __rmul__(self, o): return self._bin(o, np.multiply, True)
    def __truediv__(self, o): return self._bin(o, np.divide)
    def __rtruediv__(self, o): return self._bin(o, np.divide, True)
    def __pow__(self, o): return self._bin(o, np.power)
    def __neg__(self): return Batched(-self.data)
    def __pos__(self): return self

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__" or kwargs.get("out") is not None:
            return NotImplemented
        batch = self.batch
        arrs = []
        nd = max((x.ndim if isinstance(x, Batched) else np.ndim(x)) for x in inputs)
        for x in inputs:
            if isinstance(x, Batched):
                d = x.data
                arrs.append(d.reshape((batch,) + (1,) * (nd - x.ndim) + d.shape[1:]))
            else:
                arrs.append(np.asarray(x))
        return Batched(ufunc(*arrs, **kwargs))

    def __repr__(self) -> str:
        return f"Batched(batch={self.batch}, shape={self.shape})"


def is_batched(x: Any) -> bool:
    return isinstance(x, Batched)


def as_param(x: Any):
    """Normalise a gate parameter to ``float`` or a ``(B,)`` float64 column."""
    if isinstance(x, Batched):
        if x.ndim != 0:
            if int(np.prod(x.shape)) != 1:
                raise ValueError(f"gate parameter must be scalar, got shape {x.shape}")
            x = x.reshape(())
        return np.asarray(x.data, dtype=np.float64)
    if isinstance(x, numbers.Real):
        return float(x)
    if hasattr(x, "detach"):  # torch tensor
        x = x.detach().cpu().numpy()
    arr = np.asarray(x, dtype=np.float64)
    if arr.size != 1:
        raise ValueError(f"gate parameter must be scalar, got shape {arr.shape}")
    return float(arr.reshape(()))


def to_numpy(x: Any):
    """Host ndarray view of numpy / torch / list input (None passes through)."""
    if x is None or isinstance(x, (np.ndarray, Batched)):
        return x
    if hasattr(x, "detach"):
        return x.detach().cpu().numpy()
    return np.asarray(x)
