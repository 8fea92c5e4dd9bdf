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
    """Array with a hidden leading batch axis (``data.shape == (B, *shape)``).

    Optionally carries *tangents* -- forward-mode derivatives with respect to leaf
    arguments of the circuit function -- as a list of terms ``(leaf_id, index, coef)``:
    element ``e`` of this value depends on element ``index[e]`` (flat index into the leaf's
    per-sample array) with coefficient ``coef[:, e]``.  Gate angles are (bi)linear in the
    leaves (``params[l][j]``, ``inputs[f] * enc_params[q, f] * 2**q``), so tracking sums,
    constant scaling and products is enough; any other transformation drops the tangents
    (:meth:`Script.gradient` then refuses to differentiate through that gate).
    """

    __array_priority__ = 1000.0
    __slots__ = ("data", "tan")

    def __init__(self, data: np.ndarray, tan=None):
        data = np.asarray(data)
        if data.dtype.kind == "f" and data.dtype.itemsize < 8:
            # host angle arithmetic runs in fp64 (inputs * 3**q reaches 1e4 rad); the angle
            # table is cast to float32 once, after the reduction mod 4 pi
            data = data.astype(np.float64)
        self.data = data
        self.tan = tan

    @classmethod
    def leaf(cls, data: np.ndarray, leaf_id: int) -> "Batched":
        """Differentiable leaf: every element depends on itself with coefficient 1."""
        data = np.asarray(data)
        idx = np.arange(int(np.prod(data.shape[1:], dtype=np.int64))).reshape(data.shape[1:])
        return cls(data, [(leaf_id, idx, np.ones_like(data, dtype=np.float64))])

    def _map_tan(self, f_idx, f_coef):
        if self.tan is None:
            return None
        return [(lid, f_idx(idx), f_coef(coef)) for lid, idx, coef in self.tan]

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
        return Batched(self.data[(slice(None),) + idx],
                       self._map_tan(lambda i: i[idx], lambda c: c[(slice(None),) + idx]))

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def reshape(self, *shape) -> "Batched":
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        shape = tuple(shape)
        return Batched(self.data.reshape((self.batch,) + shape),
                       self._map_tan(lambda i: i.reshape(shape),
                                     lambda c: c.reshape((self.batch,) + shape)))

    def squeeze(self) -> "Batched":
        keep = tuple(s for s in self.shape if s != 1)
        return self.reshape(keep)

    def any(self) -> bool:
        return bool(self.data.any())

    def mean(self, axis=None) -> "Batched":
        if axis is None:
            axis = tuple(range(1, self.data.ndim))
        else:
            axis = axis + 1 if axis >= 0 else axis
        return Batched(self.data.mean(axis=axis))

    def astype(self, dt) -> "Batched":
        return Batched(self.data.astype(dt), self.tan)

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

    @staticmethod
    def _scaled(tan, factor, out_shape):
        """tangent terms times ``factor`` (broadcast to ``out_shape`` = (B, *shape))."""
        if tan is None:
            return []
        out = []
        for lid, idx, coef in tan:
            c = np.broadcast_to(coef.reshape((coef.shape[0],) + (1,) * (len(out_shape) - coef.ndim)
                                             + coef.shape[1:]) * factor, out_shape)
            i = np.broadcast_to(idx.reshape((1,) * (len(out_shape) - 1 - idx.ndim) + idx.shape),
                                out_shape[1:])
            out.append((lid, i, np.array(c, dtype=np.float64)))
        return out

    def _bin(self, other, fn, swap=False, kind=None):
        a, b = self._align(other)
        res = fn(b, a) if swap else fn(a, b)
        tan = None  # None = derivative unknown; [] = known to be constant
        o_tan = other.tan if isinstance(other, Batched) else []
        if self.tan is not None and o_tan is not None and kind is not None:
            shp = res.shape
            if kind == "add":
                tan = self._scaled(self.tan, 1.0, shp) + self._scaled(o_tan, 1.0, shp)
            elif kind == "sub":
                sa, sb = (-1.0, 1.0) if swap else (1.0, -1.0)
                tan = self._scaled(self.tan, sa, shp) + self._scaled(o_tan, sb, shp)
            elif kind == "mul":  # product rule; a constant `other` has no tangent
                tan = self._scaled(self.tan, b, shp) + self._scaled(o_tan, a, shp)
            elif kind == "div" and not swap and not o_tan:
                tan = self._scaled(self.tan, 1.0 / b, shp)
            elif kind == "div":
                tan = None
        return Batched(res, tan)

    def __add__(self, o): return self._bin(o, np.add, kind="add")
    def __radd__(self, o): return self._bin(o, np.add, True, kind="add")
    def __sub__(self, o): return self._bin(o, np.subtract, kind="sub")
    def __rsub__(self, o): return self._bin(o, np.subtract, True, kind="sub")
    def __mul__(self, o): return self._bin(o, np.multiply, kind="mul")
    def __rmul__(self, o): return self._bin(o, np.multiply, True, kind="mul")
    def __truediv__(self, o): return self._bin(o, np.divide, kind="div")
    def __rtruediv__(self, o): return self._bin(o, np.divide, True)
    def __pow__(self, o): return self._bin(o, np.power)
    def __neg__(self):
        return Batched(-self.data,
                       None if self.tan is None else self._scaled(self.tan, -1.0, self.data.shape))
    def __pos__(self): return self

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__" or kwargs.get("out") is not None:
            return NotImplemented
        table = {np.add: "__add__", np.subtract: "__sub__", np.multiply: "__mul__",
                 np.true_divide: "__truediv__", np.negative: "__neg__"}
        if ufunc in table and len(inputs) <= 2 and not kwargs:
            if len(inputs) == 1:
                return -inputs[0]
            x, y = inputs
            if isinstance(x, Batched):
                return getattr(x, table[ufunc])(y)
            return getattr(y, "__r" + table[ufunc][2:])(x)
        batch = self.batch
        arrs = []
        nd = max((x.ndim if isinstance(x, Batched) else np.ndim(x)) for x in inputs)
        for x in inputs:
            if isinstance(x, Batched):
                d = x.data
                arrs.append(d.reshape((batch,) + (1,) * (nd - x.ndim) + d.shape[1:]))
            else:
                arrs.append(np.asarray(x))
        const = all((x.tan == [] if isinstance(x, Batched) else True) for x in inputs)
        return Batched(ufunc(*arrs, **kwargs), [] if const else None)  # non-linear: unknown

    def __repr__(self) -> str:
        state = "unknown" if self.tan is None else ("const" if not self.tan else "tracked")
        return f"Batched(batch={self.batch}, shape={self.shape}, tan={state})"


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


def param_tangent(x: Any):
    """Tangent terms of a scalar gate parameter: list of (leaf_id, flat_index, coef (B,)),
    ``[]`` for constants, ``None`` if the value is batched but its derivative is unknown."""
    if not isinstance(x, Batched):
        return []
    if x.tan is None:
        return None
    out = []
    for lid, idx, coef in x.tan:
        out.append((lid, int(np.asarray(idx).reshape(-1)[0]),
                    np.asarray(coef, dtype=np.float64).reshape(coef.shape[0], -1)[:, 0]))
    return out


def to_numpy(x: Any):
    """Host ndarray view of numpy / torch / list input (None passes through)."""
    if x is None or isinstance(x, Batched):
        return x
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    elif isinstance(x, (list, tuple, numbers.Number, np.generic)):
        x = np.asarray(x)
    if isinstance(x, np.ndarray) and x.dtype.kind == "f" and x.dtype.itemsize < 8:
        return x.astype(np.float64)  # host angle arithmetic in fp64, see Batched.__init__
    return x  # dicts (noise_params passed positionally), PRNG keys, strings: not array data
