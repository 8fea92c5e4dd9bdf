"""``Gates.<NAME>(...)`` router.

API mirror of ``qml_essentials/gates.py:24-225``: attribute access on the *class*
returns a handler whose ``__name__`` is the gate name (``Block`` and
``is_rotational/is_entangling`` rely on it), unknown keyword arguments are
dropped, ``gate_mode`` selects the backend.  Only the unitary backend exists here:
pulse-level gates synthesise their 2x2/4x4 unitaries by ODE integration
(``evolution.py``), which never touches a statevector and is out of scope.
"""
from __future__ import annotations

from typing import Callable, List, Union

from . import operations as _op
from .unitary import UnitaryGates

_FORWARDED = ("w", "wires", "phi", "theta", "omega", "noise_params", "random_key")
_ROTATIONAL = {"RX", "RY", "RZ", "Rot", "CRX", "CRY", "CRZ", "GolombEncoding", "CPhase"}
_ENTANGLING = {"CX", "CY", "CZ", "CRX", "CRY", "CRZ", "CPhase"}


class _GatesMeta(type):
    def __getattr__(cls, gate_name: str) -> Callable:
        if gate_name.startswith("__"):
            raise AttributeError(gate_name)

        def handler(*args, **kwargs):
            return cls._dispatch(gate_name, *args, **kwargs)

        handler.__name__ = gate_name
        return handler


class Gates(metaclass=_GatesMeta):
    """Dynamic accessor: ``Gates.RX(w, wires=0)`` records an RX on the active tape."""

    def __getattr__(self, gate_name: str) -> Callable:
        def handler(**kwargs):
            return type(self)._dispatch(gate_name, **kwargs)

        handler.__name__ = gate_name
        return handler

    @classmethod
    def _dispatch(cls, gate_name: str, *args, **kwargs):
        if gate_name == "Barrier":
            wires = kwargs.get("wires", args[0] if args else 0)
            return _op.Barrier(wires)
        mode = kwargs.pop("gate_mode", "unitary")
        if mode == "pulse":
            raise NotImplementedError(
                "gate_mode='pulse' (pulse-level simulation) is outside the MI355X hot path"
            )
        if mode != "unitary":
            raise ValueError(f"Unknown gate mode: {mode}. Use 'unitary' or 'pulse'.")
        kwargs = {k: v for k, v in kwargs.items() if k in _FORWARDED}
        gate = getattr(UnitaryGates, gate_name, None)
        if gate is None:
            raise AttributeError(f"'UnitaryGates' object has no attribute '{gate_name}'")
        return gate(*args, **kwargs)

    # kept for signature parity with the reference; there is no pulse manager here
    _inner_getattr = _dispatch

    @classmethod
    def parse_gates(cls, gates: Union[str, Callable, List[Union[str, Callable]], None],
                    set_of_gates=None) -> List[Callable]:
        """Names / callables / lists thereof -> list of gate callables (``gates.py:173-207``)."""
        pool = set_of_gates or cls
        if gates is None:
            return [lambda *a, **k: None]
        if isinstance(gates, str):
            return [getattr(pool, gates)]
        if isinstance(gates, list):
            out = []
            for g in gates:
                if isinstance(g, str):
                    out.append(getattr(pool, g))
                elif callable(g):
                    out.append(g)
                else:
                    raise ValueError(f"Operation {g} is not a valid gate or callable. Got {type(g)}")
            return out
        if callable(gates):
            return [gates]
        raise ValueError(f"Operation {gates} is not a valid gate or callable or list of both.")

    @classmethod
    def is_rotational(cls, gate) -> bool:
        return gate.__name__ in _ROTATIONAL

    @classmethod
    def is_entangling(cls, gate) -> bool:
        return gate.__name__ in _ENTANGLING
