"""Ansatz library and input encodings (pure-Python gate-list generators).

API mirror of ``qml_essentials/ansaetze.py``: ``Circuit`` / ``DeclarativeCircuit``
(``:13-221``), ``Block`` (``:224-371``), the 23 ``Ansaetze.<Name>`` classes
(``:374-756``) and ``Encoding`` (``:759-1000``).  The structures below are a table
of ``(gate, topology, kwargs)`` rows; wire pairs come from :mod:`topologies`
(pinned by ``tests/golden/topologies.json``), the per-ansatz tapes are pinned
against ``oracle/circuits.py`` in ``tests/test_frontend_cpu.py``.
"""
from __future__ import annotations

import logging
import warnings
from abc import ABC, abstractmethod
from typing import Any, Callable, List, Optional, Tuple, Union

import numpy as np

from .gates import Gates
from .topologies import Topology

log = logging.getLogger(__name__)


class Circuit(ABC):
    """One ansatz layer: ``circuit(w, n_qubits, **kwargs)`` records its gates."""

    @abstractmethod
    def n_params_per_layer(self, n_qubits: int) -> int:
        raise NotImplementedError("n_params_per_layer method is not implemented")

    def n_pulse_params_per_layer(self, n_qubits: int) -> int:
        raise NotImplementedError("pulse-level simulation is outside the MI355X hot path")

    @abstractmethod
    def get_control_indices(self, n_qubits: int) -> Optional[List[int]]:
        raise NotImplementedError("get_control_indices method is not implemented")

    def get_control_angles(self, w, n_qubits: int):
        """Weights that feed controlled rotations (``ansaetze.py:75-94``)."""
        idx = self.get_control_indices(n_qubits)
        if idx is None:
            return np.array([])
        if len(idx) == 3 and None in idx:
            return w[idx[0]:idx[1]:idx[2]]
        return np.asarray(w)[np.asarray(idx)]

    def _build(self, w, n_qubits: int, **kwargs: Any) -> Any:
        if kwargs.get("gate_mode", "unitary") == "pulse":
            raise NotImplementedError("gate_mode='pulse' is outside the MI355X hot path")
        return self.build(w, n_qubits, **kwargs)

    @abstractmethod
    def build(self, w, n_qubits: int, **kwargs: Any) -> Any:
        raise NotImplementedError("build method is not implemented")

    def __call__(self, *args: Any, **kwds: Any) -> Any:
        self._build(*args, **kwds)


class Block:
    """One row of an ansatz: a gate applied to every wire, or to every pair of a topology."""

    def __init__(self, gate: Union[str, Callable], topology: Any = None, **kwargs) -> None:
        self.gate = getattr(Gates, gate) if isinstance(gate, str) else gate
        if self.is_entangling and topology is None:
            raise AssertionError("Topology must be specified for entangling gates")
        self.topology = topology
        self.kwargs = kwargs

    def __repr__(self) -> str:
        inner = self.gate.__name__
        if self.topology is not None:
            inner = f"{self.topology.__name__}[{inner}]"
        return f"Block({inner})"

    @property
    def is_entangling(self) -> bool:
        return Gates.is_entangling(self.gate)

    @property
    def is_rotational(self) -> bool:
        return Gates.is_rotational(self.gate)

    @property
    def is_controlled_rotation(self) -> bool:
        return self.is_entangling and self.is_rotational

    def enough_qubits(self, n_qubits: int) -> bool:
        if not self.is_entangling:
            return n_qubits >= 1
        span = self.kwargs.get("span", 1)
        span = span(n_qubits) if callable(span) else span
        return n_qubits >= 2 and n_qubits > span

    def _pairs(self, n_qubits: int):
        return self.topology(n_qubits=n_qubits, **self.kwargs)

    def _warn_skip(self, n_qubits: int) -> None:
        warnings.warn(
            f"Skipping {self.topology.__name__} with n_qubits={n_qubits} "
            "as there are not enough qubits for this topology."
        )

    def n_params(self, n_qubits: int) -> int:
        assert n_qubits > 0, "Number of qubits must be positive"
        if not self.is_rotational:
            return 0
        if self.is_entangling:
            if not self.enough_qubits(n_qubits):
                self._warn_skip(n_qubits)
                return 0
            return len(self._pairs(n_qubits))
        return 3 * n_qubits if self.gate.__name__ == "Rot" else n_qubits

    def n_pulse_params(self, n_qubits: int) -> int:
        raise NotImplementedError("pulse-level simulation is outside the MI355X hot path")

    def apply(self, n_qubits: int, w=None, w_idx: Optional[int] = None, **kwargs) -> int:
        """Record the block; weights are consumed in wire / pair order
        (``ansaetze.py:323-371``).  Returns the next unused weight index."""
        assert n_qubits > 0, "Number of qubits must be positive"
        if self.is_entangling:
            if not self.enough_qubits(n_qubits):
                for _ in self._pairs(n_qubits):
                    self._warn_skip(n_qubits)
                return w_idx
            targets = [list(p) for p in self._pairs(n_qubits)]
        else:
            targets = list(range(n_qubits))
        rotational, is_rot3 = self.is_rotational, self.gate.__name__ == "Rot"
        if rotational:
            assert w is not None, "w must be provided for rotational gates"
            assert w_idx is not None, "w_idx must be provided for rotational gates"
        for wires in targets:
            if not rotational:
                self.gate(wires=wires, **kwargs)
            elif is_rot3:
                self.gate(w[w_idx], w[w_idx + 1], w[w_idx + 2], wires=wires, **kwargs)
                w_idx += 3
            else:
                self.gate(w[w_idx], wires=wires, **kwargs)
                w_idx += 1
        return w_idx


class DeclarativeCircuit(Circuit):
    """Circuit given by ``structure()`` -> tuple of :class:`Block`; every block is
    followed by a Barrier on all wires (``ansaetze.py:215-221``)."""

    @classmethod
    def structure(cls) -> Tuple[Block, ...]:
        raise NotImplementedError

    @classmethod
    def n_params_per_layer(cls, n_qubits: int) -> int:
        return sum(b.n_params(n_qubits) for b in cls.structure())

    @classmethod
    def n_pulse_params_per_layer(cls, n_qubits: int) -> int:
        raise NotImplementedError("pulse-level simulation is outside the MI355X hot path")

    @classmethod
    def get_control_indices(cls, n_qubits: int) -> Optional[List]:
        """``[-k, None, None]`` when the controlled-rotation weights are the last k of the
        layer (the common case), the raw index list otherwise, ``None`` if there are none."""
        picked, offset = [], 0
        for block in cls.structure():
            k = block.n_params(n_qubits)
            if block.is_controlled_rotation:
                picked.extend(range(offset, offset + k))
            offset += k
        if not picked:
            return None
        if picked == list(range(offset - len(picked), offset)):
            return [-len(picked), None, None]
        return picked

    @classmethod
    def build(cls, w, n_qubits: int, **kwargs: Any) -> None:
        cursor = 0
        every_wire = list(range(n_qubits))
        for block in cls.structure():
            cursor = block.apply(n_qubits, w, cursor, **kwargs)
            Gates.Barrier(wires=every_wire, **kwargs)


# ---- the 23 structures (ansaetze.py:408-756), one row per Block -------------------------------
_S, _B, _A = "stairs", "bricks", "all_to_all"
_ring = dict(wrap=True, reverse=True, mirror=False)
_far = dict(reverse=False, mirror=False, offset=lambda n: n - 1, span=3, wrap=True)
_TABLE = {
    "No_Ansatz": (),
    "Circuit_1": (("RX",), ("RZ",)),
    "Circuit_2": (("RX",), ("RZ",), ("CX", _S, {})),
    "Circuit_3": (("RX",), ("RZ",), ("CRZ", _S, {})),
    "Circuit_4": (("RX",), ("RZ",), ("CRX", _S, {})),
    "Circuit_5": (("RX",), ("RZ",), ("CRZ", _A, {}), ("RX",), ("RZ",)),
    "Circuit_6": (("RX",), ("RZ",), ("CRX", _A, {}), ("RX",), ("RZ",)),
    "Circuit_7": (("RX",), ("RZ",), ("CRZ", _B, {}), ("RX",), ("RZ",), ("CRZ", _B, dict(offset=1))),
    "Circuit_8": (("RX",), ("RZ",), ("CRX", _B, {}), ("RX",), ("RZ",), ("CRX", _B, dict(offset=1))),
    "Circuit_9": (("H",), ("CZ", _S, {}), ("RX",)),
    "Circuit_10": (("RY",), ("CZ", _S, dict(offset=-1, wrap=True)), ("RY",)),
    "Circuit_13": (("RY",), ("CRZ", _S, _ring), ("RY",), ("CRZ", _S, _far)),
    "Circuit_14": (("RY",), ("CRX", _S, _ring), ("RY",), ("CRX", _S, _far)),
    "Circuit_15": (("RY",), ("CX", _S, _ring), ("RY",), ("CX", _S, _far)),
    "Circuit_16": (("RX",), ("RZ",), ("CRZ", _B, {}), ("CRZ", _B, dict(offset=1))),
    "Circuit_17": (("RX",), ("RZ",), ("CRX", _B, {}), ("CRX", _B, dict(offset=1))),
    "Circuit_18": (("RX",), ("RZ",), ("CRZ", _S, dict(wrap=True, mirror=False))),
    "Circuit_19": (("RX",), ("RZ",), ("CRX", _S, dict(wrap=True, mirror=False))),
    "Circuit_20": (("RY",), ("CX", _S, _ring), ("RY",),
                   ("CX", _S, dict(reverse=False, offset=lambda n: n - 2, span=1, wrap=True))),
    "No_Entangling": (("Rot",),),
    "Hardware_Efficient": (("RY",), ("RZ",), ("RY",), ("CX", _B, dict(mirror=False)),
                           ("CX", _B, dict(offset=-1, modulo=True, wrap=True, mirror=False))),
    "Strongly_Entangling": (("Rot",), ("CX", _S, dict(wrap=True, reverse=False, mirror=False)),
                            ("Rot",),
                            ("CX", _S, dict(reverse=False, span=lambda n: n // 2, wrap=True,
                                            mirror=False))),
}
_PARAMETERISED = [f"Circuit_{i}" for i in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 13, 14, 15, 16, 17, 18,
                                             19, 20)] + ["No_Entangling", "Strongly_Entangling",
                                                         "Hardware_Efficient"]


def _make_declarative(name: str, rows) -> type:
    def structure(cls):
        return tuple(
            Block(gate=r[0]) if len(r) == 1
            else Block(gate=r[0], topology=getattr(Topology, r[1]), **r[2])
            for r in rows
        )

    return type(name, (DeclarativeCircuit,), {"structure": classmethod(structure),
                                              "__qualname__": f"Ansaetze.{name}"})


class _GHZ(DeclarativeCircuit):
    """H on wire 0, then a CX chain; hand-written build (``ansaetze.py:415-433``)."""

    @classmethod
    def structure(cls):
        return (Block(gate=Gates.H), Block(gate=Gates.CX, topology=Topology.stairs, reverse=True))

    @classmethod
    def build(cls, w, n_qubits: int, **kwargs) -> None:
        Gates.H(wires=0, **kwargs)
        for q in range(n_qubits - 1):
            Gates.CX(wires=[q, q + 1], **kwargs)


_GHZ.__name__ = "GHZ"
_GHZ.__qualname__ = "Ansaetze.GHZ"


class Ansaetze:
    """Namespace of the available ansatz classes (``Ansaetze.Hardware_Efficient`` ...)."""

    GHZ = _GHZ

    @staticmethod
    def get_available(parameterized_only: bool = False) -> List[type]:
        names = list(_PARAMETERISED)
        if not parameterized_only:
            names += ["No_Ansatz", "GHZ"]
        return [getattr(Ansaetze, n) for n in names]


for _name, _rows in _TABLE.items():
    setattr(Ansaetze, _name, _make_declarative(_name, _rows))


class Encoding:
    """Input-encoding strategy wrapping one gate per input feature
    (``ansaetze.py:759-1000``; Peters & Schuld, Quantum 7, 1210 (2023))."""

    STRATEGIES = ("hamming", "binary", "ternary", "golomb")

    def __init__(self, strategy: str, gates: Union[str, Callable, List[Union[str, Callable]]]):
        if strategy not in self.STRATEGIES:
            raise ValueError(
                f"Encoding strategy {strategy} not implemented. "
                "Available options: ['hamming', 'binary', 'ternary', 'golomb']"
            )
        self._strategy = strategy
        wrap = getattr(self, strategy)
        if strategy == "golomb":
            self._gates = []
            self.callable = [wrap(None)]
        else:
            try:
                self._gates = Gates.parse_gates(gates, Gates)
            except ValueError as e:
                raise ValueError(f"Error parsing encodings: {e}")
            self.callable = [wrap(g) for g in self._gates]

    def __len__(self) -> int:
        return len(self.callable)

    def __getitem__(self, idx):
        return self.callable[idx]

    @property
    def is_golomb(self) -> bool:
        return self._strategy == "golomb"

    def _golomb_max_mark(self) -> int:
        from .unitary import golomb_ruler

        n = getattr(self, "_n_qubits", None)
        if n is None:
            raise ValueError("Golomb encoding requires n_qubits to be set")
        return max(golomb_ruler(2**n))

    def get_n_freqs(self, omegas: int) -> int:
        """Number of frequencies (both signs + 0) for ``omegas`` encoding gates."""
        if self._strategy == "hamming":
            return int(2 * omegas + 1)
        if self._strategy == "binary":
            return int(2 ** (omegas + 1) - 1)
        if self._strategy == "ternary":
            return int(3**omegas)
        return int(2 * omegas * self._golomb_max_mark() + 1)

    def get_spectrum(self, omegas: int) -> np.ndarray:
        """Integer spectrum reachable with ``omegas`` encoding gates."""
        if self._strategy == "hamming":
            top = omegas
        elif self._strategy == "binary":
            top = 2**omegas - 1
        elif self._strategy == "ternary":
            top = int(np.floor(3**omegas / 2))
        else:
            top = omegas * self._golomb_max_mark()
        return np.arange(-top, top + 1)

    # ---- strategies: each returns ``enc(inputs, wires, **kwargs)`` -----------------------------
    def hamming(self, enc):
        return enc

    def binary(self, enc):
        def scaled(inputs, wires, **kwargs):
            return enc(inputs * (2**wires), wires, **kwargs)

        return scaled

    def ternary(self, enc):
        def scaled(inputs, wires, **kwargs):
            return enc(inputs * (3**wires), wires, **kwargs)

        return scaled

    def golomb(self, enc):
        def joint(inputs, wires, **kwargs):
            Gates.GolombEncoding(w=inputs, wires=wires, **kwargs)

        return joint
