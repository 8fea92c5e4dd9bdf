"""Wire-pair generators for entangling blocks.

Behavioural mirror of ``qml_essentials/topologies.py:21-121`` (same keyword
surface and defaults).  Outputs are pinned against the real reference module by
``tests/golden/topologies.json``.  Pairs are ``(control, target)``.
"""
from __future__ import annotations

import logging
from typing import Callable, List, Tuple, Union

log = logging.getLogger(__name__)

IntOrFn = Union[int, Callable[[int], int]]


def _resolve(value: IntOrFn, n_qubits: int) -> int:
    return value(n_qubits) if callable(value) else value


class Topology:
    """Static generators; every ansatz block names one of them."""

    @classmethod
    def stairs(cls, n_qubits: int, offset: IntOrFn = 0, wrap: bool = False, reverse: bool = True,
               mirror: bool = True, span: IntOrFn = 1, stride: int = 1,
               modulo: bool = True) -> List[Tuple[int, int]]:
        """Ladder of pairs ``(q+offset, q+offset+span)`` for q = 0, stride, 2*stride, ...

        ``wrap`` adds the pair that closes the ring, ``modulo=False`` drops pairs that
        would leave the register instead of wrapping them, ``reverse`` flips the
        emission order and ``mirror`` swaps control and target.
        """
        shift, reach = _resolve(offset, n_qubits), _resolve(span, n_qubits)
        count = n_qubits if wrap else n_qubits - 1
        ladder: List[Tuple[int, int]] = []
        for q in range(0, count, stride):
            lo, hi = q + shift, q + shift + reach
            if not modulo and (hi >= n_qubits or lo < 0):
                continue
            lo, hi = lo % n_qubits, hi % n_qubits
            if lo == hi:
                log.warning("Skipping gate where control == target")
                continue
            ladder.append((lo, hi))
        if reverse:
            ladder.reverse()
        if mirror:
            ladder = [(b, a) for a, b in ladder]
        return ladder

    @classmethod
    def bricks(cls, n_qubits: int, **kwargs) -> List[Tuple[int, int]]:
        """Every second rung of :meth:`stairs`, without wrap-around by default."""
        opts = {"stride": 2, "modulo": False}
        opts.update(kwargs)
        return cls.stairs(n_qubits=n_qubits, **opts)

    @classmethod
    def all_to_all(cls, n_qubits: int) -> List[List[int]]:
        """Every ordered pair of distinct wires, highest control first."""
        top = n_qubits - 1
        return [[top - a, top - b] for a in range(n_qubits) for b in range(n_qubits) if a != b]
