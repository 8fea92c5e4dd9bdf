"""Unitary gate vocabulary behind ``Gates.<NAME>`` (noise-free branch).

API mirror of ``qml_essentials/unitary.py:248-701`` (``UnitaryGates``) and
``:18-84`` (``golomb_ruler``).  Each method records the matching
:mod:`operations` class on the active tape.  The reference's ``GateError`` /
``Noise`` hooks (``unitary.py:92-246``) belong to the density-matrix/noise path,
which SURVEY.md section 8-f ranks as a later row: any non-zero ``noise_params`` raises
``NotImplementedError`` instead of silently ignoring the request.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np

from . import operations as op

_RULERS: Dict[int, Tuple[int, ...]] = {}


def golomb_ruler(d: int) -> Tuple[int, ...]:
    """Greedy Golomb ruler with ``d`` marks: all pairwise differences distinct."""
    if d <= 0:
        raise ValueError(f"Golomb ruler order must be positive, got {d}")
    ruler = _RULERS.get(d)
    if ruler is None:
        marks, seen, cand = [0], set(), 1
        while len(marks) < d:
            gaps = {cand - m for m in marks}
            if len(gaps) == len(marks) and not (gaps & seen):
                marks.append(cand)
                seen |= gaps
            cand += 1
        ruler = _RULERS[d] = tuple(marks)
    return ruler


def _reject_noise(noise_params) -> None:
    if noise_params is None:
        return
    active = {k: v for k, v in noise_params.items() if v not in (None, 0, 0.0)}
    if active:
        raise NotImplementedError(
            f"noise channels {sorted(active)} need the density-matrix path, which this "
            "MI355X build does not provide yet (SURVEY.md 8-f rank 3)"
        )


def _plain(cls, n_params: int):
    """Build a ``UnitaryGates`` static method for an operations class."""
    if n_params == 0:
        def gate(wires, noise_params=None, random_key=None):
            _reject_noise(noise_params)
            cls(wires=wires)
    elif n_params == 1:
        def gate(w, wires, noise_params=None, random_key=None):
            _reject_noise(noise_params)
            cls(w, wires=wires)
    else:
        def gate(phi, theta, omega, wires, noise_params=None, random_key=None):
            _reject_noise(noise_params)
            cls(phi, theta, omega, wires=wires)
    gate.__name__ = cls.__name__
    gate.__doc__ = f"Record ``{cls.__name__}`` on the active tape."
    return staticmethod(gate)


class UnitaryGates:
    """Collection of unitary gates; the default backend of :class:`gates.Gates`."""

    batch_gate_error = True  # kept for API parity (script.py:475 cache key); unused here

    Rot = _plain(op.Rot, 3)
    RX, RY, RZ = _plain(op.RX, 1), _plain(op.RY, 1), _plain(op.RZ, 1)
    CRX, CRY, CRZ = _plain(op.CRX, 1), _plain(op.CRY, 1), _plain(op.CRZ, 1)
    RXX, RYY, RZZ, RZX = (_plain(op.RXX, 1), _plain(op.RYY, 1), _plain(op.RZZ, 1),
                          _plain(op.RZX, 1))
    CX, CY, CZ, H = _plain(op.CX, 0), _plain(op.CY, 0), _plain(op.CZ, 0), _plain(op.H, 0)

    @staticmethod
    def CPhase(w, wires, noise_params=None, random_key=None):
        """diag(1,1,1,e^{iw}); ``w = pi`` is CZ (``unitary.py:560-583``)."""
        _reject_noise(noise_params)
        op.ControlledPhaseShift(w, wires=wires)

    @staticmethod
    def PauliRot(theta, pauli, wires, noise_params=None, random_key=None):
        _reject_noise(noise_params)
        op.PauliRot(theta, pauli, wires=wires)

    @staticmethod
    def GolombEncoding(w, wires, noise_params=None, random_key=None):
        """S(x) = exp(-i diag(golomb marks) x) on all ``wires`` (``unitary.py:661-701``)."""
        _reject_noise(noise_params)
        wl = list(wires) if isinstance(wires, (list, tuple)) else [wires]
        marks = np.asarray(golomb_ruler(2 ** len(wl)), dtype=float)
        op.DiagonalQubitUnitary.from_phases(marks, w, wires=wl)
