"""Unitary gate vocabulary behind ``Gates.<NAME>``.

API mirror of ``qml_essentials/unitary.py:248-701`` (``UnitaryGates``) and
``:18-84`` (``golomb_ruler``).  Each method records the matching
:mod:`operations` class on the active tape, preceded by the coherent ``GateError``
(Gaussian angle noise) and followed by the ``Noise`` channels named in ``noise_params``
(``unitary.py:92-246``).  A recorded channel puts the tape on the density-matrix path
(``simulation.py``).  Random draws use :mod:`utils` keys (Philox, not threefry).
"""
from __future__ import annotations

import itertools
from typing import Dict, Optional, Tuple

import numpy as np

from . import operations as op
from .batching import Batched
from .tape import current_batch
from .utils import as_key, safe_random_split

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


def _n_qubit_depolarizing_kraus(p: float, n: int):
    """K0 = sqrt(1 - p (4^n - 1)/4^n) I and sqrt(p / 4^n) P for the 4^n - 1 non-identity
    Pauli words (``unitary.py:116-148``)."""
    if not (0.0 <= p <= 1.0):
        raise ValueError(f"Probability p must be between 0 and 1, got {p}")
    if n < 2:
        raise ValueError(f"Number of qubits must be >= 2, got {n}")
    paulis = [np.eye(2, dtype=np.complex128),
              np.array([[0, 1], [1, 0]], dtype=np.complex128),
              np.array([[0, -1j], [1j, 0]], dtype=np.complex128),
              np.array([[1, 0], [0, -1]], dtype=np.complex128)]
    out = [np.sqrt(1 - p * (4**n - 1) / (4**n)) * np.eye(2**n, dtype=np.complex128)]
    for k, idx in enumerate(itertools.product(range(4), repeat=n)):
        if k == 0:
            continue
        P = np.eye(1, dtype=np.complex128)
        for i in idx:
            P = np.kron(P, paulis[i])
        out.append(np.sqrt(p / (4**n)) * P)
    return out


def _plain(cls, n_params: int):
    """Build a ``UnitaryGates`` static method for an operations class: GateError on the
    angle(s), the gate itself, then the noise channels on its wires."""
    if n_params == 0:
        def gate(wires, noise_params=None, random_key=None):
            cls(wires=wires)
            UnitaryGates.Noise(wires, noise_params)
    elif n_params == 1:
        def gate(w, wires, noise_params=None, random_key=None):
            w, random_key = UnitaryGates.GateError(w, noise_params, random_key)
            cls(w, wires=wires)
            UnitaryGates.Noise(wires, noise_params)
    else:
        def gate(phi, theta, omega, wires, noise_params=None, random_key=None):
            if noise_params is not None and "GateError" in noise_params:
                phi, random_key = UnitaryGates.GateError(phi, noise_params, random_key)
                theta, random_key = UnitaryGates.GateError(theta, noise_params, random_key)
                omega, random_key = UnitaryGates.GateError(omega, noise_params, random_key)
            cls(phi, theta, omega, wires=wires)
            UnitaryGates.Noise(wires, noise_params)
    gate.__name__ = cls.__name__
    gate.__doc__ = f"Record ``{cls.__name__}`` (and its noise) on the active tape."
    return staticmethod(gate)


class UnitaryGates:
    """Collection of unitary gates; the default backend of :class:`gates.Gates`."""

    batch_gate_error = True  # True: every batch element draws its own angle error

    @staticmethod
    def NQubitDepolarizingChannel(p: float, wires) -> op.QubitChannel:
        """n-qubit depolarizing channel as a ``QubitChannel`` (``unitary.py:92-148``)."""
        return op.QubitChannel(_n_qubit_depolarizing_kraus(p, len(wires)), wires=wires)

    @staticmethod
    def Noise(wires, noise_params: Optional[Dict[str, float]] = None) -> None:
        """BitFlip / PhaseFlip / Depolarizing on every wire, MultiQubitDepolarizing after a
        multi-wire gate (``unitary.py:150-197``)."""
        if noise_params is None:
            return
        wl = [wires] if isinstance(wires, (int, np.integer)) else list(wires)
        for w in wl:
            bf = noise_params.get("BitFlip", 0.0)
            if bf > 0:
                op.BitFlip(bf, wires=w)
            pf = noise_params.get("PhaseFlip", 0.0)
            if pf > 0:
                op.PhaseFlip(pf, wires=w)
            dp = noise_params.get("Depolarizing", 0.0)
            if dp > 0:
                op.DepolarizingChannel(dp, wires=w)
        if len(wl) > 1:
            p = noise_params.get("MultiQubitDepolarizing", 0.0)
            if p > 0:
                UnitaryGates.NQubitDepolarizingChannel(p, wl)

    @staticmethod
    def GateError(w, noise_params: Optional[Dict[str, float]] = None, random_key=None):
        """``w + sigma N(0, 1)``, ``sigma = noise_params["GateError"]`` (``unitary.py:199-246``):
        one draw per batch element, or a single fixed-key draw shared by the batch when
        ``batch_gate_error`` is False.  Returns ``(w, random_key)``."""
        if noise_params is None or noise_params.get("GateError", None) is None:
            return w, random_key
        assert random_key is not None, "A random_key must be provided when using GateError"
        sigma = float(noise_params["GateError"])
        if sigma == 0.0:
            return w, random_key
        if UnitaryGates.batch_gate_error:
            random_key, sub = safe_random_split(random_key)
            B = w.batch if isinstance(w, Batched) else current_batch()
        else:
            sub, B = as_key(0), 1
        noise = as_key(sub).generator().standard_normal(B) * sigma
        if isinstance(w, Batched):
            col = noise if B == w.batch else np.full(w.batch, noise[0])
            return w + Batched(col, []), random_key
        if B > 1:
            return Batched(float(np.asarray(w, dtype=np.float64)) + noise, []), random_key
        return float(np.asarray(w, dtype=np.float64)) + float(noise[0]), random_key

    Rot = _plain(op.Rot, 3)
    RX, RY, RZ = _plain(op.RX, 1), _plain(op.RY, 1), _plain(op.RZ, 1)
    CRX, CRY, CRZ = _plain(op.CRX, 1), _plain(op.CRY, 1), _plain(op.CRZ, 1)
    RXX, RYY, RZZ, RZX = (_plain(op.RXX, 1), _plain(op.RYY, 1), _plain(op.RZZ, 1),
                          _plain(op.RZX, 1))
    CX, CY, CZ, H = _plain(op.CX, 0), _plain(op.CY, 0), _plain(op.CZ, 0), _plain(op.H, 0)

    @staticmethod
    def CPhase(w, wires, noise_params=None, random_key=None):
        """diag(1,1,1,e^{iw}); ``w = pi`` is CZ (``unitary.py:560-583``)."""
        w, random_key = UnitaryGates.GateError(w, noise_params, random_key)
        op.ControlledPhaseShift(w, wires=wires)
        UnitaryGates.Noise(wires, noise_params)

    @staticmethod
    def PauliRot(theta, pauli, wires, noise_params=None, random_key=None):
        if noise_params is not None and "GateError" in noise_params:
            theta, random_key = UnitaryGates.GateError(theta, noise_params, random_key)
        op.PauliRot(theta, pauli, wires=wires)
        UnitaryGates.Noise(wires, noise_params)

    @staticmethod
    def GolombEncoding(w, wires, noise_params=None, random_key=None):
        """S(x) = exp(-i diag(golomb marks) x) on all ``wires`` (``unitary.py:661-701``)."""
        wl = list(wires) if isinstance(wires, (list, tuple)) else [wires]
        marks = np.asarray(golomb_ruler(2 ** len(wl)), dtype=float)
        w, random_key = UnitaryGates.GateError(w, noise_params, random_key)
        op.DiagonalQubitUnitary.from_phases(marks, w, wires=wl)
        UnitaryGates.Noise(wl, noise_params)
