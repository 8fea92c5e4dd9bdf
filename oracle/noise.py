"""NumPy restatement of the reference's density-matrix (noise) path (oracle).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Parity pinned by the closed-form
known answers of the reference's own tests (``tests/test_jaqsi.py:587-696``): trace
preservation, BitFlip/PhaseFlip/Depolarizing/damping expectation values.

Follows:
  * ``qml_essentials/simulation.py:106-128``   simulate_mixed
  * ``qml_essentials/simulation.py:274-317``   measure_density
  * ``qml_essentials/operations.py:485-512``   Operation.apply_to_density (U on the ket axes,
    conj(U) on the bra axes of rho viewed as a rank-2n tensor)
  * ``qml_essentials/operations.py:1552-1578`` KrausChannel.apply_to_density (sum over K)
  * ``qml_essentials/operations.py:1583-1929`` channel Kraus operators
  * ``qml_essentials/unitary.py:92-148``       NQubitDepolarizingChannel
  * ``qml_essentials/model.py:1000-1064``      state-prep / general noise placement

A tape entry is ``(name, wires, params)`` as in :mod:`einsum_sim`; channel entries use the
names below with ``params`` = the channel's constructor arguments (``"QubitChannel"``:
``params = (list_of_kraus,)``).
"""
import itertools

import numpy as np

from . import gates as G
from .einsum_sim import einsum_subscript

I2 = np.eye(2, dtype=np.complex128)
X = np.array([[0, 1], [1, 0]], dtype=np.complex128)
Y = np.array([[0, -1j], [1j, 0]], dtype=np.complex128)
Z = np.array([[1, 0], [0, -1]], dtype=np.complex128)


def thermal_relaxation_kraus(pe, t1, t2, tg):
    """operations.py:1851-1893."""
    eT1, eT2 = np.exp(-tg / t1), np.exp(-tg / t2)
    p_reset = 1.0 - eT1
    if t2 <= t1:
        pz = (1.0 - p_reset) * (1.0 - eT2 / eT1) / 2.0
        pr0, pr1 = (1.0 - pe) * p_reset, pe * p_reset
        pid = 1.0 - pz - pr0 - pr1
        return [np.sqrt(pid) * I2, np.sqrt(pz) * Z,
                np.sqrt(pr0) * np.array([[1, 0], [0, 0]], dtype=np.complex128),
                np.sqrt(pr0) * np.array([[0, 1], [0, 0]], dtype=np.complex128),
                np.sqrt(pr1) * np.array([[0, 0], [1, 0]], dtype=np.complex128),
                np.sqrt(pr1) * np.array([[0, 0], [0, 1]], dtype=np.complex128)]
    choi = np.array([[1 - pe * p_reset, 0, 0, eT2], [0, pe * p_reset, 0, 0],
                     [0, 0, (1 - pe) * p_reset, 0], [eT2, 0, 0, 1 - (1 - pe) * p_reset]],
                    dtype=np.complex128)
    lam, vec = np.linalg.eigh(choi)
    return [np.sqrt(np.abs(lam[i])) * vec[:, i].reshape(2, 2, order="F") for i in range(4)]


def n_qubit_depolarizing_kraus(p, n):
    """unitary.py:116-148."""
    paulis = [I2, X, Y, Z]
    words = list(itertools.product(range(4), repeat=n))
    out = [np.sqrt(1 - p * (4**n - 1) / 4**n) * np.eye(2**n, dtype=np.complex128)]
    for word in words[1:]:
        P = np.eye(1, dtype=np.complex128)
        for i in word:
            P = np.kron(P, paulis[i])
        out.append(np.sqrt(p / 4**n) * P)
    return out


def kraus(name, params):
    """Kraus operators per channel class (operations.py:1583-1929)."""
    if name == "BitFlip":
        (p,) = params
        return [np.sqrt(1 - p) * I2, np.sqrt(p) * X]
    if name == "PhaseFlip":
        (p,) = params
        return [np.sqrt(1 - p) * I2, np.sqrt(p) * Z]
    if name == "DepolarizingChannel":
        (p,) = params
        return [np.sqrt(1 - p) * I2, np.sqrt(p / 3) * X, np.sqrt(p / 3) * Y, np.sqrt(p / 3) * Z]
    if name == "AmplitudeDamping":
        (g,) = params
        return [np.array([[1, 0], [0, np.sqrt(1 - g)]], dtype=np.complex128),
                np.array([[0, np.sqrt(g)], [0, 0]], dtype=np.complex128)]
    if name == "PhaseDamping":
        (g,) = params
        return [np.array([[1, 0], [0, np.sqrt(1 - g)]], dtype=np.complex128),
                np.array([[0, 0], [0, np.sqrt(g)]], dtype=np.complex128)]
    if name == "ThermalRelaxationError":
        return thermal_relaxation_kraus(*params)
    if name == "QubitChannel":
        return [np.asarray(k, dtype=np.complex128) for k in params[0]]
    if name == "NQubitDepolarizing":
        p, n = params
        return n_qubit_depolarizing_kraus(p, n)
    return None


CHANNELS = ("BitFlip", "PhaseFlip", "DepolarizingChannel", "AmplitudeDamping", "PhaseDamping",
            "ThermalRelaxationError", "QubitChannel", "NQubitDepolarizing")


def _contract(rho_t, gate_t, k, axes, total):
    """operations.py:53-98 (_contract_and_restore) as one einsum."""
    return np.einsum(einsum_subscript(total, k, axes), gate_t, rho_t)


def apply_to_density(rho, n_qubits, name, wires, params):
    """rho -> U rho U^+ (operations.py:485-512) or sum_k K rho K^+ (:1552-1578)."""
    k = len(wires)
    dim = 2**n_qubits
    bra = [w + n_qubits for w in wires]
    ks = kraus(name, params)
    if ks is None:
        if name == "DiagU" and list(wires) == list(range(n_qubits)):  # operations.py:944-961
            d = np.asarray(params[0], dtype=np.complex128)
            return d[:, None] * np.conj(d)[None, :] * rho
        ks = [np.asarray(G.matrix(name, params), dtype=np.complex128)]
    out = np.zeros_like(rho)
    for K in ks:
        Kt = K.reshape((2,) * (2 * k))
        rt = rho.reshape((2,) * (2 * n_qubits))
        rt = _contract(rt, Kt, k, list(wires), 2 * n_qubits)
        rt = _contract(rt, np.conj(Kt), k, bra, 2 * n_qubits)
        out = out + rt.reshape(dim, dim)
    return out


def simulate_mixed(tape, n_qubits, dtype=np.complex128):
    """simulation.py:106-128."""
    dim = 2**n_qubits
    rho = np.zeros((dim, dim), dtype=dtype)
    rho[0, 0] = 1.0
    for name, wires, params in tape:
        if name == "Barrier":
            continue
        rho = apply_to_density(rho, n_qubits, name, wires, params).astype(dtype)
    return rho


def measure_density(rho, n_qubits, type, obs=()):
    """simulation.py:274-317.  ``obs`` = list of dense (2^n x 2^n) observable matrices."""
    if type == "density":
        return rho
    if type == "probs":
        return np.real(np.diag(rho))
    if type == "expval":
        return np.array([np.real(np.einsum("ij,ji->", O, rho)) for O in obs])
    raise ValueError(
        "Measurement type 'state' is not defined for mixed (noisy) circuits. "
        "Use 'density' instead."
    )


def with_gate_noise(tape, noise):
    """Insert the per-gate channels of ``UnitaryGates.Noise`` (unitary.py:150-197) after every
    gate of a noise-free tape, state-prep noise before and general noise after it
    (model.py:1000-1064).  ``noise`` may hold BitFlip, PhaseFlip, Depolarizing,
    MultiQubitDepolarizing, StatePreparation, AmplitudeDamping, PhaseDamping, Measurement,
    ThermalRelaxation=(t1, t2, tg)."""
    n = 1 + max(w for _, ws, _ in tape for w in ws)
    out = []
    if noise.get("StatePreparation", 0) > 0:
        out += [("BitFlip", [q], (noise["StatePreparation"],)) for q in range(n)]
    for name, wires, params in tape:
        out.append((name, wires, params))
        if name == "Barrier":
            continue
        for w in wires:
            if noise.get("BitFlip", 0) > 0:
                out.append(("BitFlip", [w], (noise["BitFlip"],)))
            if noise.get("PhaseFlip", 0) > 0:
                out.append(("PhaseFlip", [w], (noise["PhaseFlip"],)))
            if noise.get("Depolarizing", 0) > 0:
                out.append(("DepolarizingChannel", [w], (noise["Depolarizing"],)))
        if len(wires) > 1 and noise.get("MultiQubitDepolarizing", 0) > 0:
            out.append(("NQubitDepolarizing", list(wires),
                        (noise["MultiQubitDepolarizing"], len(wires))))
    for q in range(n):
        if noise.get("AmplitudeDamping", 0) > 0:
            out.append(("AmplitudeDamping", [q], (noise["AmplitudeDamping"],)))
        if noise.get("PhaseDamping", 0) > 0:
            out.append(("PhaseDamping", [q], (noise["PhaseDamping"],)))
        if noise.get("Measurement", 0) > 0:
            out.append(("BitFlip", [q], (noise["Measurement"],)))
        if noise.get("ThermalRelaxation") is not None:
            t1, t2, tg = noise["ThermalRelaxation"]
            out.append(("ThermalRelaxationError", [q], (1.0, t1, t2, tg)))
    return out
