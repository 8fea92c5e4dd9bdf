"""Dense 2^n x 2^n cross-check oracle (independent of the einsum restatement).

TEST INFRASTRUCTURE ONLY.  Builds the full unitary of each gate by explicit bit
arithmetic on basis-state indices with wire 0 = most significant bit
(``qml_essentials/simulation.py:100-104`` reshapes the flat state to ``(2,)*n``
so axis i = wire i; pinned by ``tests/test_jaqsi.py:416-427``: H(0), CX[0,2] on
3 qubits -> indices 0 and 5).  Matrix row/col index of a k-qubit gate is
``sum_j bit[wires[j]] << (k-1-j)`` (``operations.py:44-49``).  Usable to n ~ 10.
"""
import numpy as np

from . import gates as G


def lift(mat, wires, n_qubits):
    """Full-space matrix of ``mat`` acting on ``wires`` (wire 0 = MSB)."""
    k = len(wires)
    dim = 2**n_qubits
    full = np.zeros((dim, dim), dtype=np.complex128)
    shifts = [n_qubits - 1 - w for w in wires]
    rest_mask = (dim - 1) & ~sum(1 << s for s in shifts)
    for col in range(dim):
        sub_in = 0
        for j, s in enumerate(shifts):
            sub_in |= ((col >> s) & 1) << (k - 1 - j)
        base = col & rest_mask
        for sub_out in range(2**k):
            amp = mat[sub_out, sub_in]
            if amp == 0:
                continue
            row = base
            for j, s in enumerate(shifts):
                row |= ((sub_out >> (k - 1 - j)) & 1) << s
            full[row, col] += amp
    return full


def circuit_unitary(tape, n_qubits):
    U = np.eye(2**n_qubits, dtype=np.complex128)
    for name, wires, params in tape:
        if name == "Barrier":
            continue
        U = lift(G.matrix(name, params), list(wires), n_qubits) @ U
    return U


def simulate(tape, n_qubits):
    psi = np.zeros(2**n_qubits, dtype=np.complex128)
    psi[0] = 1.0
    for name, wires, params in tape:
        if name == "Barrier":
            continue
        psi = lift(G.matrix(name, params), list(wires), n_qubits) @ psi
    return psi


def expval_z(psi, n_qubits, wire):
    idx = np.arange(psi.size)
    sign = 1 - 2 * ((idx >> (n_qubits - 1 - wire)) & 1)
    return float(np.sum(sign * np.abs(psi) ** 2))
