"""Gate matrices of the reference, restated in NumPy (oracle; test infrastructure).

Every function cites the line of ``/root/reference/qml_essentials/operations.py``
it follows.  Matrices are returned in complex128; the simulator casts to the
working dtype.  A *tape entry* of the oracle is ``(name, wires, params)`` with
``params`` a tuple of floats (or a matrix / diagonal for the generic kinds).
"""
from functools import reduce

import numpy as np

I2 = np.eye(2, dtype=np.complex128)
X = np.array([[0, 1], [1, 0]], dtype=np.complex128)  # operations.py:749
Y = np.array([[0, -1j], [1j, 0]], dtype=np.complex128)  # operations.py:765
Z = np.array([[1, 0], [0, -1]], dtype=np.complex128)  # operations.py:781
HAD = np.array([[1, 1], [1, -1]], dtype=np.complex128) / np.sqrt(2)  # :797
S_GATE = np.array([[1, 0], [0, 1j]], dtype=np.complex128)  # :819
P0 = np.array([[1, 0], [0, 0]], dtype=np.complex128)  # :1049
P1 = np.array([[0, 0], [0, 1]], dtype=np.complex128)  # :1050
PAULI = {"I": I2, "X": X, "Y": Y, "Z": Z}  # :994-999


def rot_pauli(theta, P):
    """cos(t/2) I - i sin(t/2) P  (operations.py:1029-1031)."""
    return np.cos(theta / 2) * np.eye(P.shape[0]) - 1j * np.sin(theta / 2) * P


def pauli_word(word):
    """Kronecker product of single-qubit Paulis (operations.py:1291-1292)."""
    return reduce(np.kron, [PAULI[c] for c in word])


def controlled(U):
    """|0><0| (x) I + |1><1| (x) U  (operations.py:1074)."""
    return np.kron(P0, np.eye(U.shape[0])) + np.kron(P1, U)


def controlled_pauli_rot(theta, word, n_controls=1):
    """Identity with R_P(theta) in the last block (operations.py:1388-1411)."""
    R = rot_pauli(theta, pauli_word(word))
    d_t = R.shape[0]
    d_c = 2**n_controls
    mat = np.eye(d_c * d_t, dtype=np.complex128)
    start = (d_c - 1) * d_t
    mat[start:, start:] = R
    return mat


def rot(phi, theta, omega):
    """Rot = RZ(omega) @ RY(theta) @ RZ(phi)  (operations.py:1234-1243)."""
    return rot_pauli(omega, Z) @ rot_pauli(theta, Y) @ rot_pauli(phi, Z)


def cphase(phi):
    """diag(1,1,1,e^{i phi})  (operations.py:1199-1200)."""
    return controlled(np.array([[1, 0], [0, np.exp(1j * phi)]], dtype=np.complex128))


CCX = np.eye(8, dtype=np.complex128)  # operations.py:1112-1124
CCX[6:, 6:] = X
CSWAP = np.eye(8, dtype=np.complex128)  # operations.py:1149-1161
CSWAP[5:7, 5:7] = X
SWAP = np.array(  # operations.py:835-837
    [[1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=np.complex128
)


def matrix(name, params=()):
    """Matrix of tape entry ``name(*params)`` in the reference's wire order."""
    p = params
    if name == "Id":
        return I2
    if name == "PauliX":
        return X
    if name == "PauliY":
        return Y
    if name == "PauliZ":
        return Z
    if name == "H":
        return HAD
    if name == "S":
        return S_GATE
    if name == "SWAP":
        return SWAP
    if name == "RX":  # operations.py:1043
        return rot_pauli(p[0], X)
    if name == "RY":  # :1044
        return rot_pauli(p[0], Y)
    if name == "RZ":  # :1045
        return rot_pauli(p[0], Z)
    if name == "CX":  # :1098
        return controlled(X)
    if name == "CY":  # :1099
        return controlled(Y)
    if name == "CZ":  # :1100
        return controlled(Z)
    if name == "CCX":
        return CCX
    if name == "CSWAP":
        return CSWAP
    if name == "CRX":  # :1485
        return controlled_pauli_rot(p[0], "X")
    if name == "CRY":  # :1486
        return controlled_pauli_rot(p[0], "Y")
    if name == "CRZ":  # :1487
        return controlled_pauli_rot(p[0], "Z")
    if name == "CPhase":  # ControlledPhaseShift, :1171-1201
        return cphase(p[0])
    if name == "Rot":
        return rot(p[0], p[1], p[2])
    if name in ("RXX", "RYY", "RZZ", "RZX"):  # :1348-1351
        return rot_pauli(p[0], pauli_word(name[1:]))
    if name == "PauliRot":  # params = (theta, word)
        return rot_pauli(p[0], pauli_word(p[1]))
    if name == "Matrix":  # generic Operation(matrix=U)
        return np.asarray(p[0], dtype=np.complex128)
    if name == "DiagU":  # DiagonalQubitUnitary, operations.py:881-909
        return np.diag(np.asarray(p[0], dtype=np.complex128))
    raise KeyError(f"oracle: unknown gate {name!r}")


def golomb_ruler(d):
    """Greedy Golomb ruler of order d  (unitary.py:18-84)."""
    marks, diffs, cand = [0], set(), 1
    while len(marks) < d:
        new = set()
        ok = True
        for m in marks:
            df = cand - m
            if df in diffs or df in new:
                ok = False
                break
            new.add(df)
        if ok:
            marks.append(cand)
            diffs |= new
        cand += 1
    return tuple(marks[:d]) if d > 0 else ()


def golomb_diag(x, n_wires):
    """exp(-i * marks * x) for the Golomb encoding (unitary.py:690-698)."""
    marks = np.array(golomb_ruler(2**n_wires), dtype=float)
    return np.exp(-1j * marks * x)
