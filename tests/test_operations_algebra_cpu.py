"""Operator algebra of ``operations.Operation`` -- the matrix-level cases of the reference's
``tests/test_jaqsi.py:1384-1618`` (TestGateOperations); no GPU needed."""
import numpy as np
import pytest

from qml_essentials_amd import operations as op
from qml_essentials_amd.tape import recording

X, Y, Z = (np.asarray(c(wires=0, record=False).matrix) for c in (op.PauliX, op.PauliY, op.PauliZ))
I2 = np.eye(2)
CXm = np.asarray(op.CX(wires=[0, 1], record=False).matrix)


def test_scalar_multiplication_both_sides_and_tape():
    x = op.PauliX(wires=0, record=False)
    for r in (x * 2.0, 2.0 * x):
        assert np.allclose(r.matrix, 2.0 * X) and r.wires == [0]
    with recording() as tape:
        op.PauliX(wires=0) * 0.5
    assert len(tape) == 1 and np.allclose(tape[0].matrix, 0.5 * X)


def test_addition():
    x, y, z = (c(wires=0, record=False) for c in (op.PauliX, op.PauliY, op.PauliZ))
    s = x + z
    assert np.allclose(s.matrix, X + Z) and s.wires == [0]
    assert np.allclose(s.matrix, np.conj(s.matrix).T)
    assert np.allclose((x + x).matrix, 2 * X)
    assert np.allclose((x + y).matrix, (y + x).matrix)
    with pytest.raises(ValueError, match="same set of wires"):
        _ = x + op.PauliZ(wires=1, record=False)


def test_matmul_and_prod():
    x0, z1 = op.PauliX(wires=0, record=False), op.PauliZ(wires=1, record=False)
    r = x0 @ z1
    assert np.allclose(r.matrix, np.kron(X, Z)) and r.wires == [0, 1]
    r = x0 @ op.PauliZ(wires=0, record=False)
    assert np.allclose(r.matrix, X @ Z) and r.wires == [0]
    r = op.CX(wires=[0, 1], record=False) @ op.CX(wires=[1, 2], record=False)
    assert np.allclose(r.matrix, np.kron(CXm, I2) @ np.kron(I2, CXm)) and r.wires == [0, 1, 2]
    assert np.allclose((x0 * op.PauliZ(wires=0, record=False)).matrix, X @ Z)
    y1, z0 = op.PauliY(wires=1, record=False), op.PauliZ(wires=0, record=False)
    for r in (op.prod(x0, y1, z0), x0.prod(y1, z0)):
        assert np.allclose(r.matrix, np.kron(X @ Z, Y)) and r.wires == [0, 1]
        assert r.name == "Prod(PauliX*PauliY*PauliZ)"
    r = x0 @ op.Id(wires=1, record=False)
    assert np.allclose(r.matrix, np.kron(X, I2)) and r.wires == [0, 1]
    r = (x0 @ y1) @ op.PauliZ(wires=2, record=False)
    assert np.allclose(r.matrix, np.kron(np.kron(X, Y), Z)) and r.matrix.shape == (8, 8)
    assert r.wires == [0, 1, 2]


def test_dagger_and_power_replace_the_gate_on_the_tape():
    with recording() as tape:
        op.RX(0.5, wires=0)
        op.RX(0.5, wires=0).dagger()
        op.PauliX(wires=0).power(2)
    assert len(tape) == 3
    assert np.allclose(np.asarray(tape[1].matrix) @ np.asarray(tape[0].matrix), I2)
    assert np.allclose(tape[2].matrix, I2)
