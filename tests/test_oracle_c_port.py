"""CPU-only: the C/OpenMP restatement (cpu_baseline port) equals the einsum oracle."""
import numpy as np
import pytest

from oracle import c_port, circuits as OC, einsum_sim as OE
from tests.helpers import random_tape


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as g

    g.build()


@pytest.mark.parametrize("n", [1, 2, 5, 10])
def test_c_port_random_circuits(n):
    rng = np.random.default_rng(n)
    tape = random_tape(n, 40, rng, three_q=False)
    got = c_port.simulate(tape, n)
    want = OE.simulate_pure(tape, n, np.complex128)
    assert np.abs(got - want).max() < 2e-6
    w = list(range(n))
    ez = OE.measure_state(want, n, "expval", [("PauliZ", [q]) for q in w])
    assert np.abs(c_port.expval_z(got, n, w) - ez).max() < 1e-6


def test_c_port_model_tape_16q():
    spec = OC.ModelSpec(16, 1, "Hardware_Efficient", data_reupload=False)
    p = np.random.default_rng(1000).uniform(0, 2 * np.pi, spec.params_shape)
    tape = OC.model_tape(spec, p, [0.0])
    a = c_port.simulate(tape, 16)
    b = OE.simulate_pure(tape, 16, np.complex64)
    assert np.abs(a - b).max() < 1e-5
