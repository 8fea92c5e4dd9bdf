"""GPU tests of the device shot sampler: counts bit-exact against
``oracle/sampler.py`` (same Philox stream, fp64 CDF), then the reference's own shot tests
(``tests/test_jaqsi.py:1230-1382``) through ``Script.execute`` and ``Model``."""
import numpy as np
import pytest
import torch

from oracle import sampler as S
from qml_essentials_amd import _native as N
from qml_essentials_amd import jaqsi as js
from qml_essentials_amd import operations as op
from qml_essentials_amd.model import Model
from qml_essentials_amd.script import Script
from qml_essentials_amd.utils import key, key_to_seed

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,B,shots", [(2, 3, 100), (5, 4, 4097), (12, 2, 20001),
                                       (13, 2, 9000), (16, 1, 50000), (1, 5, 7)])
def test_counts_equal_oracle(n, B, shots):
    rng = np.random.default_rng(n * 100 + B)
    p = rng.random((B, 2**n)).astype(np.float32) ** 3
    p[:, rng.integers(0, 2**n, size=max(1, 2**n // 4))] = 0.0      # exact zeros
    p /= p.sum(axis=1, keepdims=True)
    seed, off = 0x0123456789ABCDEF, 1000
    counts, est = N.sample_counts(torch.from_numpy(p).cuda(), shots, seed, off)
    counts, est = counts.cpu().numpy(), est.cpu().numpy()
    for b in range(B):
        want = S.sample_counts(p[b], shots, seed, off + b)
        assert np.array_equal(counts[b], want), (b, np.abs(counts[b] - want).sum())
    assert np.array_equal(counts.sum(axis=1), np.full(B, shots))
    assert np.all(counts[p == 0] == 0)
    assert np.allclose(est, counts / np.float32(shots), atol=1e-7)


def test_diag_expval_kernel():
    rng = np.random.default_rng(0)
    n, B = 6, 5
    p = rng.random((B, 2**n)).astype(np.float32)
    p /= p.sum(axis=1, keepdims=True)
    d2 = rng.normal(size=4)
    specs = [([3], None), ([0, 5, 2], None), ([4, 1], d2), ([2], np.array([0.0, 0.0]))]
    got = N.probs_diag_expval(torch.from_numpy(p).cuda(), specs).cpu().numpy()
    idx = np.arange(2**n)
    bit = lambda w: (idx >> (n - 1 - w)) & 1  # noqa: E731
    want = np.stack([
        p @ (1.0 - 2.0 * bit(3)),
        p @ (1.0 - 2.0 * (bit(0) ^ bit(5) ^ bit(2))),
        p @ d2[2 * bit(4) + bit(1)],
        np.zeros(B)], axis=1)
    assert np.allclose(got, want, atol=1e-6)


def bell(theta):
    op.H(wires=0)
    op.CX(wires=[0, 1])
    op.RZ(theta, wires=0)


def test_script_shots_reference_cases():
    script = Script(bell, n_qubits=2)
    r = script.execute(type="probs", args=(0.5,), shots=4096, key=key(42))
    assert r.shape == (4,) and np.isclose(r.sum(), 1.0, atol=1e-6) and np.all(r >= 0)
    exact = script.execute(type="probs", args=(0.5,))
    sampled = script.execute(type="probs", args=(0.5,), shots=100000, key=key(123))
    assert np.allclose(exact, sampled, atol=0.02)
    # counts are exactly the oracle's for the same key
    want = S.sample_counts(exact.astype(np.float32), 100000, key_to_seed(key(123))) / 100000
    assert np.allclose(sampled, want, atol=1e-7)
    obs = [op.PauliZ(wires=0, record=False), op.PauliZ(wires=1, record=False)]
    e_exact = script.execute(type="expval", obs=obs, args=(0.5,))
    e_shot = script.execute(type="expval", obs=obs, args=(0.5,), shots=100000, key=key(7))
    assert e_shot.shape == e_exact.shape and np.allclose(e_exact, e_shot, atol=0.02)
    k = key(99)
    for _ in range(10):
        k, sub = k.split()
        v = script.execute(type="expval", obs=obs[:1], args=(0.5,), shots=100, key=sub)
        assert -1.0 <= float(v[0]) <= 1.0
    r1 = script.execute(type="probs", args=(0.5,), shots=100, key=key(0))
    r2 = script.execute(type="probs", args=(0.5,), shots=100, key=key(1))
    assert not np.allclose(r1, r2)
    r_default = script.execute(type="probs", args=(0.5,), shots=100)       # key defaults to 0
    assert np.allclose(r_default, r1)
    # state: shots ignored
    st = script.execute(type="state", args=(0.5,))
    st_shots = script.execute(type="state", args=(0.5,), shots=100, key=key(0))
    assert np.allclose(st, st_shots)
    # X observable: basis estimate Tr(O diag(p)) = 0
    x = script.execute(type="expval", obs=[op.PauliX(wires=0, record=False)], args=(0.5,),
                       shots=1000, key=key(3))
    assert np.allclose(x, 0.0)


def test_script_shots_batched_and_chunk_invariant():
    script = Script(bell, n_qubits=2)
    thetas = np.array([0.1, 0.5, 1.0, 2.0])
    r = script.execute(type="probs", args=(thetas,), in_axes=(0,), shots=10000, key=key(42))
    assert r.shape == (4, 4) and np.allclose(r.sum(axis=1), 1.0, atol=1e-6)
    assert not np.allclose(r[0], r[1])           # rows use independent streams
    # row b of a batch == the same sample run alone at row offset b
    from qml_essentials_amd import simulation
    from qml_essentials_amd.tape import recording
    with recording() as tape:
        bell(thetas[2])
    solo = simulation.simulate_and_measure(tape, 2, "probs", shots=10000, key=key(42),
                                           row_offset=2)
    assert np.allclose(solo[0], r[2], atol=1e-7)
    obs = [op.PauliZ(wires=0, record=False)]
    exact = script.execute(type="expval", obs=obs, args=(thetas[:3],), in_axes=(0,))
    sampled = script.execute(type="expval", obs=obs, args=(thetas[:3],), in_axes=(0,),
                             shots=100000, key=key(42))
    assert sampled.shape == exact.shape and np.allclose(exact, sampled, atol=0.02)
    par = [js.build_parity_observable([0, 1])]
    e = script.execute(type="expval", obs=par, args=(thetas,), in_axes=(0,), shots=1000,
                       key=key(5))
    assert np.allclose(e, 1.0)                    # Bell state: parity +1 on every shot


def test_noisy_tape_with_shots():
    def noisy():
        op.H(wires=0)
        op.BitFlip(0.3, wires=0)
        op.CX(wires=[0, 1])
        op.AmplitudeDamping(0.2, wires=1)

    script = Script(noisy, n_qubits=2)
    exact = script.execute(type="probs")
    sampled = script.execute(type="probs", shots=200000, key=key(1))
    assert np.allclose(exact, sampled, atol=0.01) and np.isclose(sampled.sum(), 1.0, atol=1e-6)


@pytest.mark.parametrize("execution_type,output_qubit,shape", [
    ("expval", -1, (3, 3)), ("expval", 0, (3,)), ("expval", [[0, 1]], (3,)),
    ("probs", -1, (3, 2, 2, 2)), ("probs", [0, 1], (3, 2, 2))])
def test_model_shots(execution_type, output_qubit, shape):
    """test_model.py:985-1030 (shots=1024 cases): shapes, and convergence at high shots."""
    model = Model(n_qubits=3, n_layers=1, circuit_type="Circuit_19", output_qubit=output_qubit,
                  shots=1024)
    x = np.array([[0.1], [0.7], [1.3]])
    with pytest.warns() if False else np.errstate():
        got = model(inputs=x, execution_type=execution_type)
    assert got.shape == shape
    model.shots = None
    exact = model(inputs=x, execution_type=execution_type)
    model.shots = 400000
    many = model(inputs=x, execution_type=execution_type)
    assert np.allclose(many, exact, atol=0.01)
    assert not np.allclose(got, exact, atol=1e-6)          # 1024 shots are visibly noisy
    if execution_type == "probs":
        assert np.allclose(got.reshape(3, -1).sum(axis=1), 1.0, atol=1e-6)


@pytest.mark.parametrize("noise", [None, {"BitFlip": 0.02, "Depolarizing": 0.03, "AmplitudeDamping": 0.05}])
@pytest.mark.parametrize("execution_type", ["expval", "probs"])
def test_compiled_call_with_shots_draws_what_the_recorded_path_draws(execution_type, noise):
    """Round 5: shot estimates come out of the compiled call too (exact probabilities of the compiled plan,
    then the device sampler): same key schedule as the per-call recorded path, so two models with the same
    seed return the same draws either way -- bit for bit, call after call, with and without noise channels."""
    fast = Model(n_qubits=4, n_layers=2, circuit_type="Hardware_Efficient", output_qubit=-1, shots=513)
    slow = Model(n_qubits=4, n_layers=2, circuit_type="Hardware_Efficient", output_qubit=-1, shots=513)
    slow.host_arrays_via_device = False
    rng = np.random.default_rng(3)
    P = rng.uniform(0, 2 * np.pi, size=(5, *fast.params.shape[1:]))
    X = rng.uniform(0, 2 * np.pi, size=(2, 1))
    for params, inputs in ((P, X), (P, None), (P[:1], X[:1]), (P, X)):
        kw = dict(noise_params=dict(noise)) if noise else {}
        a = fast(params=params, inputs=inputs, execution_type=execution_type, **kw)
        b = slow(params=params, inputs=inputs, execution_type=execution_type, **kw)
        assert a.shape == b.shape and np.array_equal(a, b)
    assert fast.script._compiled and not slow.script._compiled
    fast.shots = slow.shots = None
    exact = slow(params=P, inputs=X, execution_type=execution_type)
    assert not np.allclose(a, exact, atol=1e-6)
