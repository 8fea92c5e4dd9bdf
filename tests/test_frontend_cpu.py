"""CPU-only: the Python front-end records exactly the tape the reference would
(checked gate by gate against ``oracle/circuits.py``), plus batching semantics."""
import json
import os

import numpy as np
import pytest

from oracle import circuits as OC
from qml_essentials_amd import operations as op
from qml_essentials_amd.ansaetze import Ansaetze, Encoding
from qml_essentials_amd.batching import Batched
from qml_essentials_amd.gates import Gates
from qml_essentials_amd.model import Model
from qml_essentials_amd.script import Script
from qml_essentials_amd.tape import copy_to_tape, recording
from qml_essentials_amd.topologies import Topology

ANSAETZE = sorted(OC.STRUCTURES) + ["GHZ"]


def tape_tuples(tape, sample=None):
    out = []
    for o in tape:
        ps = []
        for p in o.parameters:
            ps.append(float(p if np.ndim(p) == 0 else p[sample]))
        name = {"ControlledPhaseShift": "CPhase"}.get(o.name, o.name)
        out.append((name, list(o.wires), tuple(ps)))
    return out


def same_tape(got, want, atol=1e-6):
    assert len(got) == len(want), (len(got), len(want))
    for g, w in zip(got, want):
        assert g[0] == w[0] and g[1] == list(w[1]), (g, w)
        assert np.allclose(g[2], w[2], atol=atol), (g, w)


def test_topology_matches_reference_fixture(golden_dir):
    fx = json.load(open(os.path.join(golden_dir, "topologies.json")))
    sym = {"n-1": lambda n: n - 1, "n-2": lambda n: n - 2, "n//2": lambda n: n // 2}
    for key, rec in fx["calls"].items():
        kw = {k: (sym[v] if isinstance(v, str) else v) for k, v in rec["kwargs"].items()}
        for n_s, want in rec["pairs"].items():
            got = [list(p) for p in getattr(Topology, rec["topology"])(n_qubits=int(n_s), **kw)]
            assert got == want, (key, n_s)
    assert Topology.all_to_all(3) == [[2, 1], [2, 0], [1, 2], [1, 0], [0, 2], [0, 1]]


@pytest.mark.parametrize("ansatz", ANSAETZE)
@pytest.mark.parametrize("n", [2, 3, 4, 5, 6])
def test_model_tape_equals_oracle_tape(ansatz, n):
    rng = np.random.default_rng(n)
    for dru in (True, False):
        m = Model(n_qubits=n, n_layers=2, circuit_type=ansatz, data_reupload=dru)
        spec = OC.ModelSpec(n, 2, ansatz, data_reupload=dru)
        assert m.params.shape[1:] == spec.params_shape
        assert m.degree == spec.degree and m.has_dru == spec.has_dru
        params = rng.uniform(0, 2 * np.pi, spec.params_shape)
        for x in (0.0, 0.7):
            tape, B = m.record_tape(params=params, inputs=np.array([x]))
            assert B == 1
            same_tape(tape_tuples(tape), OC.model_tape(spec, params, [x]))


def test_param_counts_match_reference_tests():
    # tests/test_ansaetze.py:184-223 -- controlled-rotation slices at 4 qubits
    for name, k in (("Circuit_3", 3), ("Circuit_4", 3), ("Circuit_16", 3), ("Circuit_17", 3),
                    ("Circuit_18", 4), ("Circuit_19", 4)):
        assert getattr(Ansaetze, name).get_control_indices(4) == [-k, None, None]
    assert Ansaetze.Circuit_1.get_control_indices(4) is None
    assert len(Ansaetze.get_available()) == 23
    assert len(Ansaetze.get_available(parameterized_only=True)) == 21
    m = Model(4, 1, "Circuit_19", initialization="zero-controlled")
    assert np.all(m.params[:, :, -4:] == 0) and np.any(m.params[:, :, :-4] != 0)
    m = Model(4, 1, "Circuit_19", initialization="pi-controlled")
    assert np.allclose(m.params[:, :, -4:], np.pi)
    assert np.all(Model(3, 1, "Circuit_1", initialization="zeros").params == 0)


def test_encodings_and_degrees():
    # tests/test_model.py:240-324, 797-836
    assert Model(2, 1, "Circuit_19").degree == (5,)
    assert Model(2, 1, "Circuit_19", encoding=["RX", "RY"]).degree == (5, 5)
    assert Model(2, 1, "Circuit_19", encoding=Encoding("binary", "RX")).degree == (7,)
    assert Model(2, 1, "Circuit_19", encoding=Encoding("ternary", "RX")).degree == (9,)
    assert Model(2, 1, "Circuit_19", encoding=Encoding("ternary", ["RX", "RY"])).degree == (9, 9)
    assert Model(1, 1, "Circuit_19", data_reupload=False).degree == (3,)
    for strat in ("binary", "ternary"):
        m = Model(3, 2, "Circuit_1", encoding=Encoding(strat, "RY"))
        spec = OC.ModelSpec(3, 2, "Circuit_1", encoding="RY", strategy=strat)
        p = np.random.default_rng(0).uniform(0, 6, spec.params_shape)
        tape, _ = m.record_tape(params=p, inputs=np.array([0.3]))
        same_tape(tape_tuples(tape), OC.model_tape(spec, p, [0.3]))
    # multi-feature + state preparation + custom mask
    mask = np.array([[1, 0, 1], [0, 1, 1]], dtype=bool)
    m = Model(3, 2, "Hardware_Efficient", encoding=["RX", "RZ"], state_preparation="H",
              data_reupload=mask)
    spec = OC.ModelSpec(3, 2, "Hardware_Efficient", encoding=["RX", "RZ"], state_preparation="H",
                        data_reupload=mask)
    p = np.random.default_rng(1).uniform(0, 6, spec.params_shape)
    tape, _ = m.record_tape(params=p, inputs=np.array([[0.3, 1.1]]))
    same_tape(tape_tuples(tape), OC.model_tape(spec, p, [0.3, 1.1]))


def test_golomb_encoding_tape():
    m = Model(2, 1, "Circuit_1", encoding=Encoding("golomb", None))
    spec = OC.ModelSpec(2, 1, "Circuit_1", strategy="golomb")
    assert m.degree == spec.degree  # tests/test_model.py:391-400
    p = np.random.default_rng(2).uniform(0, 6, spec.params_shape)
    tape, _ = m.record_tape(params=p, inputs=np.array([0.4]))
    diag_ops = [o for o in tape if o.name == "DiagU"]
    want = [g for g in OC.model_tape(spec, p, [0.4]) if g[0] == "DiagU"]
    assert len(diag_ops) == len(want) == 1
    assert np.allclose(diag_ops[0].diag, want[0][2][0])
    from qml_essentials_amd.unitary import golomb_ruler
    from oracle.gates import golomb_ruler as oracle_ruler
    for d in (1, 2, 4, 8, 16):  # tests/test_model.py:328-357 validity
        r = golomb_ruler(d)
        assert r == oracle_ruler(d)
        diffs = [b - a for i, a in enumerate(r) for b in r[i + 1:]]
        assert len(diffs) == len(set(diffs))


def test_batch_semantics_inputs_slowest():
    """model.py:1449-1481: B = B_I * B_P, inputs slowest; one tape for the batch."""
    m = Model(3, 1, "Circuit_19")
    spec = OC.ModelSpec(3, 1, "Circuit_19")
    rng = np.random.default_rng(4)
    P = rng.uniform(0, 6, (4, *spec.params_shape))
    X = rng.uniform(0, 3, (5, 1))
    tape, B = m.record_tape(params=P, inputs=X)
    assert B == 20 and m.batch_shape == (5, 4, 1)
    for b in (0, 3, 4, 19):
        same_tape(tape_tuples(tape, b), OC.model_tape(spec, P[b % 4], X[b // 4],
                                                      zero_inputs_batch1=False))
    # repeat_batch_axis=[False, True, True] zips inputs with params (test_model.py:134-150)
    m2 = Model(2, 1, "Circuit_19", repeat_batch_axis=[False, True, True])
    P2 = rng.uniform(0, 6, (10, *m2.params.shape[1:]))
    tape, B = m2.record_tape(params=P2, inputs=rng.uniform(0, 1, (10, 1)))
    assert B == 10 and tuple(m2.eff_batch_shape) == (10, 1)


def test_zero_input_removes_encoding_only_for_batch1():
    m = Model(2, 1, "Circuit_19")
    n_enc = lambda t: sum(1 for o in t if o.name == "RX" and np.ndim(o.theta) == 0 and o.theta == 0)
    tape, _ = m.record_tape(inputs=None)
    assert sum(1 for o in tape if o.name not in ("Barrier",)) == 2 * (2 + 2 + 2)
    m2 = Model(2, 1, "Circuit_19", remove_zero_encoding=False)
    tape2, _ = m2.record_tape(inputs=None)
    assert len(tape2) == len(tape) + 2


def test_gates_router_and_validation():
    assert Gates.RX.__name__ == "RX" and Gates.is_rotational(Gates.CRZ)
    assert Gates.is_entangling(Gates.CX) and not Gates.is_entangling(Gates.RY)
    with recording() as t:
        Gates.RX(0.3, wires=1, gate_mode="unitary", pulse_params=None, something_else=5)
        Gates.CX(wires=[0, 1])
        Gates.Barrier(wires=[0, 1])
    assert [o.name for o in t] == ["RX", "CX", "Barrier"]
    with pytest.raises(ValueError, match="expects 2 wire"):
        op.CX(wires=[0])
    with pytest.raises(ValueError, match="duplicate wires"):
        op.CX(wires=[1, 1])
    with pytest.raises(NotImplementedError):
        Gates.RX(0.1, wires=0, gate_mode="pulse")
    with recording() as t:
        Gates.RX(0.1, wires=0, noise_params={"BitFlip": 0.1})
        Gates.NQubitDepolarizingChannel(0.1, wires=[0, 1])
    assert [o.name for o in t] == ["RX", "BitFlip", "QubitChannel"]
    with pytest.raises(ValueError, match="Invalid execution type"):
        Model(2, 1, "Circuit_1").execution_type = "nope"


def test_operation_matrices_match_oracle():
    from oracle import gates as G
    for name, args in (("RX", (0.3,)), ("RY", (1.2,)), ("RZ", (2.2,)), ("CRX", (0.4,)),
                       ("CRY", (0.5,)), ("CRZ", (0.6,)), ("RXX", (0.7,)), ("RYY", (0.8,)),
                       ("RZZ", (0.9,)), ("RZX", (1.0,)), ("Rot", (0.1, 0.2, 0.3))):
        got = getattr(op, name)(*args, wires=list(range(2)) if name[0] == "C" or len(name) == 3 and name != "Rot" else 0, record=False).matrix
        assert np.allclose(got, G.matrix(name, args)), name
    assert np.allclose(op.ControlledPhaseShift(0.37, wires=[0, 1], record=False).matrix, G.cphase(0.37))
    for cls, nm in ((op.CX, "CX"), (op.CY, "CY"), (op.CZ, "CZ"), (op.CCX, "CCX"),
                    (op.CSWAP, "CSWAP"), (op.SWAP, "SWAP"), (op.H, "H"), (op.S, "S")):
        assert np.allclose(cls._matrix, G.matrix(nm)), nm
    # batched parameter -> batch of matrices
    m = op.RX(np.array([0.1, 0.2, 0.3]) if False else Batched(np.array([0.1, 0.2, 0.3])), wires=0,
              record=False).matrix
    assert m.shape == (3, 2, 2) and np.allclose(m[1], G.matrix("RX", (0.2,)))
    # embedding: CX on [0,2] of 3 wires (MSB = wire 0)
    from oracle.dense import lift
    assert np.allclose(op.embed_matrix(G.matrix("CX"), [0, 2], [0, 1, 2]), lift(G.matrix("CX"), [0, 2], 3))
    assert np.allclose(op.embed_matrix(G.matrix("CRX", (0.3,)), [2, 0], [0, 1, 2]),
                       lift(G.matrix("CRX", (0.3,)), [2, 0], 3))


def test_script_in_axes_validation_and_tape_copy():
    s = Script(lambda th: op.RX(th, wires=0))
    with pytest.raises(ValueError, match="in_axes has"):  # tests/test_jaqsi.py:824-833
        s.execute(type="expval", obs=[op.PauliZ(0, record=False)], args=(np.zeros(3),), in_axes=(0, 0))
    with recording() as t:
        copy_to_tape(lambda: (op.H(wires=0), op.CX(wires=[0, 1])), offset=2)
    assert [(o.name, o.wires) for o in t] == [("H", [2]), ("CX", [2, 3])]


def test_tangent_tracking_of_gate_angles():
    """Forward-mode tangents of the batch tracer (used by Script.gradient): sums, constant
    scaling and products are tracked; non-linear maps are flagged unknown, constants known."""
    from qml_essentials_amd.batching import param_tangent

    x = Batched.leaf(np.array([[1.0, 2.0], [3.0, 4.0]]), 7)   # B = 2, leaf shape (2,)
    c = Batched(np.array([10.0, 20.0]), [])                    # batched constant
    y = 3.0 * x[1] + x[0] * c - x[1] / 2.0
    t = param_tangent(y)
    got = {}
    for lid, flat, coef in t:
        assert lid == 7
        got[flat] = got.get(flat, 0) + coef
    assert np.allclose(got[1], [2.5, 2.5]) and np.allclose(got[0], [10.0, 20.0])
    assert np.allclose(y.data, [3 * 2 + 1 * 10 - 1, 3 * 4 + 3 * 20 - 2])
    assert param_tangent(np.sin(x[0])) is None          # unknown derivative
    assert param_tangent(np.sin(c)) == []               # still a constant
    assert param_tangent(0.3) == [] and param_tangent(Batched(np.zeros(2))) is None
    prod = x[0] * x[1]                                    # product rule
    tt = {flat: coef for _, flat, coef in param_tangent(prod)}
    assert np.allclose(tt[0], [2.0, 4.0]) and np.allclose(tt[1], [1.0, 3.0])


def test_entanglement_density_matrix_helpers_host_math():
    """relative entropy / entanglement of formation: the host side (entanglement.py:307-372,
    438-468) on density matrices with known answers -- no GPU involved."""
    from qml_essentials_amd.entanglement import Entanglement, logm_v

    p, q = np.array([0.5, 0.25, 0.125, 0.125]), np.array([0.25, 0.25, 0.25, 0.25])
    rho, sigma = np.diag(p).astype(np.complex128), np.diag(q).astype(np.complex128)
    log_rho, log_sigma = logm_v(rho[None]) / np.log(2), logm_v(sigma) / np.log(2)
    assert log_rho.shape == (1, 4, 4) and np.allclose(np.diag(log_rho[0]).real, np.log2(p))
    kl = float(np.sum(p * np.log2(p / q)))
    one = Entanglement._compute_rel_entropies(rho[None], log_rho, log_sigma)
    assert one.shape == (1,) and abs(one[0] - kl) < 1e-12
    many = Entanglement._compute_rel_entropies(np.stack([rho, sigma]), np.stack([log_rho[0], log_sigma]),
                                               np.stack([log_sigma, log_rho[0]]))
    assert many.shape == (2, 2)                                   # (n_sigmas, n_rhos)
    assert abs(many[0, 0] - kl) < 1e-12 and abs(many[0, 1]) < 1e-12 and abs(many[1, 0]) < 1e-12
    assert abs(many[1, 1] - float(np.sum(q * np.log2(q / p)))) < 1e-12
    with pytest.raises(NotImplementedError):
        logm_v(np.zeros((2, 2, 2, 2)))
    # entanglement of formation, pure-state shortcut: Bell state -> 1, product state -> 0
    bell = np.zeros(4, dtype=np.complex128); bell[[0, 3]] = 2 ** -0.5
    prod = np.zeros(4, dtype=np.complex128); prod[1] = 1.0
    rhos = np.stack([np.outer(bell, bell.conj()), np.outer(prod, prod.conj())])
    eof = Entanglement._compute_entanglement_of_formation(rhos, 2, always_decompose=False)
    assert np.allclose(eof, [1.0, 0.0], atol=1e-12)


def test_real_input_fft_equals_the_complex_transform_and_is_exactly_hermitian():
    """Coefficients._fft_real (rfft + conjugate mirror) against fft / N of coefficients.py:143-150:
    odd and even lengths, one and several output columns; c_k == conj(c_-k) bit for bit."""
    from qml_essentials_amd.coefficients import Coefficients

    rng = np.random.default_rng(3)
    for n in (3, 4, 5, 8, 9, 120, 121, 4096):
        for shape in ((n,), (n, 3)):
            v = rng.standard_normal(shape)
            got = Coefficients._fft_real(v)
            np.testing.assert_allclose(got, np.fft.fft(v, axis=0) / n, rtol=0, atol=1e-15)
            assert np.array_equal(got[1:], np.conj(got[1:][::-1]))


def test_key_with_a_sequence_entropy_draws():
    """``SeedSequence([1, 2, 3])`` has a list entropy; the cached Philox key words must not
    need it hashable, and the draw equals numpy's own stream for that sequence."""
    from qml_essentials_amd import utils

    seq = np.random.SeedSequence([1, 2, 3])
    k = utils.PRNGKey(seq)
    got = utils.uniform(k, (7,), 0.0, 1.0)
    again = utils.uniform(utils.PRNGKey(np.random.SeedSequence([1, 2, 3])), (7,), 0.0, 1.0)
    assert np.array_equal(np.asarray(got), np.asarray(again))
    w = utils._philox_words(k)
    assert np.array_equal(w, seq.generate_state(2, np.uint64))
