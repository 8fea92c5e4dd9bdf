"""GPU parity of the Model / analysis-loop path against the oracle.

Tolerance: complex64 engine vs the complex128 oracle, |diff| <= 1e-6 on
expectation values (observables with unit norm, so this is the 1e-6 relative
bound of BASELINE.json's north_star) unless a test states otherwise."""
import json
import os

import numpy as np
import pytest

from oracle import analysis as OA
from oracle import circuits as OC
from oracle import einsum_sim as OE

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
c128 = np.complex128


def oracle_expval(spec, params, x, wires=None, **kw):
    tape = OC.model_tape(spec, params, np.atleast_1d(x), **kw)
    wires = range(spec.n_qubits) if wires is None else wires
    return OE.simulate_and_measure(tape, spec.n_qubits, "expval", [("PauliZ", [w]) for w in wires], c128)


def test_config1_model_4q_2l_hardware_efficient():
    """BASELINE config 1 (C1): expval for inputs 0 and 0.5."""
    from qml_essentials_amd.model import Model

    rng = np.random.default_rng(1000)
    m = Model(4, 2, "Hardware_Efficient")
    spec = OC.ModelSpec(4, 2, "Hardware_Efficient")
    assert m.params.shape == (1, 3, 12)
    p = rng.uniform(0, 2 * np.pi, (3, 12)).astype(np.float32)
    for x in (0.0, 0.5):
        got = m(params=p, inputs=np.array([x], dtype=np.float32))
        assert got.shape == (4,)
        assert np.abs(got - oracle_expval(spec, p, x)).max() < 1e-6


@pytest.mark.parametrize("ansatz", sorted(OC.STRUCTURES) + ["GHZ"])
def test_every_ansatz_expval_4q(ansatz):
    from qml_essentials_amd.model import Model

    rng = np.random.default_rng(7)
    m = Model(4, 2, ansatz)
    spec = OC.ModelSpec(4, 2, ansatz)
    p = rng.uniform(0, 2 * np.pi, spec.params_shape).astype(np.float32)
    got = m(params=p, inputs=np.array([0.8], dtype=np.float32))
    assert np.abs(got - oracle_expval(spec, p, 0.8)).max() < 1e-6


def test_batched_equals_sequential_and_shapes():
    """test_jaqsi.py:701-725 / test_model.py:107-130: batched == sequential; B_I x B_P."""
    from qml_essentials_amd.model import Model

    rng = np.random.default_rng(3)
    m = Model(3, 1, "Circuit_19")
    spec = OC.ModelSpec(3, 1, "Circuit_19")
    P = rng.uniform(0, 2 * np.pi, (4, *spec.params_shape)).astype(np.float32)
    X = rng.uniform(0, 3, (5, 1)).astype(np.float32)
    got = m(params=P, inputs=X)
    assert got.shape == (5, 4, 3)
    for i in range(5):
        for j in range(4):
            want = oracle_expval(spec, P[j], X[i], zero_inputs_batch1=False)
            assert np.abs(got[i, j] - want).max() < 1e-6
    dens = m(params=P, execution_type="density")
    assert dens.shape == (4, 8, 8)
    for j in range(4):
        single = m(params=P[j], execution_type="density")
        assert np.abs(single - dens[j]).max() < 1e-6
    m = Model(3, 1, "Circuit_19")
    m.params = P
    assert m(inputs=X, force_mean=True).shape == (5, 4)


def test_output_shapes_and_partial_measurements():
    """tests/test_model.py:928-1053 (the shot-free rows) + test_parity :1057-1079."""
    from qml_essentials_amd.model import Model
    import warnings

    x3 = np.array([0.1, 0.2, 0.3], dtype=np.float32)
    cases = [(np.array(0.1), "expval", [0, 1], False, (2,)), (x3, "expval", [0, 1], False, (3, 2)),
             (x3, "expval", [0, 1], True, (3,)), (None, "density", -1, False, (4, 4)),
             (x3, "density", -1, False, (3, 4, 4)), (x3, "density", 0, False, (3, 2, 2)),
             (x3, "probs", -1, False, (3, 2, 2)), (x3, "probs", 0, False, (3, 2)),
             (x3, "state", -1, False, (3, 4))]
    for inputs, et, oq, fm, shape in cases:
        m = Model(2, 1, "Circuit_19", output_qubit=oq)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = m(m.params, inputs=inputs, force_mean=fm, execution_type=et)
        assert out.shape == shape, (et, oq, out.shape)
    # partial density / probs agree with the oracle reductions
    rng = np.random.default_rng(5)
    spec = OC.ModelSpec(3, 1, "Circuit_19")
    p = rng.uniform(0, 6, spec.params_shape).astype(np.float32)
    psi = OE.simulate_pure(OC.model_tape(spec, p, [0.4]), 3, c128)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = Model(3, 1, "Circuit_19", output_qubit=[0, 2])
        rho = m(p, inputs=np.array([0.4]), execution_type="density")
        assert np.abs(rho - OA.partial_trace(np.outer(psi, psi.conj()), 3, [0, 2])).max() < 1e-6
        pr = m(p, inputs=np.array([0.4]), execution_type="probs")
        assert np.abs(pr.reshape(-1) - OA.marginalize_probs(np.abs(psi) ** 2, 3, [0, 2])[0]).max() < 1e-6
    # parity observable Z0 Z1 vs oracle general path
    ma = Model(2, 1, "Circuit_1", output_qubit=[[0, 1]])
    pa = np.asarray(ma.params[0])
    ra = ma(params=pa, inputs=None)
    zz = np.diag([1, -1, -1, 1]).astype(c128)
    spec1 = OC.ModelSpec(2, 1, "Circuit_1")
    want = OE.simulate_and_measure(OC.model_tape(spec1, pa, [0.0]), 2, "expval",
                                   [("ZZ", [0, 1], zz)], c128)
    assert np.abs(np.atleast_1d(ra) - want).max() < 1e-6


def test_general_observables_x_and_hermitian():
    """simulation.py:263-269 general path, matrix-free on the GPU; Bell <X>=<Z>=0
    (test_jaqsi.py:358-362) and a random 2-qubit Hermitian."""
    from qml_essentials_amd import operations as op
    from qml_essentials_amd.script import Script

    def bell():
        op.H(wires=0)
        op.CX(wires=[0, 1])

    s = Script(bell)
    for cls in (op.PauliX, op.PauliZ, op.PauliY):
        r = s.execute(type="expval", obs=[cls(0, record=False), cls(1, record=False)])
        assert np.abs(r).max() < 1e-6
    assert np.allclose(s.execute(type="probs"), [0.5, 0, 0, 0.5], atol=1e-6)
    rng = np.random.default_rng(9)
    A = rng.normal(size=(4, 4)) + 1j * rng.normal(size=(4, 4))
    Hm = (A + A.conj().T) / 2

    def circ(th):
        op.RY(th, wires=0)
        op.CRX(0.7, wires=[0, 2])
        op.RX(th * 2, wires=1)

    s = Script(circ, n_qubits=3)
    th = np.linspace(0, 2, 5).astype(np.float32)
    got = s.execute(type="expval", obs=[op.Hermitian(Hm, wires=[2, 0], record=False),
                                        op.PauliX(1, record=False)], args=(th,), in_axes=(0,))
    for b, t in enumerate(th):
        tape = [("RY", [0], (float(t),)), ("CRX", [0, 2], (0.7,)), ("RX", [1], (float(2 * t),))]
        want = OE.simulate_and_measure(tape, 3, "expval", [("H", [2, 0], Hm), ("PauliX", [1])], c128)
        assert np.abs(got[b] - want).max() < 2e-6
    # two-argument broadcast: cos(theta + phi)  (test_jaqsi.py:789-821)
    s2 = Script(lambda a, b: (op.RX(a, wires=0), op.RX(b, wires=0)))
    r = s2.execute(type="expval", obs=[op.PauliZ(0, record=False)],
                   args=(th, np.float32(0.3)), in_axes=(0, None))
    assert np.abs(r[:, 0] - np.cos(th + 0.3)).max() < 1e-6


def test_config2_model_20q_4l_vs_oracle():
    """BASELINE config 2 (C2): n=20, L=4, HE, 480 gates; <= 1e-6 vs the fp64 oracle."""
    from qml_essentials_amd.model import Model

    rng = np.random.default_rng(1000)
    m = Model(20, 4, "Hardware_Efficient")
    spec = OC.ModelSpec(20, 4, "Hardware_Efficient")
    p = rng.uniform(0, 2 * np.pi, spec.params_shape).astype(np.float32)
    got = m(params=p, inputs=np.array([0.5], dtype=np.float32))
    want = oracle_expval(spec, p, 0.5)
    assert got.shape == (20,)
    assert np.abs(got - want).max() < 1e-6, np.abs(got - want).max()


def test_golomb_model_vs_oracle():
    from qml_essentials_amd.ansaetze import Encoding
    from qml_essentials_amd.model import Model

    m = Model(3, 1, "Circuit_19", encoding=Encoding("golomb", None))
    spec = OC.ModelSpec(3, 1, "Circuit_19", strategy="golomb")
    p = np.random.default_rng(2).uniform(0, 6, spec.params_shape).astype(np.float32)
    X = np.array([[0.1], [0.45]], dtype=np.float32)
    got = m(params=p, inputs=X)
    for i in range(2):
        assert np.abs(got[i] - oracle_expval(spec, p, X[i], zero_inputs_batch1=False)).max() < 5e-6


def test_expressibility_pure_formula_equals_reference_form():
    """F via the overlap kernel == the reference's density + sqrtm formula (n=3)."""
    from qml_essentials_amd.expressibility import Expressibility
    from qml_essentials_amd.model import Model

    m = Model(3, 1, "Circuit_19", data_reupload=False)
    fid = Expressibility._sample_state_fidelities(m, 12, random_key=11, kwargs={}).cpu().numpy()
    P = np.asarray(m.params)
    assert P.shape[0] == 24 and fid.shape == (12,)
    spec = OC.ModelSpec(3, 1, "Circuit_19", data_reupload=False)
    states = np.array([OE.simulate_pure(OC.model_tape(spec, P[i], [0.0]), 3, c128) for i in range(24)])
    rhos = np.array([np.outer(s, s.conj()) for s in states])
    assert np.abs(fid - OA.fidelities_reference_form(rhos, 12)).max() < 1e-5
    assert np.abs(fid - OA.fidelities_pure(states, 12)).max() < 1e-6
    y, z = Expressibility.state_fidelities(n_samples=10, n_bins=4, model=Model(2, 1, "Circuit_1"),
                                           scale=True)
    assert z.shape == (8,) and abs(z.sum() - 1) < 1e-6  # test_expressiblity.py:192-215
    _, h = Expressibility.haar_integral(n_qubits=2, n_bins=4, scale=True)
    assert h.shape == (8,)
    assert np.allclose(Expressibility.haar_integral(4, 75)[1], OA.haar_integral(4, 75), atol=1e-10)
    ha = Expressibility.haar_integral(2, 10)[1]
    assert abs(Expressibility.kullback_leibler_divergence(ha, ha).mean()) < 1e-3


@pytest.mark.parametrize("layers", [1, 3])
def test_expressibility_sim_et_al_table(layers, golden_dir):
    """tests/test_expressiblity.py:116-188: KL to Haar within 40 % of Sim et al."""
    from qml_essentials_amd.expressibility import Expressibility
    from qml_essentials_amd.model import Model

    fx = json.load(open(os.path.join(golden_dir, "sim_et_al.json")))["expressibility"]
    skip = set(fx[f"skip_layers_{layers}"])
    got, want = [], []
    for cid, ref in zip(fx["circuits"], fx[f"kl_layers_{layers}"]):
        if cid in skip:
            continue
        m = Model(n_qubits=4, n_layers=layers, circuit_type=f"Circuit_{cid}",
                  initialization_domain=fx["initialization_domain"], data_reupload=False)
        kl = float(Expressibility.kl_divergence_to_haar(m, n_samples=fx["n_samples"],
                                                        n_bins=fx["n_bins"], random_key=1000).mean())
        assert abs(kl - ref) / ref < fx["tolerance"], (cid, kl, ref)
        got.append(kl)
        want.append(ref)
    # rank agreement (the reference asserts identical order with its own RNG stream)
    from scipy.stats import spearmanr
    assert spearmanr(got, want).correlation > 0.9


def test_meyer_wallach_table_and_anchors(golden_dir):
    """tests/test_entanglement.py:100-181: within 55 % of Sim et al.; C1 -> 0, C9 -> 1."""
    from qml_essentials_amd.entanglement import Entanglement
    from qml_essentials_amd.model import Model

    fx = json.load(open(os.path.join(golden_dir, "sim_et_al.json")))["meyer_wallach"]
    got, want = [], []
    for cid, ref in zip(fx["circuits"], fx["mw_layers_1"]):
        if cid in fx["skip"]:
            continue
        m = Model(4, 1, f"Circuit_{cid}", data_reupload=False)
        q = Entanglement.meyer_wallach(m, n_samples=fx["n_samples"], random_key=1000)
        if ref in (0.0, 1.0):
            assert abs(q - ref) < 1e-5, (cid, q)
        else:
            assert abs(q - ref) / ref < fx["tolerance"], (cid, q, ref)
        got.append(q)
        want.append(ref)
    from scipy.stats import spearmanr
    assert spearmanr(got, want).correlation > 0.9
    # no sampling: uses the model's current params (tests/test_entanglement.py:185-200)
    m = Model(3, 2, "Strongly_Entangling")
    q = Entanglement.meyer_wallach(m, n_samples=None)
    spec = OC.ModelSpec(3, 2, "Strongly_Entangling")
    psi = OE.simulate_pure(OC.model_tape(spec, m.params[0], [0.0]), 3, c128)
    assert abs(q - OA.meyer_wallach_pure(psi, 3)) < 1e-6
    rho = np.outer(psi, psi.conj())[None]
    assert abs(Entanglement._compute_meyer_wallach_meas(rho, 3)[0] - OA.meyer_wallach_reference_form(rho[0], 3)) < 1e-10


def test_coefficients_spectrum_vs_oracle():
    """coefficients.py:109-150 vs the oracle FFT on the same parameters; series
    re-evaluation (test_coefficients.py:59-70) and |c_k| = |c_-k| (:259-269)."""
    from qml_essentials_amd.coefficients import Coefficients
    from qml_essentials_amd.model import Model

    m = Model(2, 1, "Circuit_19")
    spec = OC.ModelSpec(2, 1, "Circuit_19")
    p = np.asarray(m.params[0])
    coeffs, freqs = Coefficients.get_spectrum(m)
    axes, grid, n_freqs = OA.fourier_grid(spec.degree)
    outs = np.array([oracle_expval(spec, p, x, zero_inputs_batch1=False).mean() for x in grid])
    want_c, want_f = OA.fourier_transform(outs, axes, n_freqs)
    assert np.allclose(freqs, want_f[0]) and np.abs(coeffs - want_c).max() < 1e-6
    for x in (0.3, 2.2):
        series = Coefficients.evaluate_Fourier_series(coeffs, freqs, np.array([x]))
        assert abs(series - m(inputs=np.array([x], dtype=np.float32), force_mean=True)) < 1e-5
    assert np.allclose(np.abs(coeffs[1:]), np.abs(coeffs[1:][::-1]), atol=1e-6)
    c2, f2 = Coefficients.get_spectrum(m, mfs=3, shift=True)
    assert len(c2) == 15 and len(f2) == 15  # test_coefficients.py:242-256
    c3, f3 = Coefficients.get_spectrum(m, mfs=2, trim=True)
    assert len(c3) == 9  # even spectrum loses exactly the Nyquist bin (:272-286)
    # two input features: 2-D grid, feature 0 slowest
    m2 = Model(2, 1, "Circuit_19", encoding=["RX", "RY"])
    c, f = Coefficients.get_spectrum(m2)
    assert c.shape == (5, 5) and len(f) == 2


def test_chunked_equals_full(monkeypatch):
    """tests/test_jaqsi.py:1914-1983: memory-aware chunking must not change results."""
    from qml_essentials_amd import memory
    from qml_essentials_amd.model import Model

    rng = np.random.default_rng(8)
    m = Model(6, 2, "Circuit_19")
    P = rng.uniform(0, 2 * np.pi, (37, *m.params.shape[1:])).astype(np.float32)
    full_e = m(params=P, inputs=np.array([0.3], dtype=np.float32))
    full_s = m(params=P, inputs=np.array([0.3], dtype=np.float32), execution_type="state")
    calls = []
    real = memory.execute_chunked

    def spy(run, batch, chunk):
        calls.append((batch, chunk))
        return real(run, batch, chunk)

    monkeypatch.setattr(memory, "available_memory_bytes", lambda: 1 << 20)  # 1 MiB "free"
    monkeypatch.setattr(memory, "execute_chunked", spy)
    m.execution_type = "expval"
    chunk_e = m(params=P, inputs=np.array([0.3], dtype=np.float32))
    chunk_s = m(params=P, inputs=np.array([0.3], dtype=np.float32), execution_type="state")
    assert calls and all(c < b for b, c in calls)
    assert np.array_equal(full_e, chunk_e) and np.array_equal(full_s, chunk_s)


def test_large_n_partial_probs_native_marginal():
    """probs on a wire subset at n > 10 goes through qmle_marginal_probs."""
    import warnings
    from qml_essentials_amd.model import Model

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = Model(12, 1, "Hardware_Efficient", output_qubit=[0, 11, 5])
        p = np.asarray(m.params[0])
        got = m(params=p, inputs=np.array([0.2], dtype=np.float32), execution_type="probs")
    spec = OC.ModelSpec(12, 1, "Hardware_Efficient")
    psi = OE.simulate_pure(OC.model_tape(spec, p, [0.2]), 12, c128)
    want = OA.marginalize_probs(np.abs(psi) ** 2, 12, [0, 11, 5])[0]
    assert got.shape == (2, 2, 2) and np.abs(got.reshape(-1) - want).max() < 1e-6


def test_baseline_sizes_by_properties():
    """Size-independent checks at BASELINE sizes the oracle cannot reach quickly:
    n = 24 fused == unfused on the same angles; norm and <Z> bounds; n = 26 Meyer-Wallach
    of a product state is 0 and of a GHZ state is 1."""
    from qml_essentials_amd import _native as N
    from qml_essentials_amd.entanglement import Entanglement
    from qml_essentials_amd.model import Model
    from tests.test_abi_cpu import he_layer_ops

    n = 24
    ops, slots = he_layer_ops(n)
    ang = torch.from_numpy(np.random.default_rng(3).uniform(0, 6.28, (3, slots)).astype(np.float32)).cuda()
    ef = N.Plan(ops, n, slots).run(ang, "expval", list(range(n)))
    eu = N.Plan(ops, n, slots, flags=N.plan_flags(no_fusion=True)).run(ang, "expval", list(range(n)))
    assert float((ef - eu).abs().max()) < 1e-5 and float(ef.abs().max()) <= 1 + 1e-5
    n = 26
    prod = N.Plan([("RY", [q], [q], -1) for q in range(n)], n, n)
    a = torch.from_numpy(np.random.default_rng(4).uniform(0, 6.28, (1, n)).astype(np.float32)).cuda()
    st = prod.run(a, "state")
    assert abs(float((st.abs() ** 2).sum()) - 1) < 1e-4
    assert abs(float(N.meyer_wallach(st)[0])) < 1e-4
    del st
    ghz = N.Plan([("H", [0], [], -1)] + [("CX", [q, q + 1], [], -1) for q in range(n - 1)], n, 0)
    st = ghz.run(None, "state")
    assert abs(float(N.meyer_wallach(st)[0]) - 1) < 1e-5
    assert abs(float(st[0, 0].abs() ** 2) - 0.5) < 1e-6 and abs(float(st[0, -1].abs() ** 2) - 0.5) < 1e-6


def test_bell_measurement_equals_meyer_wallach():
    """tests/test_entanglement.py:288-313: MW == Bell measurement (abs 1e-5) on the same
    parameter samples; plus the 2n-qubit circuit against the oracle for fixed parameters."""
    from copy import deepcopy
    from qml_essentials_amd.entanglement import Entanglement
    from qml_essentials_amd.model import Model

    for nq in (2, 4):
        model = Model(n_qubits=nq, n_layers=1, circuit_type="Circuit_4", data_reupload=False)
        mw = Entanglement.meyer_wallach(deepcopy(model), n_samples=1000, random_key=1000)
        bell = Entanglement.bell_measurements(deepcopy(model), n_samples=1000, random_key=1000)
        assert abs(mw - bell) < 1e-5, (nq, mw, bell)
    # explicit oracle for one parameter set, n = 3
    m = Model(3, 1, "Circuit_19", data_reupload=False)
    p = np.asarray(m.params[0])
    spec = OC.ModelSpec(3, 1, "Circuit_19", data_reupload=False)
    base = [g for g in OC.model_tape(spec, p, [0.0]) if g[0] != "Barrier"]
    shifted = [(g[0], [w + 3 for w in g[1]], g[2]) for g in base]
    tape = base + shifted + [x for q in range(3) for x in (("CX", [q, q + 3], ()), ("H", [q], ()))]
    pr = np.abs(OE.simulate_pure(tape, 6, c128)) ** 2
    exp = np.array([1 - 2 * OA.marginalize_probs(pr, 6, [q, q + 3])[0][-1] for q in range(3)])
    want = min(max(float((2 * (1 - exp)).mean()), 0.0), 1.0)
    got = Entanglement.bell_measurements(m, n_samples=None)
    assert abs(got - want) < 1e-6


def test_concentratable_entanglement_vs_oracle():
    """entanglement.py:471-576 swap test on 3n qubits (CSWAP path) for fixed parameters."""
    from qml_essentials_amd.entanglement import Entanglement
    from qml_essentials_amd.model import Model

    n = 2
    m = Model(n, 1, "Circuit_19", data_reupload=False)
    p = np.asarray(m.params[0])
    spec = OC.ModelSpec(n, 1, "Circuit_19", data_reupload=False)
    base = [g for g in OC.model_tape(spec, p, [0.0]) if g[0] != "Barrier"]
    tape = [(g[0], [w + n for w in g[1]], g[2]) for g in base]
    tape += [(g[0], [w + 2 * n for w in g[1]], g[2]) for g in base]
    tape += [("H", [i], ()) for i in range(n)]
    tape += [("CSWAP", [i, i + n, i + 2 * n], ()) for i in range(n)]
    tape += [("H", [i], ()) for i in range(n)]
    pr = np.abs(OE.simulate_pure(tape, 3 * n, c128)) ** 2
    want = 1 - OA.marginalize_probs(pr, 3 * n, list(range(n)))[0][0]
    got = Entanglement.concentratable_entanglement(m, n_samples=None)
    assert abs(got - want) < 1e-6
    # product state (no entangler) -> 0;  sampling path runs
    m0 = Model(2, 1, "Circuit_1", data_reupload=False)
    assert abs(Entanglement.concentratable_entanglement(m0, n_samples=8, random_key=3)) < 1e-6


def test_device_resident_arguments_match_host_path():
    """CUDA-tensor params / inputs take the compiled device path (qmle_build_angles):
    identical results to the host-array path, result stays on the GPU."""
    import warnings
    from qml_essentials_amd.ansaetze import Encoding
    from qml_essentials_amd.model import Model

    rng = np.random.default_rng(12)
    for kw in (dict(circuit_type="Hardware_Efficient"), dict(circuit_type="Circuit_19"),
               dict(circuit_type="Strongly_Entangling", encoding=Encoding("binary", ["RX", "RY"])),
               dict(circuit_type="Circuit_19", data_reupload=False)):
        m = Model(4, 2, **kw)
        P = rng.uniform(0, 2 * np.pi, (5, *m.params.shape[1:])).astype(np.float32)
        X = rng.uniform(0, 3, (3, m.n_input_feat)).astype(np.float32)
        Pd, Xd = torch.from_numpy(P).cuda(), torch.from_numpy(X).cuda()
        for et in ("expval", "state", "probs"):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                host = m(params=P, inputs=X, execution_type=et)
                dev = m(params=Pd, inputs=Xd, execution_type=et)
            assert torch.is_tensor(dev) and dev.is_cuda and tuple(dev.shape) == host.shape
            assert np.abs(dev.cpu().numpy() - host).max() < 1e-6, (kw, et)
        # mixed: params on device, inputs None / host scalar; single sample
        a = m(params=Pd, inputs=None, execution_type="expval", force_mean=True)
        b = m(params=P, inputs=None, force_mean=True)
        assert np.abs(a.cpu().numpy() - b).max() < 1e-6
        a1 = m(params=Pd[0], inputs=Xd[:1])
        b1 = m(params=P[0], inputs=X[:1])
        assert np.abs(a1.cpu().numpy() - b1).max() < 1e-6
    # zipped batch axes
    m = Model(2, 1, "Circuit_19", repeat_batch_axis=[False, True, True])
    P = rng.uniform(0, 6, (6, *m.params.shape[1:])).astype(np.float32)
    X = rng.uniform(0, 1, (6, 1)).astype(np.float32)
    d = m(params=torch.from_numpy(P).cuda(), inputs=torch.from_numpy(X).cuda())
    assert tuple(d.shape) == (6, 2) and np.abs(d.cpu().numpy() - m(params=P, inputs=X)).max() < 1e-6
    # compiled-call cache: second call re-uses the trace
    n_before = len(m.script._compiled)
    m(params=torch.from_numpy(P).cuda(), inputs=torch.from_numpy(X).cuda())
    assert len(m.script._compiled) == n_before


def test_device_arguments_of_an_ansatz_without_parameters():
    """GHZ / No_Ansatz have zero parameters per layer: ``model.params`` is ``(1, L, 0)`` and its CUDA copy an
    empty tensor (NULL data pointer).  The device route must take it as the one parameter set it is
    (``B_P = 1 if 0 in params.shape``, model.py:1444) -- below and above the 64 samples from which the gate
    matrices are built straight from the angle map -- and agree with the host route."""
    from qml_essentials_amd.model import Model

    rng = np.random.default_rng(31)
    for kw in (dict(circuit_type="GHZ"), dict(circuit_type="No_Ansatz", encoding="RY"),
               dict(circuit_type="GHZ", data_reupload=False)):
        m = Model(5, 2, **kw)
        assert 0 in m.params.shape
        Pd = torch.from_numpy(np.asarray(m.params, dtype=np.float32)).cuda()
        for B in (3, 70):
            X = rng.uniform(-3, 3, (B, m.n_input_feat)).astype(np.float32)
            for et in ("expval", "probs"):
                host = m(inputs=X, execution_type=et)
                dev = m(params=Pd, inputs=torch.from_numpy(X).cuda(), execution_type=et)
                assert tuple(dev.shape) == host.shape, (kw, B, et, tuple(dev.shape), host.shape)
                assert np.abs(dev.cpu().numpy() - host).max() < 1e-6, (kw, B, et)


def test_concentratable_entanglement_estimation_equals_swap_test():
    """entanglement.py:579-684 vs :471-576: the 2n-qubit Bell-basis estimate and the 3n-qubit
    swap test measure the same quantity; order of circuits as test_entanglement.py:411-468."""
    from copy import deepcopy

    from qml_essentials_amd.entanglement import Entanglement
    from qml_essentials_amd.model import Model
    from qml_essentials_amd.utils import key

    vals = []
    for circuit in ["Circuit_1", "Circuit_16", "Circuit_19", "Circuit_15", "Strongly_Entangling"]:
        model = Model(n_qubits=3, n_layers=1, circuit_type=circuit)
        a = Entanglement.concentratable_entanglement(deepcopy(model), n_samples=200,
                                                     random_key=key(1000))
        b = Entanglement.concentratable_entanglement_estimation(deepcopy(model), n_samples=200,
                                                                random_key=key(1000))
        assert abs(a - b) < 1e-5, (circuit, a, b)
        vals.append(b)
    assert vals[0] < 1e-6                              # Circuit_1: product states
    assert all(vals[i] <= vals[i + 1] + 1e-9 for i in range(len(vals) - 1)), vals
    # no sampling / scaling smoke (test_entanglement.py:185-199,317-326)
    model = Model(n_qubits=2, n_layers=1, circuit_type="Hardware_Efficient", data_reupload=False)
    for fn in (Entanglement.meyer_wallach, Entanglement.bell_measurements,
               Entanglement.concentratable_entanglement,
               Entanglement.concentratable_entanglement_estimation):
        assert 0.0 <= fn(deepcopy(model), n_samples=None) <= 1.0
        assert 0.0 <= fn(deepcopy(model), n_samples=10, scale=True) <= 1.0


def test_expressibility_scaling_and_haar_cache():
    """test_expressiblity.py:83-111 (cached == uncached Haar integral) and :192-215 (scale=True
    gives n_bins * n_qubits = 8 bins)."""
    from qml_essentials_amd.expressibility import Expressibility
    from qml_essentials_amd.model import Model

    _, ya = Expressibility.haar_integral(n_qubits=2, n_bins=10, cache=True)
    _, yb = Expressibility.haar_integral(n_qubits=2, n_bins=10, cache=False)
    assert abs(float(np.mean(Expressibility.kullback_leibler_divergence(ya, yb)))) < 1e-3
    model = Model(n_qubits=2, n_layers=1, circuit_type="Circuit_1")
    _, z = Expressibility.state_fidelities(n_bins=4, n_samples=10, model=model, scale=True)
    assert z.shape == (8,)
    _, y = Expressibility.haar_integral(n_qubits=model.n_qubits, n_bins=4, cache=False, scale=True)
    assert y.shape == (8,)


def test_large_encoding_angles_keep_float32_accuracy():
    """Ternary encoding scales the input by 3^q: at 8 qubits the encoding angles reach
    ~10^4 rad.  Host path (fp64 angles, reduced mod 4 pi before the float32 cast) and device
    path (qmle_build_angles: fp64 sum + the same reduction) both stay within 2e-6 of the
    fp64 oracle fed the float32-representable input (float32 angles of 7e3 rad would be off by
    ~5e-5)."""
    import warnings
    from qml_essentials_amd.ansaetze import Encoding
    from qml_essentials_amd.model import Model

    rng = np.random.default_rng(99)
    n = 8
    m = Model(n, 1, "Circuit_19", encoding=Encoding("ternary", ["RX"]))
    spec = OC.ModelSpec(n, 1, "Circuit_19", strategy="ternary")
    p = rng.uniform(0, 2 * np.pi, spec.params_shape).astype(np.float32)
    X = rng.uniform(1.0, 5.0, (3, 1)).astype(np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        host = m(params=p, inputs=X)
        dev = m(params=torch.from_numpy(p).cuda(), inputs=torch.from_numpy(X).cuda()).cpu().numpy()
        m.host_arrays_via_device = False   # host-built angle table (LoweredTape.angle_table)
        assert np.abs(m(params=p, inputs=X) - host).max() < 1e-6
    for i in range(3):
        want = oracle_expval(spec, p, X[i].astype(np.float64))
        assert np.abs(host[i] - want).max() < 2e-6, np.abs(host[i] - want).max()
        assert np.abs(dev[i] - want).max() < 2e-6, np.abs(dev[i] - want).max()


def test_state_ignores_output_qubit():
    """execution_type='state' returns the full register whatever output_qubit says (the
    reference sizes the state result by output_qubit, model.py:365, and then cannot reshape)."""
    from qml_essentials_amd.model import Model

    m = Model(3, 1, "Circuit_19", output_qubit=0)
    x = np.array([[0.2], [0.7]], dtype=np.float32)
    s = m(inputs=x, execution_type="state")
    full = Model(3, 1, "Circuit_19")
    assert s.shape == (2, 8)
    assert np.abs(s - full(params=m.params, inputs=x, execution_type="state")).max() < 1e-6
    assert np.allclose(np.sum(np.abs(s) ** 2, axis=-1), 1.0, atol=1e-5)


def test_output_qubit_change_between_calls_rebuilds_observables():
    """The observable list is cached per ``output_qubit``: changing it between calls (int, list,
    parity pairs) must change the measurement, on the host-array and the device-tensor path."""
    from qml_essentials_amd.model import Model

    m = Model(4, 2, "Hardware_Efficient")
    x = np.array([[0.3], [0.9]], dtype=np.float32)
    full = m(inputs=x)                                   # all wires
    assert full.shape == (2, 4)
    m.output_qubit = 1
    one = m(inputs=x)
    assert one.shape == (2,) and np.abs(one - full[:, 1]).max() < 1e-6
    m.output_qubit = [0, 3]
    two = m(inputs=torch.from_numpy(x).cuda()).cpu().numpy()
    assert two.shape == (2, 2) and np.abs(two - full[:, [0, 3]]).max() < 1e-6
    m.output_qubit = [[0, 1], [2, 3]]
    par = m(inputs=x)
    spec = OC.ModelSpec(4, 2, "Hardware_Efficient")
    for i in range(2):
        tape = OC.model_tape(spec, m.params[0] if m.params.ndim == 3 else m.params, x[i])
        psi = OE.simulate_pure(tape, 4, np.complex128)
        pr = np.abs(psi) ** 2
        idx = np.arange(16)
        for k, (a_, b_) in enumerate([(0, 1), (2, 3)]):
            sign = 1.0 - 2.0 * (((idx >> (3 - a_)) ^ (idx >> (3 - b_))) & 1)
            assert abs(par[i, k] - (pr * sign).sum()) < 1e-6
    m.output_qubit = -1
    assert np.abs(m(inputs=x) - full).max() < 1e-7


def test_entanglement_of_formation_order_and_pure_state_limit():
    """tests/test_entanglement.py:381-407: entanglement of formation orders the circuits like
    Meyer-Wallach does; for pure states it IS Meyer-Wallach unless always_decompose is set, and
    then the eigen-decomposition of a pure state gives the same number."""
    from qml_essentials_amd.entanglement import Entanglement
    from qml_essentials_amd.model import Model
    from qml_essentials_amd.utils import random

    vals = []
    for circuit in ["Circuit_1", "Circuit_16", "Circuit_19", "Circuit_15", "Strongly_Entangling"]:
        m = Model(n_qubits=3, n_layers=1, circuit_type=circuit)
        vals.append(Entanglement.entanglement_of_formation(m, n_samples=500, random_key=random.key(1000)))
    assert all(vals[i] <= vals[i + 1] + 1e-9 for i in range(len(vals) - 1)), vals
    assert abs(vals[0]) < 1e-6                                  # Circuit_1: product states
    m = Model(3, 1, "Strongly_Entangling")
    key = random.key(7)
    eof = Entanglement.entanglement_of_formation(m, n_samples=40, random_key=key)
    mw = Entanglement.meyer_wallach(m, n_samples=None)          # same parameters
    dec = Entanglement.entanglement_of_formation(m, n_samples=None, always_decompose=True)
    assert abs(eof - mw) < 1e-5 and abs(dec - mw) < 1e-4
    # a mixed state: depolarised Bell-like output has less entanglement of formation
    noisy = Entanglement.entanglement_of_formation(m, n_samples=None, noise_params={"Depolarizing": 0.2})
    assert 0.0 <= noisy < eof


def test_relative_entropy_order():
    """tests/test_entanglement.py:330-377: relative entropy of entanglement against random
    separable states, normalised by GHZ -- product-state circuit lowest, GHZ itself 1."""
    from qml_essentials_amd.entanglement import Entanglement, sample_random_separable_states
    from qml_essentials_amd.model import Model
    from qml_essentials_amd.utils import random

    sig = sample_random_separable_states(3, 5, random.key(3))
    assert sig.shape == (5, 8, 8)
    assert np.allclose(np.trace(sig, axis1=1, axis2=2), 1.0, atol=1e-5)
    assert np.allclose(np.trace(sig @ sig, axis1=1, axis2=2).real, 1.0, atol=1e-5)   # pure
    ent = []
    for circuit in ["Circuit_1", "Circuit_16", "Circuit_19", "Strongly_Entangling"]:
        m = Model(n_qubits=3, n_layers=1, circuit_type=circuit)
        ent.append(Entanglement.relative_entropy(m, n_samples=50, n_sigmas=100, random_key=random.key(1000)))
    ghz = Model(n_qubits=3, n_layers=1, circuit_type="GHZ", data_reupload=False)
    ent.append(Entanglement.relative_entropy(ghz, n_samples=1, n_sigmas=100, random_key=random.key(1000)))
    assert all(ent[i] <= ent[i + 1] for i in range(len(ent) - 1)), ent
    assert abs(ent[-1] - 1.0) < 1e-6


def test_analysis_loops_beyond_the_launch_row_limit():
    """More samples than one launch holds (65 535 states; 32 767 for the adjoint sweep, which keeps
    psi and lambda of a sample side by side): Meyer-Wallach over 70 000 parameter samples and
    adjoint gradients of 33 000 inputs run in slices; rows around the seam equal the same rows
    computed in a small batch."""
    from qml_essentials_amd.entanglement import Entanglement
    from qml_essentials_amd.model import Model

    m = Model(3, 1, "Hardware_Efficient")
    q = Entanglement.meyer_wallach(m, n_samples=70000, random_key=3)
    assert np.isfinite(q) and 0.0 < float(q) < 1.0
    q_small = Entanglement.meyer_wallach(m, n_samples=5000, random_key=3)
    assert abs(float(q) - float(q_small)) < 0.02
    m2 = Model(3, 1, "Hardware_Efficient")
    x = np.linspace(-1.0, 1.0, 33000)
    g = np.asarray(m2.gradient(inputs=x, method="adjoint", force_mean=True))
    rows = [0, 32766, 32767, 32768, 32999]
    g_sub = np.asarray(m2.gradient(inputs=x[rows], method="adjoint", force_mean=True))
    assert g.shape[0] == 33000 and np.allclose(g[rows], g_sub, atol=1e-6)


def test_prepared_state_call_leaves_the_model_consistent():
    """the analysis loops' prepared state call switches the model to "state" through the
    execution_type setter (which derives the result shape).  Expressibility -> expval call ->
    Expressibility again (the prepared path) -> model() must return states of shape (.., 2^n)."""
    from qml_essentials_amd.expressibility import Expressibility
    from qml_essentials_amd.model import Model

    model = Model(n_qubits=3, n_layers=1, circuit_type="Hardware_Efficient")
    S = 512  # 2 * 512 * 18 values: drawn by the device sampler, so the second call is the prepared path
    Expressibility.state_fidelities(n_bins=8, n_samples=S, model=model)
    assert model.device_params() is not None
    ev = np.asarray(model(execution_type="expval"))
    assert ev.shape == (2 * S, 3)
    Expressibility.state_fidelities(n_bins=8, n_samples=S, model=model)  # the prepared path
    assert model.execution_type == "state"
    out = np.asarray(model())  # no explicit type: whatever the analysis left must be consistent
    assert out.shape == (2 * S, 8)
    assert np.allclose((np.abs(out) ** 2).sum(-1), 1.0, atol=1e-6)


def test_device_resident_params_host_mirror_is_a_read_only_snapshot():
    """with parameters living on the GPU, ``model.params`` is a read-only snapshot (an
    in-place edit raises instead of being silently ignored), assignment takes effect, and in-place
    updates of a CUDA tensor handed to the setter are seen by the next read and the next call."""
    from qml_essentials_amd.model import Model

    model = Model(n_qubits=3, n_layers=1, circuit_type="Hardware_Efficient")
    p = torch.full((1, *model.params.shape[-2:]), 0.3, device="cuda", dtype=torch.float32)
    model.params = p
    snap = model.params
    with pytest.raises(ValueError):
        snap[0, 0, 0] = 1.0
    a = np.asarray(model(execution_type="expval"))
    p.add_(0.4)  # an optimizer-style in-place step on the tensor the model holds by reference
    assert np.allclose(model.params, 0.7, atol=1e-6)
    b = np.asarray(model(execution_type="expval"))
    assert np.abs(a - b).max() > 1e-3
    new = np.full(snap.shape, 0.3)
    model.params = new
    assert np.allclose(np.asarray(model(execution_type="expval")), a, atol=1e-6)


def test_large_results_come_back_through_the_pinned_buffer_unchanged():
    """Results of >= 1 MiB are copied into a page-locked buffer and returned as its numpy view (round 5,
    ``_native.to_host``): same values as the pageable copy, writable, and independent of the next call's
    result (the host block is reused only after the array is dropped)."""
    import torch

    from qml_essentials_amd import _native as N
    from qml_essentials_amd.model import Model

    t = torch.randn((3, 1 << 17, 2), device="cuda")
    big = N.to_host(t)
    assert big.nbytes >= N.PINNED_RESULT_BYTES and np.array_equal(big, t.cpu().numpy()) and big.flags.writeable
    small = N.to_host(t[0, :8])
    assert np.array_equal(small, t[0, :8].cpu().numpy())
    model = Model(n_qubits=17, n_layers=1, circuit_type="Hardware_Efficient")
    a = model(execution_type="state")
    keep = a.copy()
    b = model(params=np.asarray(model.params) + 0.1, execution_type="state")
    assert a.shape == (1 << 17,) and np.array_equal(a, keep) and not np.allclose(a, b)
    assert abs(np.vdot(a, a).real - 1.0) < 1e-5
