"""Pin the CPU oracle against the reference's own known-answer tests (CPU only).

Every test re-states a test of ``/root/reference/tests`` (file:line cited) with
the oracle in place of jaqsi; the reference cannot be imported here.
"""
import json
import os

import numpy as np
import pytest

from oracle import analysis as A
from oracle import circuits as C
from oracle import dense as D
from oracle import einsum_sim as E
from oracle import gates as G

c128 = np.complex128
BELL = [("H", [0], ()), ("CX", [0, 1], ())]
GHZ3 = [("H", [0], ()), ("CX", [0, 1], ()), ("CX", [1, 2], ())]
GHZ4 = GHZ3 + [("CX", [2, 3], ())]


def probs(tape, n):
    return E.simulate_and_measure(tape, n, "probs", dtype=c128)


def expz(tape, n, wires):
    return E.simulate_and_measure(tape, n, "expval", [("PauliZ", [w]) for w in wires], c128)


# ---- indexing / wire order ------------------------------------------------
def test_probs_bell():  # test_jaqsi.py:365-369
    assert np.allclose(probs(BELL, 2), [0.5, 0, 0, 0.5], atol=1e-10)


def test_probs_ghz3_ghz4():  # test_jaqsi.py:383-388, 399-404
    e3 = np.zeros(8)
    e3[[0, 7]] = 0.5
    e4 = np.zeros(16)
    e4[[0, 15]] = 0.5
    assert np.allclose(probs(GHZ3, 3), e3, atol=1e-10)
    assert np.allclose(probs(GHZ4, 4), e4, atol=1e-10)


def test_ghz_toffoli_equals_cnot_chain():  # test_jaqsi.py:407-413
    toff = [("H", [0], ()), ("CX", [0, 1], ()), ("CCX", [0, 1, 2], ())]
    assert np.allclose(probs(GHZ3, 3), probs(toff, 3), atol=1e-10)


def test_non_adjacent_cx_msb_convention():  # test_jaqsi.py:416-427
    p = probs([("H", [0], ()), ("CX", [0, 2], ())], 3)
    e = np.zeros(8)
    e[[0, 5]] = 0.5
    assert np.allclose(p, e, atol=1e-10)


# ---- values ---------------------------------------------------------------
def test_expval_bell_x_and_z():  # test_jaqsi.py:358-362
    for name in ("PauliX", "PauliZ"):
        r = E.simulate_and_measure(BELL, 2, "expval", [(name, [0]), (name, [1])], c128)
        assert np.allclose(r, 0, atol=1e-10)


def test_rx_expval_is_cos():  # test_jaqsi.py:372-380, 728-746
    for th in np.linspace(0, np.pi, 9):
        assert np.isclose(expz([("RX", [0], (th,))], 1, [0])[0], np.cos(th), atol=1e-10)


def test_rx_0_and_pi_probs():  # test_jaqsi.py:749-761
    assert np.allclose(probs([("RX", [0], (0.0,))], 1), [1, 0], atol=1e-10)
    assert np.allclose(probs([("RX", [0], (np.pi,))], 1), [0, 1], atol=1e-10)


def test_two_arg_broadcast_cos_sum():  # test_jaqsi.py:789-821
    for th, ph in [(0.3, 0.9), (1.2, 2.0)]:
        t = [("RX", [0], (th,)), ("RX", [0], (ph,))]
        assert np.isclose(expz(t, 1, [0])[0], np.cos(th + ph), atol=1e-10)


def test_ghz_expval_z_zero():  # test_jaqsi.py:391-396
    assert np.allclose(expz(GHZ3, 3, [0, 1, 2]), 0, atol=1e-10)


def test_density_is_projector_and_consistent():  # test_jaqsi.py:430-491
    t = [("RY", [0], (0.4,)), ("CRX", [0, 1], (1.1,)), ("RZ", [1], (0.3,))]
    rho = E.simulate_and_measure(t, 2, "density", dtype=c128)
    psi = E.simulate_pure(t, 2, c128)
    assert np.allclose(rho, np.outer(psi, psi.conj()), atol=1e-12)
    assert np.allclose(rho, rho.conj().T, atol=1e-12)
    assert np.allclose(np.real(np.diag(rho)), probs(t, 2), atol=1e-12)
    z0 = D.lift(G.Z, [0], 2)
    assert np.isclose(np.real(np.trace(z0 @ rho)), expz(t, 2, [0])[0], atol=1e-12)


def test_rot_equals_rz_ry_rz_sequence():  # test_jaqsi.py:558-584
    phi, theta, omega = 0.7, 1.5, 0.3
    a = E.simulate_pure([("Rot", [0], (phi, theta, omega))], 1, c128)
    b = E.simulate_pure(
        [("RZ", [0], (phi,)), ("RY", [0], (theta,)), ("RZ", [0], (omega,))], 1, c128
    )
    phase = a[0] / b[0]
    assert np.allclose(a, phase * b, atol=1e-10)


def test_cphase_properties():  # test_ansaetze.py:472-552
    assert np.allclose(G.cphase(np.pi), G.matrix("CZ"), atol=1e-12)
    assert np.allclose(G.cphase(0.0), np.eye(4), atol=1e-12)
    assert np.allclose(G.cphase(0.37), np.diag([1, 1, 1, np.exp(0.37j)]), atol=1e-12)
    # <Z> on |11> stays -1
    t = [("PauliX", [0], ()), ("PauliX", [1], ()), ("CPhase", [0, 1], (0.8,))]
    assert np.allclose(expz(t, 2, [0, 1]), [-1, -1], atol=1e-10)
    # phase kick-back: H(0) X(1) CPhase(phi) H(0): <Z0> = cos(phi)
    for phi, want in [(0.0, 1.0), (np.pi, -1.0), (np.pi / 2, 0.0)]:
        t = [("H", [0], ()), ("PauliX", [1], ()), ("CPhase", [0, 1], (phi,)), ("H", [0], ())]
        assert np.isclose(expz(t, 2, [0])[0], want, atol=1e-10)


def test_rx_arccos_model_expval():  # test_model.py:96-103
    for x in (-0.7, 0.0, 0.25, 0.9):
        assert np.isclose(expz([("RX", [0], (np.arccos(x),))], 1, [0])[0], x, atol=1e-10)


def test_controlled_rotations_against_dense():  # content of test_jaqsi.py:494-555
    rng = np.random.default_rng(3)
    for name in ("CY", "CZ", "CRX", "CRY", "CRZ"):
        for wires in ([0, 1], [1, 0], [2, 0], [0, 2]):
            p = (float(rng.uniform(0, 2 * np.pi)),) if name.startswith("CR") else ()
            t = [("H", [0], ()), ("RY", [1], (0.4,)), ("RX", [2], (1.3,)), (name, wires, p)]
            assert np.allclose(E.simulate_pure(t, 3, c128), D.simulate(t, 3), atol=1e-12)


# ---- partial trace / marginals ---------------------------------------------
def test_partial_trace_cases():  # test_jaqsi.py:862-894
    rho = E.simulate_and_measure(BELL, 2, "density", dtype=c128)
    for keep in ([0], [1]):
        assert np.allclose(A.partial_trace(rho, 2, keep), 0.5 * np.eye(2), atol=1e-10)
    rho2 = E.simulate_and_measure([("H", [1], ())], 2, "density", dtype=c128)
    assert np.allclose(A.partial_trace(rho2, 2, [1]), 0.5 * np.ones((2, 2)), atol=1e-10)
    assert np.allclose(A.partial_trace(rho, 2, [0, 1]), rho, atol=1e-10)


def test_marginalize_probs():  # test_jaqsi.py:916-950
    p = probs(BELL, 2)
    assert np.allclose(A.marginalize_probs(p, 2, [0])[0], [0.5, 0.5], atol=1e-10)
    assert np.allclose(A.marginalize_probs(p, 2, [0, 1])[0], p, atol=1e-10)
    pb = np.stack([probs([("RX", [0], (t,)), ("H", [1], ())], 2) for t in (0.0, np.pi / 2)])
    m = A.marginalize_probs(pb, 2, [0])
    assert m.shape == (2, 2) and np.allclose(m.sum(axis=1), 1, atol=1e-10)


# ---- front-end golden: topologies from the real reference module -------------
def test_topologies_match_reference_fixture(golden_dir):
    fx = json.load(open(os.path.join(golden_dir, "topologies.json")))
    sym = {"n-1": lambda n: n - 1, "n-2": lambda n: n - 2, "n//2": lambda n: n // 2}
    for key, rec in fx["calls"].items():
        fn = getattr(C, rec["topology"])
        kw = {k: (sym[v] if isinstance(v, str) else v) for k, v in rec["kwargs"].items()}
        for n_s, want in rec["pairs"].items():
            got = [list(p) for p in fn(int(n_s), **kw)]
            assert got == want, (key, n_s)


def test_structures_use_fixture_topologies(golden_dir):
    """Every entangling block of oracle.circuits.STRUCTURES reproduces its fixture."""
    fx = json.load(open(os.path.join(golden_dir, "topologies.json")))["calls"]
    for key, rec in fx.items():
        ansatz, idx = key.split(".")
        if ansatz == "GHZ":
            continue
        block = C.STRUCTURES[ansatz][int(idx)]
        for n in range(2, 9):
            assert [list(p) for p in block[1](n, **block[2])] == rec["pairs"][str(n)], key


def test_param_counts_and_degrees():
    # ansaetze.py:286-303; HE: 3n rotations, Circuit_19: 2n + n CRX (n >= 3)
    assert C.n_params_per_layer("Hardware_Efficient", 4) == 12
    assert C.n_params_per_layer("Circuit_19", 4) == 12
    assert C.n_params_per_layer("Strongly_Entangling", 4) == 24
    assert C.n_params_per_layer("Circuit_9", 4) == 4
    assert C.n_params_per_layer("No_Ansatz", 3) == 0
    # test_model.py:797-836 style degrees: hamming full DRU -> 2*L*n+1
    assert C.ModelSpec(2, 1, "Circuit_19").degree == (5,)
    assert C.ModelSpec(10, 6, "Hardware_Efficient").degree == (121,)
    s = C.ModelSpec(4, 2, "Hardware_Efficient", data_reupload=False)
    assert s.degree == (3,) and not s.has_dru and s.impl_n_layers == 2  # Appendix B
    assert C.ModelSpec(2, 1, "Circuit_19", encoding=["RX", "RY"]).degree == (5, 5)
    assert C.ModelSpec(2, 1, "Circuit_19", strategy="binary").degree == (7,)
    assert C.ModelSpec(2, 1, "Circuit_19", strategy="ternary").degree == (9,)


def test_model_gate_counts_survey_8a():
    # cfg1 = 56 ops (48 with zero input), cfg2 = 480, cfg4 = 340
    def count(spec, x):
        rng = np.random.default_rng(0)
        t = C.model_tape(spec, rng.uniform(0, 6, spec.params_shape), [x])
        return sum(1 for g in t if g[0] != "Barrier")

    s1 = C.ModelSpec(4, 2, "Hardware_Efficient")
    assert count(s1, 0.5) == 56 and count(s1, 0.0) == 48
    assert count(C.ModelSpec(20, 4, "Hardware_Efficient"), 0.5) == 480
    assert count(C.ModelSpec(10, 6, "Hardware_Efficient"), 0.5) == 340


# ---- analysis: reference form == pure-state form ------------------------------
def _random_states(n, count, seed):
    rng = np.random.default_rng(seed)
    spec = C.ModelSpec(n, 2, "Hardware_Efficient")
    out = []
    for _ in range(count):
        t = C.model_tape(spec, rng.uniform(0, 2 * np.pi, spec.params_shape), [0.3])
        out.append(E.simulate_pure(t, n, c128))
    return np.array(out)


def test_fidelity_reference_form_equals_overlap():
    st = _random_states(3, 8, 1)
    rhos = np.array([np.outer(s, s.conj()) for s in st])
    assert np.allclose(A.fidelities_reference_form(rhos, 4), A.fidelities_pure(st, 4), atol=1e-6)


def test_meyer_wallach_reference_form_equals_purity_formula():
    for n in (2, 3, 4):
        for s in _random_states(n, 3, n):
            rho = np.outer(s, s.conj())
            assert np.isclose(
                A.meyer_wallach_reference_form(rho, n), A.meyer_wallach_pure(s, n), atol=1e-10
            )


def test_meyer_wallach_exact_anchors():  # test_entanglement.py:16-96: C1 -> 0, C9 -> 1
    rng = np.random.default_rng(5)
    for ansatz, want in (("Circuit_1", 0.0), ("Circuit_9", 1.0)):
        spec = C.ModelSpec(4, 1, ansatz, data_reupload=False)
        for _ in range(4):
            t = C.model_tape(spec, rng.uniform(0, 2 * np.pi, spec.params_shape), [0.0])
            s = E.simulate_pure(t, 4, c128)
            assert np.isclose(A.meyer_wallach_pure(s, 4), want, atol=1e-10)


def test_haar_kl_self_is_zero():  # test_expressiblity.py:83-111
    h = A.haar_integral(2, 10)
    assert np.isclose(A.kl_divergence(h, h).mean(), 0.0, atol=1e-3)
    # closed form of the bin mass
    N = 4
    edges = np.linspace(0, 1, 11)
    assert np.allclose(h, (1 - edges[:-1]) ** (N - 1) - (1 - edges[1:]) ** (N - 1), atol=1e-10)


def test_fourier_series_reevaluation():  # test_coefficients.py:59-70,148-158
    spec = C.ModelSpec(2, 1, "Circuit_19")
    rng = np.random.default_rng(7)
    params = rng.uniform(0, 2 * np.pi, spec.params_shape)

    def model(x):
        t = C.model_tape(spec, params, [x])
        return expz(t, 2, [0, 1]).mean()

    axes, grid, n_freqs = A.fourier_grid(spec.degree)
    outs = np.array([model(x[0]) for x in grid])
    coeffs, freqs = A.fourier_transform(outs, axes, n_freqs)
    assert abs(np.sum(coeffs).imag) < 1e-6  # coefficients.py:67-71
    for x in (0.1, 1.7, 4.0):
        series = np.real(np.sum(coeffs * np.exp(1j * freqs[0] * x)))
        assert np.isclose(series, model(x), atol=1e-5)
    assert np.allclose(np.abs(coeffs[1:]), np.abs(coeffs[1:][::-1]), atol=1e-10)  # :259-269
