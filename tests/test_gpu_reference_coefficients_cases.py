"""The reference's ``tests/test_coefficients.py`` numerical-FFT cases (TestCoefficients) with
its own models and assertions (FourierTree cases are out of scope)."""
import numpy as np
import pytest

from qml_essentials_amd.ansaetze import Encoding
from qml_essentials_amd.coefficients import Coefficients
from qml_essentials_amd.model import Model
from qml_essentials_amd.utils import key

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("circuit_type,n_qubits,n_layers,output_qubit", [
    ("Circuit_1", 3, 1, [0, 1]), ("Circuit_9", 4, 1, 0), ("Circuit_19", 5, 1, 0)])
def test_coefficients_reproduce_the_model(circuit_type, n_qubits, n_layers, output_qubit):
    """test_coefficients.py:25-70: the Fourier series rebuilt from the spectrum equals the model."""
    model = Model(n_qubits=n_qubits, n_layers=n_layers, circuit_type=circuit_type,
                  output_qubit=output_qubit)
    coeffs, freqs = Coefficients.get_spectrum(model)
    assert coeffs.shape == model.degree
    ref = np.linspace(-np.pi, np.pi, 10)
    exp_model = model(params=None, inputs=ref, force_mean=True)
    exp_fourier = Coefficients.evaluate_Fourier_series(coefficients=coeffs, frequencies=freqs,
                                                       inputs=ref)
    assert np.allclose(exp_model, exp_fourier, atol=1e-5)


@pytest.mark.parametrize("output_qubit,output_size,force_mean", [
    (-1, 1, True), ([0, 1], 1, True), (-1, 3, False), ([0, 1], 2, False)])
def test_multi_dim_input(output_qubit, output_size, force_mean):
    """test_coefficients.py:120-157."""
    model = Model(n_qubits=3, n_layers=1, circuit_type="Hardware_Efficient",
                  output_qubit=output_qubit, encoding=["RX", "RY"],
                  data_reupload=[[[1, 0], [1, 0], [1, 1]]])
    coeffs, freqs = Coefficients.get_spectrum(model, force_mean=force_mean)
    assert coeffs.shape == tuple(model.degree) or coeffs.shape == (*model.degree, output_size)
    ref = np.array([1, 2, 3, 4])
    exp_model = model(params=None, inputs=ref, force_mean=force_mean)
    exp_fourier = Coefficients.evaluate_Fourier_series(coefficients=coeffs, frequencies=freqs,
                                                       inputs=ref)
    assert np.isclose(exp_model, exp_fourier, atol=1e-5).all()


def test_batch():
    """test_coefficients.py:187-232: parameter batches == one spectrum per parameter set."""
    model = Model(n_qubits=2, n_layers=1, circuit_type="Circuit_15", output_qubit=-1)
    model.initialize_params(key(1000), repeat=3)
    params = model.params
    par, _ = Coefficients.get_spectrum(model, shift=True, trim=True)
    for i in range(3):
        model.params = params[i]
        single, _ = Coefficients.get_spectrum(model, params=params[i], shift=True, trim=True)
        assert np.allclose(par[:, i], single, rtol=1e-5, atol=1e-7)
    model = Model(n_qubits=2, n_layers=1, circuit_type="Circuit_19", output_qubit=-1,
                  encoding=["RX", "RY"])
    model.initialize_params(key(1001), repeat=3)
    params = model.params
    par, _ = Coefficients.get_spectrum(model, shift=True, trim=True)
    for i in range(3):
        single, _ = Coefficients.get_spectrum(model, params=params[i], shift=True, trim=True)
        assert np.allclose(par[:, :, i], single, rtol=1e-5, atol=1e-7)


def test_oversampling_shift_trim_psd():
    """test_coefficients.py:235-286,337-344."""
    model = Model(n_qubits=2, n_layers=1, circuit_type="Circuit_19")
    assert Coefficients.get_spectrum(model, mts=3)[0].shape[0] == 15
    assert Coefficients.get_spectrum(model, mfs=3)[0].shape[0] == 15
    coeffs, _ = Coefficients.get_spectrum(model, shift=True)
    assert np.allclose(np.abs(coeffs), np.abs(coeffs[::-1]), atol=1e-7)
    assert Coefficients.get_psd(coeffs).shape == coeffs.shape
    model = Model(n_qubits=3, n_layers=1, circuit_type="Hardware_Efficient", output_qubit=-1)
    full, _ = Coefficients.get_spectrum(model, mts=2, trim=False)
    trimmed, _ = Coefficients.get_spectrum(model, mts=2, trim=True)
    assert full.size - 1 == trimmed.size


def test_frequencies():
    """test_coefficients.py:289-334."""
    model = Model(n_qubits=2, n_layers=1, circuit_type="Circuit_19")
    coeffs, freqs = Coefficients.get_spectrum(model)
    assert np.shape(freqs) == coeffs.shape
    model = Model(n_qubits=2, n_layers=1, circuit_type="Circuit_19", encoding=["RX", "RY"])
    coeffs, freqs = Coefficients.get_spectrum(model)
    assert np.size(freqs[0]) * np.size(freqs[1]) == coeffs.size
    model = Model(n_qubits=2, n_layers=2, circuit_type="Circuit_19", encoding=["RX", "RY"],
                  data_reupload=[[[True, True], [False, True]], [[False, True], [True, True]]])
    coeffs, freqs = Coefficients.get_spectrum(model)
    assert np.size(freqs[0]) * np.size(freqs[1]) == coeffs.size


def test_numerical_cap_trims_spectrum():
    """test_coefficients.py:347-405."""
    model = Model(n_qubits=3, n_layers=3, circuit_type="Strongly_Entangling",
                  encoding=Encoding("hamming", "RZ"), output_qubit=-1)
    model.initialize_params(key(1000), repeat=20)
    c_full, f_full = Coefficients.get_spectrum(model, shift=True, trim=True, numerical_cap=-1)
    per_freq_max = np.max(np.abs(c_full), axis=tuple(range(1, c_full.ndim)))
    cap = float(np.median(per_freq_max))
    c_cap, f_cap = Coefficients.get_spectrum(model, shift=True, trim=True, numerical_cap=cap)
    assert c_cap.shape[0] == f_cap.shape[0] < f_full.shape[0]
    assert np.array_equal(np.sort(f_cap), np.sort(f_full[per_freq_max >= cap]))
    assert np.all(np.any(c_cap != 0, axis=tuple(range(1, c_cap.ndim))))
    assert np.array_equal(np.sort(f_cap), np.sort(-f_cap))
