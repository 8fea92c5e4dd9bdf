"""Host-side FCC statistics against scipy / explicit definitions --
the checks of the reference's ``tests/test_coefficients.py:743-952`` (TestFCC)."""
import numpy as np
import pytest
from scipy.stats import pearsonr, spearmanr

from qml_essentials_amd.coefficients import FCC


def test_pearson_and_spearman_match_scipy():
    rng = np.random.default_rng(1000)
    coeffs = rng.normal(size=(1000, 5))
    pear, spear = FCC._pearson(coeffs), FCC._spearman(coeffs)
    for i in range(5):
        for j in range(5):
            assert np.isclose(pear[i, j], pearsonr(coeffs[:, i], coeffs[:, j])[0], atol=1e-9)
            assert np.isclose(spear[i, j], spearmanr(coeffs[:, i], coeffs[:, j])[0], atol=1e-9)


def test_complex_input_is_stacked():
    rng = np.random.default_rng(42)
    coeffs = rng.normal(size=(1000, 5)) + 1j * rng.normal(size=(1000, 5))
    stacked = np.concatenate([coeffs.real, coeffs.imag], axis=0)
    pear, spear = FCC._pearson(coeffs), FCC._spearman(coeffs)
    for i in range(5):
        for j in range(5):
            assert np.isclose(pear[i, j], pearsonr(stacked[:, i], stacked[:, j])[0], atol=1e-9)
            assert np.isclose(spear[i, j], spearmanr(stacked[:, i], stacked[:, j])[0], atol=1e-9)
    assert not np.allclose(pear, FCC._pearson(coeffs.real), atol=1e-3)


def _with_holes():
    rng = np.random.default_rng(314)
    coeffs = rng.normal(size=(1000, 5)) + 1j * rng.normal(size=(1000, 5))
    coeffs[0, 1] = np.nan + 0.0j
    coeffs[1, 2] = np.inf + 0.0j
    return coeffs


def test_complex_pearson_and_covariance_pairwise_complete():
    coeffs = _with_holes()
    cp, cov = FCC._complex_pearson(coeffs), FCC._covariance(coeffs)
    assert np.allclose(cov, FCC._correlate(coeffs, method="covariance"), equal_nan=True)
    for i in range(5):
        for j in range(5):
            ok = np.isfinite(coeffs[:, i]) & np.isfinite(coeffs[:, j])
            x, y = coeffs[ok, i], coeffs[ok, j]
            xc, yc = x - x.mean(), y - y.mean()
            den = np.sqrt(np.sum(np.abs(xc) ** 2) * np.sum(np.abs(yc) ** 2))
            assert np.isclose(cp[i, j], np.sum(np.conj(xc) * yc) / den, atol=1e-9)
            assert np.isclose(cov[i, j], np.sum(np.conj(xc) * yc) / (len(x) - 1), atol=1e-9)


def test_complex_pearson_phase():
    rng = np.random.default_rng(2718)
    x = rng.normal(size=200) + 1j * rng.normal(size=200)
    coeffs = np.stack([x, np.exp(0.37j) * x], axis=1)
    corr = FCC._complex_pearson(coeffs)
    assert np.allclose(corr, FCC._correlate(coeffs, method="complex_pearson"))
    assert np.isclose(abs(corr[0, 1]), 1.0) and np.isclose(np.angle(corr[0, 1]), 0.37)
    assert np.isclose(np.angle(corr[1, 0]), -0.37)


def test_degenerate_columns_and_minp():
    mat = np.array([[1.0, 2.0, np.nan], [1.0, 3.0, np.nan], [1.0, 5.0, 4.0]])
    r = FCC._pearson(mat)
    assert np.isnan(r[0, 1]) and np.isnan(r[0, 0]) and np.isclose(r[1, 1], 1.0)
    assert np.all(np.isnan(FCC._pearson(mat, minp=4)))
    assert np.all(np.isnan(FCC._spearman(mat, minp=4)))
    with pytest.raises(ValueError, match="Unknown correlation method"):
        FCC._correlate(mat, method="kendall")


def test_weighting():
    fp = np.arange(16, dtype=float).reshape(4, 4)
    coeffs = np.array([[[1.0, 3.0], [-2.0, 4.0]], [[5.0, 7.0], [8.0, 10.0]]])
    m = np.abs(np.mean(coeffs, axis=-1)).T.reshape(-1)
    assert np.allclose(FCC._weighting_mean(fp, coeffs), fp * np.outer(m, m))
    w = FCC._weighting_linear(np.ones((5, 5)))
    assert np.isclose(w[2, 2], 1.0) and np.isclose(w[0, 0], 0.0) and np.isclose(w[0, 2], 0.5)
    with pytest.raises(AssertionError, match="odd dimensions"):
        FCC._weighting_linear(np.ones((4, 4)))


def test_mask_and_flat_frequencies():
    f1 = np.array([-2.0, -1.0, 0.0, 1.0, 2.0])
    assert FCC._calculate_mask(f1).tolist() == [2, 3, 4]
    assert np.array_equal(FCC._flat_frequencies(f1), f1)
    f2 = np.array([[-1.0, 0.0, 1.0], [-1.0, 0.0, 1.0]])
    keep = FCC._calculate_mask(f2)
    flat = FCC._flat_frequencies(f2)
    assert flat.shape == (9, 2) and keep.tolist() == [4, 5, 7, 8]
    assert flat[keep].tolist() == [[0, 0], [0, 1], [1, 0], [1, 1]]
    assert FCC.calculate_fcc(np.array([[np.nan, -0.5], [0.25, np.nan]])) == 0.375


@pytest.mark.parametrize("n_feat,strategy,cmin,cmax,zero", [
    (2, "hamming", 0.0, 1.0, False), (3, "hamming", 0.0, 1.0, False),
    (1, "hamming", 0.1, 0.9, False), (1, "hamming", 0.0, 1.0, True),
    (1, "binary", 0.0, 1.0, False), (1, "ternary", 0.0, 1.0, False)])
def test_fourier_series_dataset(n_feat, strategy, cmin, cmax, zero):
    """test_coefficients.py:1205-1330: shapes, FFT consistency, symmetry and coefficient range
    of ``Datasets.generate_fourier_series`` (host only)."""
    from qml_essentials_amd.ansaetze import Encoding
    from qml_essentials_amd.coefficients import Datasets
    from qml_essentials_amd.model import Model
    from qml_essentials_amd.utils import key

    model = Model(n_qubits=2, n_layers=1, encoding=Encoding(strategy, ["RY"] * n_feat))
    k = key(1000)
    for _ in range(20):
        x, f, c = Datasets.generate_fourier_series(k, model=model, coefficients_min=cmin,
                                                   coefficients_max=cmax, zero_centered=zero)
        k, _ = k.split()
        c_hat = np.fft.fftshift(np.fft.fftn(f, axes=list(range(model.n_input_feat))))
        assert np.allclose(c, c_hat, atol=1e-6)
        assert x.shape == (*model.degree, model.n_input_feat)
        assert f.shape == tuple(model.degree) == c.shape
        flat = c.reshape(-1)
        assert np.allclose(flat, np.conj(flat[::-1]), atol=1e-12)       # real-valued series
        mid = flat[flat.size // 2]
        assert abs(mid.imag) < 1e-12 and (not zero or mid == 0)
        mags = np.abs(np.delete(flat, flat.size // 2)) ** 2
        assert mags.min() >= cmin - 1e-9 and mags.max() <= cmax + 1e-9
