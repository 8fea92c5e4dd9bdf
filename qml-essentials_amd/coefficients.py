"""Numerical Fourier coefficients of a model (``Coefficients``, FFT path).

API mirror of ``qml_essentials/coefficients.py:23-237``: ``get_spectrum``,
``_fourier_transform``, ``get_psd``, ``evaluate_Fourier_series``.  The cost is the
batched model evaluation on the input grid (``:130``) -- that is the HIP engine's
job, sharded across ranks by :class:`script.Script`; the FFT of the few thousand
expectation values is done with NumPy on the host.

``FCC`` (Fourier-coefficient correlation, ``coefficients.py:966-1650``) sits downstream:
its cost is again the ``B_I x B_P`` batch of circuit evaluations (grid points x parameter
samples) on the engine; the correlation of the resulting ``(n_freq, n_samples)`` matrix is
small dense linear algebra done here on the host.  ``Datasets`` generates the synthetic
Fourier-series targets the reference trains on (``coefficients.py:1652-1788``).
``FourierTree`` (analytic) is out of scope.
"""
from __future__ import annotations

import logging
import math
from typing import Any, List, Optional, Tuple, Union

import numpy as np
from scipy.stats import rankdata

from .model import Model

log = logging.getLogger(__name__)


class Coefficients:
    @classmethod
    def get_spectrum(cls, model: Model, mfs: int = 1, mts: int = 1, shift: bool = False,
                     trim: bool = False, numerical_cap: Optional[float] = -1,
                     **kwargs) -> Tuple[np.ndarray, np.ndarray]:
        """FFT coefficients and their frequencies; ``mfs`` / ``mts`` oversample in
        frequency / time (``coefficients.py:25-106``)."""
        kwargs.setdefault("force_mean", True)
        kwargs.setdefault("execution_type", "expval")
        coeffs, freqs = cls._fourier_transform(model, mfs=mfs, mts=mts, **kwargs)
        imag = float(np.sum(coeffs).imag)
        if not abs(imag) <= 1.0e-6:  # (= np.isclose(imag, 0.0, atol=1e-6), without its 8 us; NaN fails too)
            raise ValueError(f"Spectrum is not real. Imaginary part of coefficients is: {imag}")
        if trim:
            for ax in range(model.n_input_feat):
                if coeffs.shape[ax] % 2 == 0:
                    coeffs = np.delete(coeffs, len(coeffs) // 2, axis=ax)
                    freqs = [np.delete(f, len(f) // 2, axis=ax) for f in freqs]
        if shift:
            coeffs = np.fft.fftshift(coeffs, axes=list(range(model.n_input_feat)))
            freqs = np.fft.fftshift(freqs)
        if numerical_cap is not None and numerical_cap > 0:
            coeffs = np.where(np.abs(coeffs) < numerical_cap, np.zeros_like(coeffs), coeffs)
            if model.n_input_feat == 1:
                alive = coeffs != 0 if coeffs.ndim == 1 else np.any(
                    coeffs != 0, axis=tuple(range(1, coeffs.ndim)))
                coeffs = coeffs[alive]
                freqs = [np.asarray(freqs[0])[alive]]
        if len(freqs) == 1:
            freqs = freqs[0]
        return coeffs, freqs

    @classmethod
    def _fourier_transform(cls, model: Model, mfs: int, mts: int, **kwargs: Any):
        """Sample the model on ``x_k = 2 pi k / N_f`` per feature and FFT
        (``coefficients.py:109-150``; feature 0 is the slowest grid axis)."""
        F = model.n_input_feat
        # the engine returns float32; the (tiny) host FFT runs in double so that it adds no
        # rounding noise of its own to the spectrum.  complex128 mode (utils.enable_x64 /
        # Model(x64=True)): the grid stays float64 too -- a float32-rounded grid point is off by
        # 1e-7 rad, which leaks 1e-8 into the analytically vanishing top-frequency coefficients
        from .utils import x64_enabled

        x64 = x64_enabled() if getattr(model, "x64", None) is None else bool(model.x64)
        # grid, frequencies and the device copy of the grid depend on (degrees, mfs, mts) only: a
        # spectrum is asked for again and again on the same grid, and the ~40 us of numpy calls that
        # build it are a fifth of a 4096-point call
        key = (tuple(mfs * model.degree[i] for i in range(F)), mts, x64)
        hit = cls._PLANS.get(key)
        if hit is None:
            n_freqs = np.array(key[0])
            axes = [np.arange(0, 2 * mts * np.pi, 2 * np.pi / n_freqs[i]) for i in range(F)]
            grid = np.array(np.meshgrid(*axes)).T.reshape(-1, F)
            freqs = [np.fft.fftfreq(int(mts * n_freqs[i]), 1 / n_freqs[i]) for i in range(F)]
            lens = [a.shape[0] for a in axes]
            if len(cls._PLANS) > 32:
                cls._PLANS.clear()
            hit = cls._PLANS[key] = (grid, freqs, lens)
        grid, freqs, lens = hit
        freqs = [f.copy() for f in freqs]  # (callers may edit what they get back)
        grid_dev = None if x64 else cls._device_grid(grid)
        if grid_dev is not None:
            # the grid lives on the GPU (cached per grid): nothing is uploaded per call, and the
            # ONE device -> host copy is the (few thousand) model values.  Their FFT stays on the
            # host in float64: numpy transforms 4096 points in ~25 us, a device FFT costs twice that
            # in launch bookkeeping alone, and numpy's real-input transform is exactly Hermitian
            # (|c_k| == |c_-k| bit for bit, which numerical_cap relies on)
            outputs = model(inputs=grid_dev, **kwargs)
            if hasattr(outputs, "is_cuda"):
                outputs = outputs.cpu().numpy()
        else:
            outputs = model(inputs=grid if x64 else grid.astype(np.float32), **kwargs)
        outputs = np.asarray(outputs, dtype=np.float64).reshape(*lens, -1).squeeze()
        if F == 1 and outputs.ndim >= 1 and outputs.shape[0] == lens[0] > 2:
            return cls._fft_real(outputs), freqs
        coeffs = np.fft.fftn(outputs, axes=list(range(F)))
        return coeffs / math.prod(outputs.shape[0:F]), freqs

    @staticmethod
    def _fft_real(values: np.ndarray) -> np.ndarray:
        """``fft(values, axis=0) / N`` of REAL float64 samples: the half spectrum (``rfft``) and its
        conjugate mirror -- exactly Hermitian by construction and half the work of the complex
        transform (4096 points: 16 us instead of 30 on the bench host)."""
        n = values.shape[0]
        half = np.fft.rfft(values, axis=0, norm="forward")
        out = np.empty((n,) + half.shape[1:], dtype=np.complex128)
        out[:n // 2 + 1] = half
        out[n // 2 + 1:] = np.conj(half[1:(n + 1) // 2][::-1])
        return out

    _GRIDS: dict = {}
    _PLANS: dict = {}  # (n_freqs, mts, x64) -> (host grid, frequency axes, grid lengths)

    @classmethod
    def _device_grid(cls, grid: np.ndarray):
        """float32 CUDA copy of an input grid, cached (a spectrum is usually asked for again and
        again on the same grid); None without a GPU."""
        from .utils import _gpu_present

        if not _gpu_present():
            return None
        import torch

        key = (grid.shape, float(grid[-1].sum()), float(grid[len(grid) // 2].sum()), torch.cuda.current_device())
        hit = cls._GRIDS.get(key)
        if hit is None or (hit[0] is not grid and not np.array_equal(hit[0], grid)):
            if len(cls._GRIDS) > 16:
                cls._GRIDS.clear()
            hit = cls._GRIDS[key] = (grid, torch.from_numpy(grid.astype(np.float32)).cuda())
        return hit[1]

    @classmethod
    def get_psd(cls, coeffs: np.ndarray) -> np.ndarray:
        coeffs = np.asarray(coeffs)
        return (2.0 / (len(coeffs) ** 2)) * (coeffs.real**2 + coeffs.imag**2)

    @classmethod
    def evaluate_Fourier_series(cls, coefficients, frequencies,
                                inputs: Union[np.ndarray, list, float]) -> np.ndarray:
        """sum_k c_k exp(i <w_k, x>) at the given points (``coefficients.py:172-237``)."""
        coefficients = np.asarray(coefficients)

        def flatten(freq_axes):
            freq_axes = [np.asarray(f) for f in freq_axes]
            mesh = np.stack(np.meshgrid(*freq_axes, indexing="ij"), axis=-1)
            ff = mesh.reshape(-1, len(freq_axes))
            return coefficients.reshape(ff.shape[0], *coefficients.shape[len(freq_axes):]), ff

        if isinstance(frequencies, list):
            fc, ff = flatten(frequencies)
        else:
            frequencies = np.asarray(frequencies)
            if frequencies.ndim == 1:
                ff = frequencies[:, None]
                fc = coefficients.reshape(ff.shape[0], *coefficients.shape[1:])
            else:
                n_feat, n_axis = frequencies.shape
                if coefficients.shape[:n_feat] == (n_axis,) * n_feat:
                    fc, ff = flatten(frequencies)
                else:
                    ff = frequencies
                    fc = coefficients.reshape(ff.shape[0], *coefficients.shape[1:])
        x = np.asarray(inputs)
        if x.ndim == 0:
            x = x.reshape(1, 1)
        elif x.ndim == 1:
            if ff.shape[1] == 1:
                x = x[:, None]
            elif x.shape[0] == ff.shape[1]:
                x = x[None, :]
            else:
                x = np.repeat(x[:, None], ff.shape[1], axis=1)
        phases = np.exp(1j * (x @ ff.T))
        return np.squeeze(np.real(np.tensordot(phases, fc, axes=([1], [0]))))


class _PairStats:
    """Pairwise-complete moments of the columns of ``mat`` (N observations x K variables):
    for every column pair (i, j) only rows finite in BOTH columns count -- the pandas
    ``corr`` convention the reference follows.  All sums are K x K matrix products."""

    def __init__(self, mat: np.ndarray) -> None:
        mat = np.asarray(mat)
        ok = np.isfinite(mat)
        x = np.where(ok, mat, 0)
        w = ok.astype(np.float64)
        self.nobs = w.T @ w
        n = np.where(self.nobs > 0, self.nobs, 1.0)
        sx = x.T @ w            # sum of column i over the rows valid for (i, j)
        sy = w.T @ x            # sum of column j over the same rows
        a2 = np.abs(x) ** 2
        self.sxy = np.conj(x).T @ x - np.conj(sx) * sy / n      # centred cross moment
        self.ssx = a2.T @ w - np.abs(sx) ** 2 / n
        self.ssy = w.T @ a2 - np.abs(sy) ** 2 / n

    def masked(self, result: np.ndarray, minp: int) -> np.ndarray:
        return np.where(self.nobs < minp, np.nan, result)


def _stack_complex(mat: np.ndarray) -> np.ndarray:
    """Complex samples count as two real samples (real parts, then imaginary parts)."""
    mat = np.asarray(mat)
    return np.concatenate([mat.real, mat.imag], axis=0) if np.iscomplexobj(mat) else mat


class FCC:
    """Fourier-coefficient correlation (arXiv:2508.20868); API of ``coefficients.py:966-1650``."""

    @classmethod
    def get_fcc(cls, model: Model, n_samples: int, random_key=None,
                method: Optional[str] = "pearson", scale: Optional[bool] = False,
                weight: Optional[bool] = False, trim_redundant: Optional[bool] = True,
                **kwargs) -> float:
        """Mean absolute correlation between the coefficients of different frequencies over
        ``n_samples`` random parameter sets (``coefficients.py:968-1044``)."""
        if trim_redundant and not weight:
            # positive-frequency block only; mean |r| over its strict lower triangle
            _, coeffs, freqs = cls._calculate_coefficients(model, n_samples, random_key, scale,
                                                           **kwargs)
            keep = cls._calculate_mask(freqs)
            block = np.abs(cls._correlate(coeffs.reshape(-1, coeffs.shape[-1])[keep].T,
                                          method=method))
            low = np.tril(np.ones(block.shape, dtype=bool), k=-1) & np.isfinite(block)
            # (the matrix is symmetric in magnitude: lower triangle == half the off-diagonal)
            return float(block[low].sum() / low.sum()) if low.any() else float("nan")
        fingerprint, _ = cls.get_fourier_fingerprint(
            model, n_samples, random_key, method, scale, weight, trim_redundant=trim_redundant,
            **kwargs)
        return cls.calculate_fcc(fingerprint)

    @classmethod
    def get_fourier_fingerprint(cls, model: Model, n_samples: int, random_key=None,
                                method: Optional[str] = "pearson", scale: Optional[bool] = False,
                                weight: Optional[bool] = False,
                                trim_redundant: Optional[bool] = True,
                                nan_to_one: Optional[bool] = False, **kwargs: Any):
        """Correlation matrix of the Fourier coefficients and its frequency labels
        (``coefficients.py:1047-1162``).  With ``trim_redundant`` only the strict lower
        triangle of the non-negative-frequency block survives and the labels are a
        ``(row_freqs, col_freqs)`` tuple."""
        _, coeffs, freqs = cls._calculate_coefficients(model, n_samples, random_key, scale,
                                                       **kwargs)
        keep = cls._calculate_mask(freqs) if trim_redundant else None
        if trim_redundant and not weight:
            fp = cls._correlate(coeffs.reshape(-1, coeffs.shape[-1])[keep].T, method=method)
        else:
            fp = cls._correlate(coeffs.transpose(), method=method)
        if nan_to_one:
            fp = np.where(np.isnan(fp), 1.0, fp)
        if weight:
            fp = cls._weighting_mean(fp, coeffs)
            if trim_redundant:
                fp = fp[keep][:, keep]
        if not trim_redundant:
            return fp, freqs
        labels = cls._flat_frequencies(freqs)[keep]
        fp = np.where(np.tril(np.ones(fp.shape, dtype=bool), k=-1), fp, np.nan)
        rows = np.any(np.isfinite(fp), axis=1)
        cols = np.any(np.isfinite(fp), axis=0)
        return fp[rows][:, cols], (labels[rows], labels[cols])

    @classmethod
    def calculate_fcc(cls, fourier_fingerprint: np.ndarray) -> float:
        """``nanmean(|fingerprint|)`` (``coefficients.py:1165-1180``)."""
        return float(np.nanmean(np.abs(fourier_fingerprint)))

    @classmethod
    def _calculate_mask(cls, freqs) -> np.ndarray:
        """Flat (C-order) indices of the coefficients whose frequency is non-negative on
        every input axis (``coefficients.py:1183-1229``)."""
        fa = np.asarray(freqs)
        if fa.ndim == 1:
            return np.flatnonzero(fa >= 0)
        nonneg = np.ones((), dtype=bool)
        for axis in fa:                       # outer "and" over the axes, C order
            nonneg = np.logical_and(nonneg[..., None], axis >= 0)
        return np.flatnonzero(nonneg.reshape(-1))

    @classmethod
    def _flat_frequencies(cls, freqs) -> np.ndarray:
        """Per-coefficient frequency labels in the same C order: the vector itself for one
        feature, ``(N, n_feat)`` tuples otherwise (``coefficients.py:1232-1254``)."""
        fa = np.asarray(freqs)
        if fa.ndim == 1:
            return fa
        return np.stack(np.meshgrid(*fa, indexing="ij"), axis=-1).reshape(-1, fa.shape[0])

    @classmethod
    def _calculate_coefficients(cls, model: Model, n_samples: int, random_key=None,
                                scale: bool = False, **kwargs: Any):
        """Re-draw ``n_samples`` (``x 2^n x n_features`` when ``scale``) parameter sets and
        return ``(params, coeffs, freqs)`` with the sample axis last
        (``coefficients.py:1257-1298``)."""
        if n_samples > 0:
            total = int(2**model.n_qubits * n_samples * model.n_input_feat) if scale else n_samples
            if scale:
                log.info("Using %d samples.", total)
            model.initialize_params(random_key, repeat=total)
        coeffs, freqs = Coefficients.get_spectrum(model, shift=True, trim=True, **kwargs)
        return model.params, coeffs, freqs

    @classmethod
    def _correlate(cls, mat: np.ndarray, method: str = "pearson") -> np.ndarray:
        """Correlate the columns of ``mat`` (samples x coefficients; extra coefficient axes
        are flattened in C order) -- ``coefficients.py:1301-1343``."""
        mat = np.asarray(mat)
        assert mat.ndim >= 2, "Input matrix must have at least 2 dimensions"
        fn = {"pearson": cls._pearson, "complex_pearson": cls._complex_pearson,
              "spearman": cls._spearman, "covariance": cls._covariance}.get(method)
        if fn is None:
            raise ValueError(
                f"Unknown correlation method: {method}. Must be 'pearson', "
                "'complex_pearson', 'spearman' or 'covariance'.")
        return fn(mat.reshape(mat.shape[0], -1))

    @classmethod
    def _covariance(cls, mat: np.ndarray, minp: Optional[int] = 1) -> np.ndarray:
        """Hermitian sample covariance ``sum conj(x_i - m_i)(x_j - m_j) / (nobs - 1)`` over
        pairwise-complete rows (``coefficients.py:1346-1395``)."""
        st = _PairStats(mat)
        return st.masked(st.sxy / np.where(st.nobs > 1, st.nobs - 1, np.nan), minp)

    @classmethod
    def _complex_pearson(cls, mat: np.ndarray, minp: Optional[int] = 1) -> np.ndarray:
        """Hermitian normalised covariance: ``|r_ij|`` = strength, ``angle(r_ij)`` = relative
        phase of column j against column i (``coefficients.py:1398-1453``)."""
        st = _PairStats(mat)
        denom = np.sqrt(st.ssx * st.ssy)
        with np.errstate(divide="ignore", invalid="ignore"):
            r = np.where(denom > 0, st.sxy / np.where(denom > 0, denom, 1.0), np.nan)
            mag = np.abs(r)
            r = np.where(mag > 1.0, r / mag, r)
        return st.masked(r, minp)

    @classmethod
    def _pearson(cls, mat: np.ndarray, minp: Optional[int] = 1) -> np.ndarray:
        """``cov_ij / sqrt(cov_ii cov_jj)``, clipped to [-1, 1]; complex input is stacked
        as real + imaginary samples (``coefficients.py:1456-1497``)."""
        cov = cls._covariance(_stack_complex(mat), minp=minp)
        sd = np.sqrt(np.real(np.diagonal(cov)))
        denom = sd[:, None] * sd[None, :]
        with np.errstate(divide="ignore", invalid="ignore"):
            r = np.where(denom > 0, np.real(cov) / np.where(denom > 0, denom, 1.0), np.nan)
        return np.clip(r, -1.0, 1.0)

    @classmethod
    def _spearman(cls, mat: np.ndarray, minp: Optional[int] = 1) -> np.ndarray:
        """Pearson correlation of the column-wise average ranks (non-finite entries are
        left out of the ranking) -- ``coefficients.py:1500-1579``."""
        mat = _stack_complex(mat)
        N, K = mat.shape
        if N < minp:
            return np.full((K, K), np.nan)
        ranks = np.full((N, K), np.nan)
        for j in range(K):
            ok = np.isfinite(mat[:, j])
            if ok.any():
                ranks[ok, j] = rankdata(mat[ok, j], method="average")
        st = _PairStats(ranks)
        denom = np.sqrt(st.ssx * st.ssy)
        with np.errstate(divide="ignore", invalid="ignore"):
            r = np.where(denom > 0, np.real(st.sxy) / np.where(denom > 0, denom, 1.0), np.nan)
        return st.masked(np.clip(r, -1.0, 1.0), minp)

    @classmethod
    def _weighting_linear(cls, fourier_fingerprint: np.ndarray) -> np.ndarray:
        """Tent weights ``u_i + u_j``, ``u_k = (c - |k - c|) / (2c)``, c = centre (zero
        frequency) of an odd-sized fingerprint (``coefficients.py:1582-1614``)."""
        fp = np.asarray(fourier_fingerprint)
        assert fp.shape[0] % 2 != 0 and fp.shape[1] % 2 != 0, (
            "Correlation matrix must have odd dimensions. "
            "Hint: use `trim` argument when calling `get_spectrum`.")
        assert fp.shape[0] == fp.shape[1], "Correlation matrix must be square."
        c = fp.shape[0] // 2
        u = (c - np.abs(np.arange(fp.shape[0]) - c)) / (2 * c)
        return fp * (u[:, None] + u[None, :])

    @classmethod
    def _weighting_mean(cls, fourier_fingerprint: np.ndarray, coeffs: np.ndarray) -> np.ndarray:
        """Weights ``|mean_i| |mean_j|`` of the coefficient means over the samples, in the
        coefficient order ``_correlate`` uses (``coefficients.py:1617-1649``)."""
        fp, coeffs = np.asarray(fourier_fingerprint), np.asarray(coeffs)
        assert fp.shape[0] == fp.shape[1], "Correlation matrix must be square."
        assert coeffs.ndim >= 2, (
            "Coefficient matrix must contain coefficient axes and a sample axis.")
        m = np.abs(np.mean(coeffs, axis=-1)).T.reshape(-1)
        assert fp.shape[0] == m.shape[0], (
            "Correlation matrix size must match the number of Fourier coefficients.")
        return fp * m[:, None] * m[None, :]


class Datasets:
    """Synthetic training targets: a random real Fourier series on the model's own spectrum."""

    @classmethod
    def generate_fourier_series(cls, random_key, model: Model, coefficients_min: float = 0.0,
                                coefficients_max: float = 1.0, zero_centered: bool = False):
        """``[x, f(x), c]``: grid points ``(*degree, D)``, series values ``(*degree)`` and the
        (conjugate-symmetric, fftshift-ordered) coefficients ``(*degree)`` with
        ``fftshift(fftn(f)) == c`` (``coefficients.py:1654-1757``).  The flattened spectrum
        is ``[conj(c_M..c_1), c_0, c_1..c_M]`` with ``c_0`` real (or 0), drawn by
        :meth:`uniform_circle`."""
        D = model.n_input_feat
        grid = np.stack(np.meshgrid(*[np.arange(0, 2 * np.pi, 2 * np.pi / d)
                                      for d in model.degree])).T.reshape(-1, D)
        freqs = np.stack(np.meshgrid(*[np.asarray(f) for f in model.frequencies])).T.reshape(-1, D)
        half = cls.uniform_circle(random_key, size=math.prod(model.degree) // 2 + 1,
                                  low=coefficients_min, high=coefficients_max)
        half[0] = 0.0 if zero_centered else half[0].real
        coeffs = np.concatenate([np.conj(half[1:][::-1]), half])
        values = np.real((np.exp(1j * (grid @ freqs.T)) * coeffs).sum(axis=1) / coeffs.size)
        shape = tuple(model.degree)
        return [grid.reshape(*shape, -1), values.reshape(shape), coeffs.reshape(shape)]

    @classmethod
    def uniform_circle(cls, random_key, size, low: float = 0.0, high: float = 1.0) -> np.ndarray:
        """Complex numbers uniform on the disc / annulus: sqrt(U[low, high]) e^{2 pi i U}
        (``coefficients.py:1759-1788``)."""
        from .utils import as_key

        size = (size,) if isinstance(size, (int, np.integer)) else tuple(np.asarray(size).tolist())
        k_r, k_phi = as_key(random_key).split()
        r = np.sqrt(k_r.generator().uniform(low, high, size=size))
        return r * np.exp(2j * np.pi * k_phi.generator().uniform(size=size))

