"""Numerical Fourier coefficients of a model (``Coefficients``, FFT path).

API mirror of ``qml_essentials/coefficients.py:23-237``: ``get_spectrum``,
``_fourier_transform``, ``get_psd``, ``evaluate_Fourier_series``.  The cost is the
batched model evaluation on the input grid (``:130``) -- that is the HIP engine's
job, sharded across ranks by :class:`script.Script`; the FFT of the few thousand
expectation values is done with NumPy on the host.  ``FourierTree`` (analytic),
``FCC`` and ``Datasets`` are out of scope.
"""
from __future__ import annotations

import math
from typing import Any, List, Optional, Tuple, Union

import numpy as np

from .model import Model


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
        if not np.isclose(np.sum(coeffs).imag, 0.0, atol=1.0e-6):
            raise ValueError(
                f"Spectrum is not real. Imaginary part of coefficients is: {np.sum(coeffs).imag}"
            )
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
        n_freqs = np.array([mfs * model.degree[i] for i in range(F)])
        axes = [np.arange(0, 2 * mts * np.pi, 2 * np.pi / n_freqs[i]) for i in range(F)]
        grid = np.array(np.meshgrid(*axes)).T.reshape(-1, F)
        outputs = np.asarray(model(inputs=grid.astype(np.float32), **kwargs))
        outputs = outputs.reshape(*[a.shape[0] for a in axes], -1).squeeze()
        coeffs = np.fft.fftn(outputs, axes=list(range(F)))
        freqs = [np.fft.fftfreq(int(mts * n_freqs[i]), 1 / n_freqs[i]) for i in range(F)]
        return coeffs / math.prod(outputs.shape[0:F]), freqs

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
