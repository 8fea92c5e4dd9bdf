"""Expressibility (Sim et al., arXiv:1905.10876) on statevectors.

API mirror of ``qml_essentials/expressibility.py``.  The reference samples
``2 * n_samples`` parameter sets, builds full density matrices and evaluates
``F = (Tr sqrt(sqrt(rho) sigma sqrt(rho)))^2`` with a Python loop of
``scipy.linalg.sqrtm`` (``:14-66``).  For the pure states a noise-free circuit
produces this equals ``|<psi_i|psi_{i+S}>|^2`` (``math.py:60-86``), which is what the
HIP path computes: states never leave the GPU, one fused overlap kernel per call,
pairs sharded across ranks with one all-gather of ``S`` floats.
"""
from __future__ import annotations

from functools import lru_cache
from typing import Any, Optional, Tuple

import numpy as np
from scipy.special import rel_entr

from . import _native as N
from . import distributed
from .model import Model


class Expressibility:
    @classmethod
    def _sample_state_fidelities(cls, model: Model, n_samples: int, random_key=None,
                                 kwargs: Any = None) -> np.ndarray:
        """Fidelities of ``n_samples`` random state pairs; sample ``i`` is paired with
        sample ``i + n_samples`` (``expressibility.py:49-52``)."""
        kwargs = dict(kwargs or {})
        n_samples = int(n_samples)
        kwargs.pop("execution_type", None)
        # Resolve the compiled call BEFORE drawing: once the sampler is launched nothing but the
        # engine's own launches stands between it and the circuits (the host work in between used
        # to leave the GPU idle for 30 us of a 0.2 ms call).
        prep = None
        if not any(v is not None for v in kwargs.values()) and not distributed.enabled():
            prep = model.prepared_state_call(2 * n_samples)
        model.initialize_params(random_key, repeat=n_samples * 2)
        torch = N.require_gpu()
        # The samples stay where they were drawn (the reference's are jax device arrays,
        # model.py:687-693 -> expressibility.py:38-46): a large draw is written by the GPU sampler
        # and handed to the engine as it lies; only small draws come from the host generator.
        params = model.device_params()
        if prep is not None and params is not None:
            cc, divs, mods, B = prep
            return N.pair_fidelity(cc.run([params], divs, mods, B, 0))
        if params is None:
            params = np.asarray(model.params)
        lo, hi, sharded = distributed.my_block(n_samples, params, kwargs.get("inputs"))
        if sharded:  # this rank's pairs {i, i + S}: both states of a pair stay on one rank
            local = params.reshape(2, n_samples, *params.shape[1:])[:, lo:hi]
            local = local.reshape(2 * (hi - lo), *params.shape[1:])
        else:
            local = params
        with distributed.local_only():
            states = model._forward(params=local, execution_type="state", as_tensor=True, **kwargs)
        model.params = params  # (all 2S sets, like the reference)
        b_i = model.batch_shape[0]
        if b_i == 1:
            fid = N.pair_fidelity(states)                       # (local pairs,)
            if sharded:
                fid = distributed.all_gather_rows(fid, n_samples)
            return fid
        states = states.reshape(b_i, 2 * (hi - lo), -1)
        fid = torch.stack([N.pair_fidelity(states[i]) for i in range(b_i)], dim=1)  # (pairs, B_I)
        if sharded:
            fid = distributed.all_gather_rows(fid, n_samples)
        return fid.transpose(0, 1)  # device tensor (B_I, S)

    @classmethod
    def state_fidelities(cls, n_samples: int, n_bins: int, model: Model, random_key=None,
                         scale: bool = False, **kwargs: Any) -> Tuple[np.ndarray, np.ndarray]:
        """Histogram of sampled fidelities over ``linspace(0, 1, n_bins + 1)``,
        normalised by ``n_samples`` (``expressibility.py:69-112``)."""
        if scale:
            n_samples = (2**model.n_qubits) * n_samples
            n_bins = model.n_qubits * n_bins
        fid = cls._sample_state_fidelities(model=model, n_samples=n_samples,
                                           random_key=random_key, kwargs=kwargs)
        y = cls._edges(int(n_bins))
        # bin on the GPU; the one device -> host copy of the call is the (rows x n_bins) counts
        if fid.dim() == 1:
            z = N.histogram(fid, n_bins, 0.0, 1.0).cpu().numpy() / n_samples
        else:
            torch = N.require_gpu()
            z = torch.stack([N.histogram(row, n_bins, 0.0, 1.0) for row in fid]).cpu().numpy() / n_samples
        return y, z

    @staticmethod
    @lru_cache(maxsize=64)
    def _edges_cached(n_bins: int):
        e = np.linspace(0, 1, n_bins + 1)
        e.setflags(write=False)
        return e

    @classmethod
    def _edges(cls, n_bins: int) -> np.ndarray:
        return cls._edges_cached(n_bins).copy()

    @classmethod
    def _haar_probability(cls, fidelity: float, n_qubits: int) -> float:
        N_ = 2**n_qubits
        return (N_ - 1) * (1 - fidelity) ** (N_ - 2)

    @staticmethod
    @lru_cache(maxsize=64)
    def _haar_bins(n_qubits: int, n_bins: int) -> Tuple[float, ...]:
        # closed form of the per-bin integral the reference evaluates with quad
        # (expressibility.py:133-152): int_v^u (N-1)(1-F)^(N-2) dF
        N_ = 2.0**n_qubits
        edges = np.linspace(0.0, 1.0, n_bins + 1)
        return tuple((1 - edges[:-1]) ** (N_ - 1) - (1 - edges[1:]) ** (N_ - 1))

    @classmethod
    def _sample_haar_integral(cls, n_qubits: int, n_bins: int) -> np.ndarray:
        return np.array(cls._haar_bins(int(n_qubits), int(n_bins)))

    @classmethod
    def haar_integral(cls, n_qubits: int, n_bins: int, cache: bool = True,
                      scale: bool = False) -> Tuple[np.ndarray, np.ndarray]:
        """Haar fidelity distribution binned like :meth:`state_fidelities`.  ``cache`` is
        accepted for API parity; values are memoised in memory, never on disk."""
        if scale:
            n_bins = n_qubits * n_bins
        return np.linspace(0, 1, n_bins), cls._sample_haar_integral(n_qubits, n_bins)

    @classmethod
    def kullback_leibler_divergence(cls, vqc_prob_dist: np.ndarray,
                                    haar_dist: np.ndarray) -> np.ndarray:
        p = np.asarray(vqc_prob_dist)
        haar_dist = np.asarray(haar_dist)
        if p.ndim > 1:
            assert all(haar_dist.shape == row.shape for row in p), (
                "All probabilities for inputs should have the same shape as Haar. "
                f"Got {haar_dist.shape} for Haar and {p.shape} for VQC"
            )
        else:
            p = p.reshape(1, -1)
        return np.array([np.sum(rel_entr(row, haar_dist)) for row in p])

    @classmethod
    def kl_divergence_to_haar(cls, model: Model, n_samples: int, n_bins: int, random_key=None,
                              scale: bool = False, **kwargs: Any) -> np.ndarray:
        _, z = cls.state_fidelities(model=model, random_key=random_key, n_samples=n_samples,
                                    n_bins=n_bins, scale=scale, **kwargs)
        _, haar = cls.haar_integral(model.n_qubits, n_bins=n_bins, scale=scale)
        return cls.kullback_leibler_divergence(z, haar)
