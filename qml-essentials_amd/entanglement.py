"""Entangling capability: Meyer-Wallach measure on statevectors.

API mirror of ``Entanglement.meyer_wallach`` / ``_compute_meyer_wallach_meas``
(``qml_essentials/entanglement.py:17-103``).  The reference traces out one qubit
at a time from full density matrices (``jaqsi.partial_trace``) and squares
2^(n-1)-dimensional matrices.  For pure states ``Tr rho_{not j}^2 = Tr rho_j^2``
(Schmidt decomposition) and ``rho_j`` is 2x2, so the HIP kernel only needs, per
wire, the two populations and one cross term: ``a^2 + d^2 + 2|c|^2``.

The multi-register measures (Bell measurement, relative entropy, entanglement of
formation, concentratable entanglement; ``entanglement.py:106-712``) are later rows
(SURVEY.md 8-f rank 1).
"""
from __future__ import annotations

import logging
from typing import Any, Optional

import numpy as np

from . import _native as N
from . import distributed
from . import jaqsi as js
from .model import Model

log = logging.getLogger(__name__)


class Entanglement:
    @classmethod
    def meyer_wallach(cls, model: Model, n_samples: Optional[int], random_key=None,
                      scale: bool = False, **kwargs: Any) -> float:
        """Mean Meyer-Wallach Q over ``n_samples`` random parameter sets (or over the
        model's current parameters when ``n_samples`` is None / <= 0)."""
        if "noise_params" in kwargs and kwargs["noise_params"]:
            log.warning("Meyer-Wallach measure not suitable for noisy circuits. "
                        "Consider 'concentratable entanglement' instead.")
        if scale:
            n_samples = (2**model.n_qubits) * n_samples
        if n_samples is not None and n_samples > 0:
            random_key = model.initialize_params(random_key, repeat=int(n_samples))
        kwargs.setdefault("inputs", None)
        kwargs.pop("execution_type", None)
        params = np.asarray(model.params)
        total = params.shape[0]
        lo, hi = 0, total
        sharded = distributed.enabled() and total >= distributed.world()[1]
        if sharded:
            lo, hi = distributed.shard_bounds(total)
        with distributed.local_only():
            states = model._forward(params=params[lo:hi], execution_type="state",
                                    as_tensor=True, **kwargs)
        model.params = params
        ent = cls._compute_meyer_wallach_meas(states, model.n_qubits)
        if sharded:
            b_i = model.batch_shape[0]
            ent = distributed.all_gather_rows(ent.reshape(b_i, -1).transpose(0, 1).contiguous(),
                                              total)
        log.debug("Variance of measure: %s", float(ent.var()) if ent.numel() > 1 else 0.0)
        return float(ent.double().mean())

    @classmethod
    def _compute_meyer_wallach_meas(cls, states, n_qubits: int):
        """Q per sample.  ``states``: ``(B, 2^n)`` device tensor / array of statevectors,
        or ``(B, 2^n, 2^n)`` density matrices (reference signature; host formula)."""
        if getattr(states, "ndim", 0) == 3:
            rhos = np.asarray(js._host(states))
            out = np.zeros(rhos.shape[0])
            for j in range(n_qubits):
                keep = [q for q in range(n_qubits) if q != j]
                red = js.partial_trace(rhos, n_qubits, keep)
                out += np.trace((red @ red).real, axis1=-2, axis2=-1)
            return 2 * (1 - out / n_qubits)
        torch = N.require_gpu()
        if not hasattr(states, "is_cuda"):
            states = torch.from_numpy(np.ascontiguousarray(states, dtype=np.complex64)).cuda()
        vals = [N.meyer_wallach(states[b0:b0 + 65535]) for b0 in range(0, states.shape[0], 65535)]
        return torch.cat(vals)
