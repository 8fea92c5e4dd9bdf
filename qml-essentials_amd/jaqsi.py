"""Entry-point namespace (``import qml_essentials_amd.jaqsi as js``).

Mirror of the hot-path part of ``qml_essentials/jaqsi.py``: ``Script`` re-export,
``partial_trace`` (:60-103), ``marginalize_probs`` (:106-146) and
``build_parity_observable`` (:149-167).  These helpers act on small host arrays
(reduced density matrices, marginals); the 2^n-sized work stays in ``libqmle_sv``
(``_native.marginal_probs`` / ``meyer_wallach``).
"""
from __future__ import annotations

from functools import reduce
from typing import List, Sequence

import numpy as np

from .operations import Hermitian, PauliZ  # noqa: F401
from .script import Script  # noqa: F401


def _host(x) -> np.ndarray:
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def partial_trace(rho, n_qubits: int, keep: Sequence[int]) -> np.ndarray:
    """Reduced density matrix on ``keep`` (kept wires stay in ascending order);
    accepts ``(2^n, 2^n)`` or ``(B, 2^n, 2^n)``."""
    rho = _host(rho)
    dim = 2**n_qubits
    single = rho.shape == (dim, dim)
    r = rho.reshape((-1,) + (2,) * (2 * n_qubits))
    gone = sorted(set(range(n_qubits)) - set(int(k) for k in keep))
    live = n_qubits
    for q in reversed(gone):
        r = np.trace(r, axis1=1 + q, axis2=1 + q + live)
        live -= 1
    d = 2**live
    r = r.reshape(-1, d, d)
    return r[0] if single else r


def marginalize_probs(probs, n_qubits: int, keep: Sequence[int]) -> np.ndarray:
    """Sum a probability vector over every wire not in ``keep`` -> ``(B, 2^k)``."""
    p = _host(probs).reshape((-1,) + (2,) * n_qubits)
    drop = tuple(1 + q for q in range(n_qubits) if q not in set(int(k) for k in keep))
    return p.sum(axis=drop).reshape(p.shape[0], -1)


def build_parity_observable(qubit_group: List[int]) -> Hermitian:
    """Z (x) Z (x) ... on ``qubit_group``; tagged so the engine takes the parity kernel."""
    z = np.diag([1.0, -1.0]).astype(np.complex128)
    obs = Hermitian(matrix=reduce(np.kron, [z] * len(qubit_group)), wires=qubit_group,
                    record=False)
    obs._pauli_label = "Z" * len(qubit_group)
    return obs
