"""HBM-aware batch chunking (the reference's RAM model, re-targeted at 288 GB HBM3E).

Counterpart of ``qml_essentials/memory.py``: ``estimate_peak_bytes`` (:54-150),
``compute_chunk_size`` (:186-261), ``execute_chunked`` (:264-345).  The engine keeps
ONE state per in-flight sample (gates are applied in place, nothing like XLA's
``n_ops`` live buffers), so the model is: output array + in-flight states +
per-sample gate matrices.
"""
from __future__ import annotations

import logging
from typing import Callable, Tuple

import numpy as np

log = logging.getLogger(__name__)

COMPLEX_BYTES, REAL_BYTES = 8, 4  # complex64 / float32
IN_FLIGHT_TARGET_BYTES = 32768 << 20  # states per engine launch (mirrors libqmle_sv's default)


def available_memory_bytes() -> int:
    """Free HBM on the current device (monkeypatched in tests, like the reference)."""
    import torch

    free, _total = torch.cuda.mem_get_info()
    return int(free)


def output_bytes(type: str, batch_size: int, n_qubits: int, n_obs: int) -> int:
    dim = 2**n_qubits
    if type == "density":
        return batch_size * dim * dim * COMPLEX_BYTES
    if type == "expval":
        return batch_size * max(n_obs, 1) * REAL_BYTES
    if type == "probs":
        return batch_size * dim * REAL_BYTES
    return batch_size * dim * COMPLEX_BYTES


def estimate_peak_bytes(n_qubits: int, batch_size: int, type: str, use_density: bool = False,
                        n_obs: int = 0, n_ops: int = 1) -> int:
    """Device bytes needed to run ``batch_size`` samples in one engine call."""
    state = (2**n_qubits) * COMPLEX_BYTES
    out = output_bytes(type, batch_size, n_qubits, n_obs)
    if use_density and type != "density":
        # noisy tape: every sample's vec(rho) (4^n amplitudes) is materialised, plus one
        # scratch copy for general observables
        return int(1.1 * (out + 2 * batch_size * state * (2**n_qubits)
                          + batch_size * max(n_ops, 1) * 64)) + (1 << 20)
    if type == "state":
        in_flight = 0  # computed in place in the output
    elif n_qubits <= 14 and type in ("probs", "expval"):
        in_flight = 0  # whole state lives in LDS, never in HBM
    else:
        in_flight = min(batch_size, max(1, IN_FLIGHT_TARGET_BYTES // state)) * state
    mats = batch_size * max(n_ops, 1) * 32
    return int(1.1 * (out + in_flight + mats)) + (1 << 20)


def compute_chunk_size(n_qubits: int, batch_size: int, type: str, use_density: bool = False,
                       n_obs: int = 0, memory_fraction: float = 0.8, n_ops: int = 1) -> int:
    """Largest batch chunk whose engine call fits in ``memory_fraction`` of free HBM."""
    avail = int(available_memory_bytes() * memory_fraction)
    if estimate_peak_bytes(n_qubits, batch_size, type, use_density, n_obs, n_ops) <= avail:
        return batch_size
    per_elem = estimate_peak_bytes(n_qubits, 1, type, use_density, n_obs, n_ops)
    chunk = max(1, min(batch_size, avail // max(per_elem, 1)))
    if chunk == 1 and per_elem > avail:
        log.warning("A single batch element needs ~%.2f GB but only ~%.2f GB of HBM is free.",
                    per_elem / 2**30, avail / 2**30)
    log.info("Batch of %d does not fit in HBM; using chunks of %d.", batch_size, chunk)
    return chunk


def execute_chunked(run: Callable[[int, int], np.ndarray], batch_size: int,
                    chunk_size: int) -> np.ndarray:
    """``run(start, end)`` per chunk, results gathered into one host array
    (results that do not fit in HBM at once are streamed to host memory)."""
    out = None
    for start in range(0, batch_size, chunk_size):
        end = min(batch_size, start + chunk_size)
        part = run(start, end)
        if hasattr(part, "cpu"):
            part = part.cpu().numpy()
        if out is None:
            out = np.empty((batch_size,) + part.shape[1:], dtype=part.dtype)
        out[start:end] = part
    return out
