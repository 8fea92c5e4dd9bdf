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

COMPLEX_BYTES, REAL_BYTES = 8, 4  # complex64 / float32 (doubled in complex128 mode)
IN_FLIGHT_TARGET_BYTES = 32768 << 20  # states per engine launch (mirrors libqmle_sv's default)


def available_memory_bytes() -> int:
    """Free HBM on the current device (monkeypatched in tests, like the reference)."""
    import torch

    free, _total = torch.cuda.mem_get_info()
    return int(free)


F64_IN_FLIGHT_TARGET_BYTES = 4 << 30  # complex128 engine: states per round of launches (qmle_f64.hip)


def output_bytes(type: str, batch_size: int, n_qubits: int, n_obs: int, x64: bool = False) -> int:
    dim = 2**n_qubits
    cb, rb = (2 * COMPLEX_BYTES, 2 * REAL_BYTES) if x64 else (COMPLEX_BYTES, REAL_BYTES)
    if type == "density":
        return batch_size * dim * dim * cb
    if type == "expval":
        return batch_size * max(n_obs, 1) * rb
    if type == "probs":
        return batch_size * dim * rb
    return batch_size * dim * cb


def estimate_peak_bytes(n_qubits: int, batch_size: int, type: str, use_density: bool = False,
                        n_obs: int = 0, n_ops: int = 1, x64: bool = False,
                        general_obs: bool = False) -> int:
    """Device bytes needed to run ``batch_size`` samples in one engine call.  ``x64``: the
    complex128 engine (``qmle_run_batch_f64``: 16-byte amplitudes, float64 matrix rows, 4 GiB of
    states in flight above 13 qubits); ``general_obs``: expectation values of observables the
    engine does not measure itself -- the states are kept and contracted afterwards (one more
    copy for the contraction's temporary)."""
    cb = 2 * COMPLEX_BYTES if x64 else COMPLEX_BYTES
    state = (2**n_qubits) * cb
    out = output_bytes(type, batch_size, n_qubits, n_obs, x64)
    if use_density and type != "density":
        # noisy tape: every sample's vec(rho) (4^n amplitudes) is materialised, plus one
        # scratch copy for general observables
        return int(1.1 * (out + 2 * batch_size * state * (2**n_qubits)
                          + batch_size * max(n_ops, 1) * (128 if x64 else 64))) + (1 << 20)
    lds_limit = 13 if x64 else 14
    if general_obs and type == "expval":
        in_flight = 2 * batch_size * state  # psi of every sample + the contraction's temporary
    elif x64 and n_qubits > lds_limit:
        # qmle_workspace_bytes_f64 (qmle_f64.hip): above 13 qubits the complex128 engine works
        # in its own round of state buffers for EVERY measurement type, "state" included
        in_flight = min(batch_size, max(1, F64_IN_FLIGHT_TARGET_BYTES // state), 65535) * state
    elif x64 and type == "density":
        in_flight = batch_size * state  # (n <= 13: psi of every sample before the outer product)
    elif type == "state":
        in_flight = 0  # computed in place in the output
    elif n_qubits <= lds_limit and type in ("probs", "expval"):
        in_flight = 0  # whole state lives in LDS, never in HBM
    else:
        target = F64_IN_FLIGHT_TARGET_BYTES if x64 else IN_FLIGHT_TARGET_BYTES
        per_chunk = max(1, target // state)
        in_flight = min(batch_size, per_chunk) * state
        if not x64 and batch_size > per_chunk:
            in_flight *= 2  # (round 5: a batch of several chunks alternates between two sets of state buffers)
    mats = batch_size * max(n_ops, 1) * (64 if x64 else 32)
    return int(1.1 * (out + in_flight + mats)) + (1 << 20)


def compute_chunk_size(n_qubits: int, batch_size: int, type: str, use_density: bool = False,
                       n_obs: int = 0, memory_fraction: float = 0.8, n_ops: int = 1,
                       x64: bool = False, general_obs: bool = False) -> int:
    """Largest batch chunk whose engine call fits in ``memory_fraction`` of free HBM."""
    avail = int(available_memory_bytes() * memory_fraction)
    if estimate_peak_bytes(n_qubits, batch_size, type, use_density, n_obs, n_ops, x64, general_obs) <= avail:
        return batch_size
    per_elem = estimate_peak_bytes(n_qubits, 1, type, use_density, n_obs, n_ops, x64, general_obs)
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
