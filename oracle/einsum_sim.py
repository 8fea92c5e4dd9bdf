"""Literal NumPy restatement of the reference's statevector engine (oracle).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

Follows, line by line:
  * ``qml_essentials/operations.py:19-50``  (_einsum_subscript)
  * ``qml_essentials/simulation.py:65-104`` (simulate_pure)
  * ``qml_essentials/simulation.py:204-271`` (measure_state)
  * ``qml_essentials/simulation.py:131-201`` (simulate_and_measure, pure branch and
    the pure->density outer-product shortcut at :183-189)

A tape is a list of ``(name, wires, params)``; Barriers are skipped exactly as
``simulation.py:93-94`` does.  ``dtype`` selects complex64 (reference default,
x64 off, ``operations.py:12-16``) or complex128 (x64 on, used by most reference
tests).
"""
import string

import numpy as np

from . import gates as G


def einsum_subscript(n, k, target_axes):
    """operations.py:19-50: gate indices = (out_0..out_{k-1}, in_0..in_{k-1})."""
    letters = string.ascii_letters
    state_idx = list(letters[:n])
    contracted = [state_idx[ax] for ax in target_axes]
    new_out = [letters[n + i] for i in range(k)]
    gate_idx = new_out + contracted
    result_idx = list(state_idx)
    for i, ax in enumerate(target_axes):
        result_idx[ax] = new_out[i]
    return "".join(gate_idx) + "," + "".join(state_idx) + "->" + "".join(result_idx)


def infer_n_qubits(tape, obs_wires=()):
    """simulation.py:25-39."""
    wires = set()
    for _, w, _ in tape:
        wires.update(w)
    for w in obs_wires:
        wires.update(w if isinstance(w, (list, tuple)) else [w])
    return max(wires) + 1 if wires else 1


def simulate_pure(tape, n_qubits, dtype=np.complex64):
    """simulation.py:65-104: psi <- einsum(sub, gate_tensor, psi) per gate."""
    dim = 2**n_qubits
    compiled = []
    for name, wires, params in tape:
        if name == "Barrier":  # simulation.py:93-94
            continue
        k = len(wires)
        gt = G.matrix(name, params).astype(dtype).reshape((2,) * (2 * k))
        compiled.append((gt, einsum_subscript(n_qubits, k, tuple(wires))))
    state = np.zeros(dim, dtype=dtype)
    state[0] = 1.0  # simulation.py:100
    psi = state.reshape((2,) * n_qubits)
    for gt, sub in compiled:
        psi = np.einsum(sub, gt, psi)  # simulation.py:103
    return psi.reshape(dim)


def measure_state(state, n_qubits, type, obs=()):
    """simulation.py:204-271.

    ``obs`` is a list of ``(name, wires)`` with name in {"PauliZ","PauliX",
    "PauliY","Id","Matrix:<ndarray>"}; 1-qubit diagonal observables take the fast
    path (:241-261), anything else the lifted-matrix path (:263-269).
    """
    if type == "state":
        return state
    if type == "probs":
        return np.abs(state) ** 2
    if type == "expval":
        rdtype = np.float32 if state.dtype == np.complex64 else np.float64

        def _mat(ob):
            return ob[2] if len(ob) > 2 else G.matrix(ob[0])

        def _diag1(ob):
            m = _mat(ob)
            return len(ob[1]) == 1 and np.allclose(m - np.diag(np.diag(m)), 0)

        if all(_diag1(ob) for ob in obs):
            probs = np.abs(state) ** 2
            psi_t = probs.reshape((2,) * n_qubits)
            res = []
            for ob in obs:
                q = ob[1][0]
                d = np.real(np.diag(_mat(ob)))
                p_q = psi_t.sum(axis=tuple(i for i in range(n_qubits) if i != q))
                res.append(d[0] * p_q[0] + d[1] * p_q[1])
            return np.array(res, dtype=rdtype)
        res = []
        for ob in obs:  # general path, one observable at a time (same maths)
            m = _mat(ob).astype(state.dtype)
            k = len(ob[1])
            psi = state.reshape((2,) * n_qubits)
            o_psi = np.einsum(
                einsum_subscript(n_qubits, k, tuple(ob[1])),
                m.reshape((2,) * (2 * k)),
                psi,
            ).reshape(-1)
            res.append(np.real(np.vdot(state, o_psi)))
        return np.array(res, dtype=rdtype)
    raise ValueError(f"Unknown measurement type: {type!r}")  # simulation.py:271


def simulate_and_measure(tape, n_qubits, type, obs=(), dtype=np.complex64):
    """simulation.py:131-201, noise-free branches only."""
    state = simulate_pure(tape, n_qubits, dtype)
    if type == "density":  # simulation.py:183-189
        return np.outer(state, np.conj(state))
    return measure_state(state, n_qubits, type, obs)


def run_batch(tape_fn, batch_args, n_qubits, type, obs=(), dtype=np.complex64):
    """Sequential stand-in for ``jax.vmap(_single_execute)`` (script.py:302-315)."""
    return np.stack(
        [simulate_and_measure(tape_fn(*a), n_qubits, type, obs, dtype) for a in batch_args]
    )
