"""Adjoint differentiation: the gradient of a weighted sum of Z / Z-parity expectation values
with respect to EVERY gate angle in one backward sweep (``qmle_adjoint_gradient``).

What ``jax.grad`` through ``Script.execute`` gives the reference (``tests/test_jaqsi.py:131-141``,
``tests/test_model.py:1097-1145``, ``docs/training.md``).  Cost: one forward simulation, then
per gate one inverse-gate pass over [psi; lambda] and, per differentiable angle, one overlap
``Im <lambda| G |psi>`` -- O(gates + angles) passes instead of the 2 x angles full circuits of
the parameter-shift rule.

This module only prepares the REVERSED, DAGGERED tape and the generator table; the sweep itself
runs in the engine.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _native as N
from .simulation import LoweredTape, _Lowered, get_plan

# rotation gates exp(-i theta/2 P): generator Pauli word per wire, (control?, word)
_ROT = {"RX": (False, "X"), "RY": (False, "Y"), "RZ": (False, "Z"),
        "CRX": (True, "X"), "CRY": (True, "Y"), "CRZ": (True, "Z"),
        "RXX": (False, "XX"), "RYY": (False, "YY"), "RZZ": (False, "ZZ"), "RZX": (False, "ZX")}
_SELF_INVERSE = {"Id", "PauliX", "PauliY", "PauliZ", "H", "CX", "CY", "CZ", "SWAP", "CCX", "CSWAP"}


class AdjointUnsupported(NotImplementedError):
    pass


def _dagger_blob(blob: np.ndarray, dim: int) -> np.ndarray:
    keep = np.float64 if np.asarray(blob).dtype == np.float64 else np.float32  # (x64 mode: full precision)
    m = np.asarray(blob, dtype=np.float64).reshape(dim, dim, 2)
    m = (m[..., 0] + 1j * m[..., 1]).conj().T
    return np.stack([m.real, m.imag], axis=-1).astype(keep).reshape(-1)


def _term(out_slot=-1, x=0, z=0, proj=0, n_y=0, coef=0.0, marks_off=-1):
    return (int(out_slot), int(x), int(z), int(proj), int(n_y), float(coef), int(marks_off))


_BLOB_LEN = {"MAT1": 8, "MAT2": 32, "MAT4": 512}


def _op_blobs(low: LoweredTape, x64: bool = False) -> List[Optional[np.ndarray]]:
    out = []
    consts = low.consts64 if x64 else low.consts
    for name, _w, _s, off in low.ops:
        if off < 0:
            out.append(None)
        else:
            size = _BLOB_LEN.get(name, 1 << low.n_qubits)  # DIAG_ALL: one mark per amplitude
            out.append(consts[off:off + size])
    return out


def build_reverse(low: LoweredTape, blobs: List[Optional[np.ndarray]], want: Sequence[bool]):
    """From the forward lowered tape: the reversed, daggered primitive tape (as ``_Lowered`` ops),
    its per-slot values (negated forward columns) and one generator term per reverse op.

    ``blobs[k]``: constant blob of forward op k (or None); ``want[s]``: forward slot s needs a
    derivative.  Rot(phi, theta, omega) = RZ(omega) RY(theta) RZ(phi) is split into its three
    rotations so that every primitive has at most one angle."""
    prims = []  # forward order: (name, wires, fwd_slot | None, blob)
    for (name, wires, slots, _off), blob in zip(low.ops, blobs):
        if name == "Rot":
            prims += [("RZ", wires, slots[0], None), ("RY", wires, slots[1], None),
                      ("RZ", wires, slots[2], None)]
        elif len(slots) > 1:
            raise AdjointUnsupported(f"{name}: more than one angle per gate")
        else:
            prims.append((name, wires, slots[0] if slots else None, blob))
    rev_ops, rev_src, terms = [], [], []
    for name, wires, fslot, blob in reversed(prims):
        params, out_blob, term = [], None, _term()
        bit = lambda ws: sum(1 << int(w) for w in ws)  # noqa: E731
        if fslot is not None:
            params = [-np.asarray(low.values[fslot], dtype=np.float64)]
            rev_src.append(fslot)
        d = want[fslot] if fslot is not None else False
        if name in _ROT:
            ctrl, word = _ROT[name]
            tw = wires[1:] if ctrl else wires
            x = bit(w for w, p in zip(tw, word) if p in "XY")
            z = bit(w for w, p in zip(tw, word) if p in "ZY")
            if d:
                term = _term(fslot, x, z, bit(wires[:1]) if ctrl else 0, word.count("Y"), 1.0)
        elif name in ("CPhase", "ControlledPhaseShift"):
            if d:  # dU = i |11><11| U
                term = _term(fslot, 0, 0, bit(wires), 0, -2.0)
        elif name == "DIAG_ALL":
            out_blob = np.asarray(blob)
            if d:  # U = exp(-i M x): dU = -i M U
                term = _term(fslot, coef=2.0, marks_off=0)  # offset patched below
        elif name in _SELF_INVERSE:
            pass
        elif name == "S":
            name, out_blob = "MAT1", np.array([1, 0, 0, 0, 0, 0, 0, -1], dtype=np.float32)
        elif name in ("MAT1", "MAT2", "MAT4"):
            out_blob = _dagger_blob(blob, 2 ** len(wires))
        else:
            raise AdjointUnsupported(f"no adjoint rule for {name}")
        rev_ops.append(_Lowered((name, list(wires), params, out_blob)))
        terms.append(term)
    return rev_ops, terms, rev_src


def patch_marks(rev: LoweredTape, terms):
    """Golomb terms point at their marks inside the reverse tape's const blob."""
    return [t[:6] + (off,) if (name == "DIAG_ALL" and t[0] >= 0) else t
            for (name, _w, _s, off), t in zip(rev.ops, terms)]


# one streaming pass per gate (always possible) ...
REV_FLAGS = N.PLAN_NO_FUSION | N.PLAN_FORCE_GLOBAL | N.PLAN_NO_ABSORB
# ... or fused tile passes over [psi; lambda]: every gate stays its own operator (one generator
# per gate), tiles of 2^12 amplitudes so that both states fit in one workgroup's LDS
REV_FLAGS_FUSED = (N.PLAN_NO_MERGE | N.PLAN_FORCE_GLOBAL | N.PLAN_NO_ABSORB
                   | N.plan_flags(tile_bits=12, low_bits=4))


def run_sweep(fwd_plan, rev: LoweredTape, a_f, a_r, w, obs_groups, terms, n_grad_slots):
    """The backward sweep with fused tile passes where the engine supports the tape (1-qubit and
    controlled 1-qubit gates), else with one streaming pass per gate.  float64 tables: the
    complex128 sweep (one streaming launch per operator, like the complex128 forward engine)."""
    if w.dtype == N.require_gpu().float64:
        return N.adjoint_gradient(fwd_plan, get_plan(rev, REV_FLAGS), a_f, a_r, w, obs_groups, terms, n_grad_slots)
    try:
        return N.adjoint_gradient(fwd_plan, get_plan(rev, REV_FLAGS_FUSED), a_f, a_r, w, obs_groups,
                                  terms, n_grad_slots)
    except N.Unsupported:
        return N.adjoint_gradient(fwd_plan, get_plan(rev, REV_FLAGS), a_f, a_r, w, obs_groups,
                                  terms, n_grad_slots)


_REV_CACHE: "OrderedDict[tuple, tuple]" = OrderedDict()


def adjoint_slot_gradient(low: LoweredTape, n_qubits: int, batch: int, obs_groups,
                          weights: np.ndarray, want: Sequence[bool], x64: bool = False) -> np.ndarray:
    """d/d(angle slot) of sum_k weights[b, k] <Z..Z>_k for every forward slot -> [B, n_slots]
    (columns of slots that are not wanted stay zero; ``weights`` of shape [K * B, n_obs] -- K cotangents per
    sample, sample-minor -- give [K * B, n_slots] from one sweep over K * B states).  The reversed tape depends only on the
    STRUCTURE of the forward tape (its angles are the negated forward columns), so it is built
    once per structure.  ``x64``: float64 angle table, complex128 states, float64 gradients."""
    torch = N.require_gpu()
    ft, fn = (torch.float64, np.float64) if x64 else (torch.float32, np.float32)
    key = (low.key, tuple(bool(x) for x in want), bool(x64))
    hit = _REV_CACHE.get(key)
    if hit is None:
        rev_ops, terms, rev_src = build_reverse(low, _op_blobs(low, x64), want)
        rev = LoweredTape(rev_ops, n_qubits)
        hit = (rev, patch_marks(rev, terms),
               torch.tensor(rev_src if rev_src else [0], dtype=torch.int64, device="cuda"))
        _REV_CACHE[key] = hit
        if len(_REV_CACHE) > 64:
            _REV_CACHE.popitem(last=False)
    else:
        _REV_CACHE.move_to_end(key)
    rev, fixed, perm = hit
    a_f = torch.from_numpy(low.angle_table(batch, dtype=np.float64) if x64 else low.angle_table(batch)).cuda()
    rep = int(np.shape(weights)[0]) // max(1, batch)
    if rep > 1:  # several cotangents per sample (a Jacobian: one row of weights per output) in ONE sweep
        a_f = a_f.repeat(rep, 1)
        batch = batch * rep
    if rev.n_slots:
        a_r = (-a_f.index_select(1, perm)).contiguous()
    else:
        a_r = torch.zeros((batch, 1), dtype=ft, device=a_f.device)
    w = torch.from_numpy(np.ascontiguousarray(weights, dtype=fn)).cuda()
    return run_sweep(get_plan(low), rev, a_f, a_r, w, obs_groups, fixed,
                     max(1, low.n_slots)).cpu().numpy()[:, : low.n_slots]
