"""Simulation + measurement seam -- now a thin driver of ``libqmle_sv`` (HIP).

API mirror of ``qml_essentials/simulation.py``: ``infer_n_qubits`` (:25-39),
``uses_density`` (:42-57), ``simulate_and_measure`` (:131-201).  A recorded tape is
lowered to the C-ABI op list + a ``[B, n_slots]`` angle table; compiled plans are
cached on the *structure* of the tape (gate names, wires, constants), which fixes
the stale-cache hazard the reference documents at ``script.py:312-314``.

There is no CPU path: without the HIP library / a GPU every call raises.
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict
from typing import List, Optional, Sequence

import numpy as np

from . import _native as N
from .operations import Barrier, Operation, z_parity_mask

MEAS_TYPES = ("expval", "probs", "state", "density")
_PLAN_CACHE: "OrderedDict[tuple, N.Plan]" = OrderedDict()
_PLAN_CACHE_MAX = 256
PLAN_FLAGS = 0  # module-level override for experiments (e.g. N.PLAN_NO_FUSION)


def infer_n_qubits(ops: Sequence[Operation], obs: Sequence[Operation]) -> int:
    wires = set()
    for o in list(ops) + list(obs):
        wires.update(o.wires)
    return max(wires) + 1 if wires else 1


def uses_density(tape: Sequence[Operation], type: str) -> bool:
    """Noise channels are not supported here, so only the explicit request counts."""
    return type == "density"


class LoweredTape:
    """Engine view of a tape: op list, per-slot values, const blob, structure key."""

    def __init__(self, tape: Sequence[Operation], n_qubits: int):
        self.ops, self.values, blobs = [], [], []
        h = hashlib.blake2b(digest_size=16)
        h.update(str(n_qubits).encode())
        const_len = 0
        for op in tape:
            low = op.lower(n_qubits)
            if low is None:  # Barrier
                continue
            name, wires, params, blob = low
            slots = []
            for p in params:
                slots.append(len(self.values))
                self.values.append(p)
            off = -1
            if blob is not None:
                off = const_len
                blobs.append(np.asarray(blob, dtype=np.float32))
                const_len += blobs[-1].size
                h.update(blobs[-1].tobytes())
            self.ops.append((name, list(wires), slots, off))
            h.update(f"{name}{wires}{len(slots)}{off};".encode())
        self.consts = np.concatenate(blobs) if blobs else np.zeros(0, dtype=np.float32)
        self.n_slots = len(self.values)
        self.n_qubits = n_qubits
        self.key = h.hexdigest()

    def angle_table(self, batch: int) -> np.ndarray:
        table = np.empty((batch, max(1, self.n_slots)), dtype=np.float32)
        if self.n_slots == 0:
            table[:] = 0
        for j, v in enumerate(self.values):
            col = np.asarray(v, dtype=np.float64)
            if col.ndim and col.shape[0] != batch:
                raise ValueError(f"parameter column has batch {col.shape[0]}, expected {batch}")
            table[:, j] = col
        return table[:, : self.n_slots] if self.n_slots else table[:, :0]


def get_plan(low: LoweredTape, flags: Optional[int] = None) -> N.Plan:
    flags = PLAN_FLAGS if flags is None else flags
    key = (low.key, flags)
    plan = _PLAN_CACHE.get(key)
    if plan is None:
        plan = N.Plan(low.ops, low.n_qubits, low.n_slots, low.consts, flags)
        _PLAN_CACHE[key] = plan
        if len(_PLAN_CACHE) > _PLAN_CACHE_MAX:
            _PLAN_CACHE.popitem(last=False)
    else:
        _PLAN_CACHE.move_to_end(key)
    return plan


def clear_plan_cache() -> None:
    _PLAN_CACHE.clear()


def _tape_batch(tape: Sequence[Operation]) -> int:
    b = 1
    for op in tape:
        for p in op.parameters:
            if isinstance(p, np.ndarray) and p.ndim > 0:
                if b not in (1, p.shape[0]):
                    raise ValueError(f"inconsistent batch sizes on the tape: {b} vs {p.shape[0]}")
                b = p.shape[0]
    return b


def _general_expval(states, n_qubits: int, obs: Sequence[Operation]):
    """<psi|O|psi> for arbitrary observables, matrix-free: parities natively, anything
    else as Re<psi|O psi> with O applied by the gate kernels (``simulation.py:263-269``)."""
    torch = N.require_gpu()
    B = states.shape[0]
    out = torch.empty((B, len(obs)), dtype=torch.float32, device=states.device)
    parity_idx, parity_groups = [], []
    for k, ob in enumerate(obs):
        mask = z_parity_mask(ob)
        if mask is not None:
            parity_idx.append(k)
            parity_groups.append(mask)
    if parity_groups:
        out[:, parity_idx] = N.expval_parity(states, parity_groups)
    for k, ob in enumerate(obs):
        if k in parity_idx:
            continue
        low = LoweredTape([ob], n_qubits)
        plan = get_plan(low, N.PLAN_NO_FUSION)
        scratch = states.clone()
        ang = torch.from_numpy(low.angle_table(B)).to(states.device) if low.n_slots else None
        for b0 in range(0, B, 65535):
            sl = slice(b0, min(B, b0 + 65535))
            N.apply_inplace(plan, None if ang is None else ang[sl], scratch[sl])
        out[:, k] = N.overlap(states, scratch).real
    return out


def simulate_and_measure(tape: Sequence[Operation], n_qubits: int, type: str,
                         obs: Sequence[Operation] = (), use_density: bool = False,
                         shots: Optional[int] = None, key=None, batch: Optional[int] = None,
                         as_tensor: bool = False):
    """Run the tape from |0..0> and measure.  Returns ``(B, ...)`` (numpy unless
    ``as_tensor``); the caller strips the batch axis for un-batched execution."""
    if type not in MEAS_TYPES:
        raise ValueError(f"Unknown measurement type: {type!r}")
    if shots is not None:
        raise NotImplementedError("shot sampling is a later row (SURVEY.md 8-f rank 4)")
    torch = N.require_gpu()
    low = LoweredTape(tape, n_qubits)
    B = int(batch) if batch is not None else _tape_batch(tape)
    plan = get_plan(low)
    angles = torch.from_numpy(low.angle_table(B)).cuda()
    if type == "expval":
        obs = list(obs)
        masks = [z_parity_mask(o) for o in obs]
        if obs and all(m is not None and len(m) == 1 for m in masks):
            res = plan.run(angles, "expval", [m[0] for m in masks])
        elif not obs:
            res = torch.empty((B, 0), dtype=torch.float32, device=angles.device)
        else:
            res = _general_expval(plan.run(angles, "state"), n_qubits, obs)
    else:
        res = plan.run(angles, type)
    if as_tensor:
        return res
    return res.cpu().numpy()


def run_expval_table(plan: N.Plan, table: np.ndarray, obs: Sequence[Operation], n_qubits: int,
                     max_rows: int = 1 << 16) -> np.ndarray:
    """Expectation values for every row of an explicit angle table (parameter-shift batches)."""
    torch = N.require_gpu()
    masks = [z_parity_mask(o) for o in obs]
    single_z = bool(obs) and all(m is not None and len(m) == 1 for m in masks)
    out = []
    from . import memory
    chunk = memory.compute_chunk_size(n_qubits, min(max_rows, table.shape[0]),
                                      "expval" if single_z else "state", False, len(obs),
                                      n_ops=plan.n_ops)
    for r0 in range(0, table.shape[0], chunk):
        ang = torch.from_numpy(np.ascontiguousarray(table[r0:r0 + chunk])).cuda()
        if single_z:
            res = plan.run(ang, "expval", [m[0] for m in masks])
        else:
            res = _general_expval(plan.run(ang, "state"), n_qubits, list(obs))
        out.append(res.cpu().numpy())
    return np.concatenate(out, axis=0)
