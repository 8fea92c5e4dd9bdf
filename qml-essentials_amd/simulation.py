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
from .operations import _CONJ_NEGATE, Barrier, KrausChannel, Operation, conj_lower, z_parity_mask

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
    """Density request, or a noise channel on the tape (``simulation.py:42-57``)."""
    return type == "density" or any(isinstance(o, KrausChannel) for o in tape)


MAX_DENSITY_QUBITS = 14  # vec(rho) is a 2n-qubit register; 2n <= 28 keeps it one tile plan


class _Lowered:
    """An already-lowered op (what ``LoweredTape`` consumes).  ``tangents``: per lowered parameter the
    affine terms of the source gate's parameter (``Operation.parameter_tangents``; a compiled call builds
    its angle map from them)."""

    __slots__ = ("_low", "_tan")

    def __init__(self, low, tangents=None):
        self._low = low
        self._tan = tangents

    def lower(self, n_qubits: int):
        return self._low

    @property
    def name(self) -> str:
        return self._low[0]

    @property
    def parameter_tangents(self) -> list:
        return self._tan if self._tan is not None else [[] for _ in self._low[2]]


def _conj_tangents(op_: Operation, name: str) -> list:
    """Tangent terms of ``conj_lower(op_)``'s parameters: the source gate's, negated where the
    parameter is (``operations._CONJ_NEGATE``); explicit-matrix conjugates have no parameters."""
    neg = _CONJ_NEGATE.get(name)
    if neg is None:
        return []
    out = []
    for j, t in enumerate(op_.parameter_tangents):
        if t is None or j not in neg:
            out.append(t)
        else:
            out.append([(lid, flat, -np.asarray(cf, dtype=np.float64)) for lid, flat, cf in t])
    return out


class _WideChannel:
    """A channel on 3 or 4 wires: too wide for one dense engine operator (its superoperator
    spans 6-8 wires), applied as the Kraus sum with one engine call per operator."""

    __slots__ = ("kraus", "wires")

    def __init__(self, kraus, wires):
        self.kraus, self.wires = kraus, list(wires)

    def kraus_plans(self, n_qubits: int):
        """One tiny tape per Kraus operator: K on the ket wires, conj(K) on the bra wires,
        both as 4-wire dense operators (3-wire K is padded with an untouched wire of the
        other half of the register)."""
        k = len(self.wires)
        ket = list(self.wires)
        bra = [w + n_qubits for w in self.wires]
        for K in self.kraus:
            K = np.asarray(K, dtype=np.complex128)
            if k == 3:
                K = np.kron(K, np.eye(2))
                wk, wb = ket + [bra[0]], bra + [ket[0]]
            else:
                wk, wb = ket, bra
            blob = lambda M: np.stack([M.real, M.imag], axis=-1).astype(np.float64).reshape(-1)  # noqa: E731
            yield [_Lowered(("MAT4", wk, [], blob(K))), _Lowered(("MAT4", wb, [], blob(np.conj(K))))]


def doubled_tape(tape: Sequence[Operation], n_qubits: int) -> list:
    """The tape acting on vec(rho), a pure "state" of a 2n-wire register whose first n wires
    carry the row (ket) index and whose last n wires carry the column (bra) index, so that
    ``vec(rho)[i * 2^n + j] = rho[i, j]``.

    * gate U -> U on the ket wires and conj(U) on the bra wires
      (``rho -> U rho U^+``, ``simulation.py:106-128`` + ``operations.py:485-512``);
    * channel {K} on wires w -> the dense superoperator ``sum_K K (x) conj(K)`` on
      ``[w.., n + w..]`` (``operations.py:1552-1578``): a 4x4 operator for 1-wire channels,
      a 16x16 one for 2-wire channels; 3- and 4-wire channels become a :class:`_WideChannel`.
    """
    out: list = []
    # Channels that follow one another on the same wires -- the five or six channels a noisy model puts
    # behind every gate (unitary.py:150-197) -- are ONE superoperator, the product of theirs: one dense
    # operator for the engine instead of one LDS round trip each.  `pending` holds the product per wire
    # tuple until an operation that shares a wire with it arrives (everything emitted in between acts on
    # other wires and commutes with it).
    pending: dict = {}

    def flush(touching=None):
        for w in [w for w in pending if touching is None or not touching.isdisjoint(w)]:
            S = pending.pop(w)
            blob = np.stack([S.real, S.imag], axis=-1).astype(np.float64).reshape(-1)
            wires = list(w) + [q + n_qubits for q in w]
            out.append(_Lowered(("MAT2" if len(w) == 1 else "MAT4", wires, [], blob)))

    for op_ in tape:
        if isinstance(op_, Barrier):
            continue
        if isinstance(op_, KrausChannel):
            k = len(op_.wires)
            if k > 4:
                raise NotImplementedError(
                    f"{op_.name}: channels on more than 4 wires are not available on the engine")
            if k > 2:
                flush()
                out.append(_WideChannel(op_.kraus_matrices(), op_.wires))
                continue
            w = tuple(op_.wires)
            S = op_.superoperator()
            if w in pending:
                pending[w] = S @ pending[w]
            else:
                flush(set(w))
                pending[w] = S
            continue
        low = op_.lower(n_qubits)
        if low is None:
            continue
        if low[0] == "DIAG_ALL":
            # rho_ij -> d_i conj(d_j) rho_ij (operations.py:944-961): one diagonal pass over
            # the doubled register with marks m_i - m_j
            flush()
            m = np.asarray(low[3], dtype=np.float64)
            out.append(_Lowered(("DIAG_ALL", [], low[2],
                                 (m[:, None] - m[None, :]).reshape(-1)), op_.parameter_tangents))
            continue
        flush(set(low[1]))
        out.append(_Lowered(low, op_.parameter_tangents))
        out.append(_Lowered(conj_lower(op_, n_qubits, n_qubits), _conj_tangents(op_, low[0])))
    flush()
    return out


def _apply_segment(seg, n2: int, B: int, rho_vec, x64: bool = False):
    """Run a run of lowered ops on the doubled register: from |0..0> when ``rho_vec`` is
    None, else in place on the resident ``rho_vec`` (complex128 engine when ``x64``)."""
    torch = N.require_gpu()
    low = LoweredTape(seg, n2)
    plan = get_plan(low)
    if x64:
        angles = torch.from_numpy(low.angle_table(B, dtype=np.float64)).cuda()
        if rho_vec is None:
            return plan.run64(angles, "state")
        for b0 in range(0, B, 65535):
            sl = slice(b0, min(B, b0 + 65535))
            N.apply_inplace64(plan, angles[sl] if low.n_slots else None, rho_vec[sl])
        return rho_vec
    angles = torch.from_numpy(low.angle_table(B)).cuda()
    if rho_vec is None:
        return plan.run(angles, "state")
    for b0 in range(0, B, 65535):
        sl = slice(b0, min(B, b0 + 65535))
        N.apply_inplace(plan, angles[sl] if low.n_slots else None, rho_vec[sl])
    return rho_vec


def _evolve_density(tape: Sequence[Operation], n_qubits: int, B: int, x64: bool = False):
    """vec(rho) [B, 4^n] after the whole (noisy) tape (complex128 when ``x64``)."""
    torch = N.require_gpu()
    n2 = 2 * n_qubits
    rho_vec, seg = None, []
    for item in doubled_tape(tape, n_qubits) + [None]:
        if isinstance(item, _Lowered):
            seg.append(item)
            continue
        if seg or rho_vec is None:
            rho_vec = _apply_segment(seg, n2, B, rho_vec, x64)
            seg = []
        if item is None:
            break
        acc = torch.zeros_like(rho_vec)
        for pair in item.kraus_plans(n_qubits):
            acc += _apply_segment(pair, n2, B, rho_vec.clone(), x64)
        rho_vec = acc
    return rho_vec


def _density_expval(rho_vec, n_qubits: int, obs: Sequence[Operation]):
    """Tr(O rho) per observable (``simulation.py:263-269`` for density matrices): Z natively,
    anything else by applying O to the ket wires and summing the diagonal."""
    torch = N.require_gpu()
    B = rho_vec.shape[0]
    masks = [z_parity_mask(o) for o in obs]
    if obs and all(m is not None and len(m) == 1 for m in masks):
        return N.density_expval_z(rho_vec, n_qubits, [m[0] for m in masks])
    out = torch.empty((B, len(obs)), dtype=torch.float32, device=rho_vec.device)
    diag = None
    for k, (ob, m) in enumerate(zip(obs, masks)):
        if m is not None:
            if diag is None:
                diag = N.density_probs(rho_vec, n_qubits).double()
            idx = torch.arange(1 << n_qubits, device=rho_vec.device)
            par = torch.zeros_like(idx)
            for w in m:
                par ^= (idx >> (n_qubits - 1 - w)) & 1
            out[:, k] = (diag * (1.0 - 2.0 * par.double())).sum(dim=1).float()
            continue
        low = LoweredTape([ob], 2 * n_qubits)
        plan = get_plan(low, N.PLAN_NO_FUSION)
        scratch = rho_vec.clone()
        N.apply_inplace(plan, None, scratch)
        out[:, k] = N.density_probs(scratch, n_qubits).double().sum(dim=1).float()
    return out


def _density_measure_x64(rho_vec, n_qubits: int, type: str, obs):
    """Measurements of a complex128 density matrix (x64 mode): float64 throughout; the diagonal
    and its signed sums are a few thousand numbers -- torch, not a kernel.  Non-diagonal
    observables are applied to the ket wires by the complex128 engine, then traced."""
    torch = N.require_gpu()
    B, D = rho_vec.shape[0], 1 << n_qubits
    rho = rho_vec.view(B, D, D)
    if type == "density":
        return rho
    diag = torch.diagonal(rho, dim1=1, dim2=2).real.contiguous()
    if type == "probs":
        return diag
    obs = list(obs)
    if not obs:
        return torch.empty((B, 0), dtype=torch.float64, device=rho_vec.device)
    out = torch.empty((B, len(obs)), dtype=torch.float64, device=rho_vec.device)
    idx = torch.arange(D, device=rho_vec.device)
    for k, ob in enumerate(obs):
        m = z_parity_mask(ob)
        if m is not None:
            par = torch.zeros_like(idx)
            for w in m:
                par ^= (idx >> (n_qubits - 1 - w)) & 1
            out[:, k] = (diag * (1.0 - 2.0 * par.double())).sum(dim=1)
            continue
        low = LoweredTape([ob], 2 * n_qubits)
        plan = get_plan(low, N.PLAN_NO_FUSION)
        scratch = rho_vec.clone()
        for b0 in range(0, B, 65535):
            N.apply_inplace64(plan, None, scratch[b0:b0 + 65535])
        out[:, k] = torch.diagonal(scratch.view(B, D, D), dim1=1, dim2=2).real.sum(dim=1)
    return out


def _simulate_mixed(tape: Sequence[Operation], n_qubits: int, type: str, obs, B: int, x64: bool = False):
    """Noisy tape -> measurement, with rho evolved as a 2n-wire pure register
    (``simulation.py:106-128`` ``simulate_mixed`` + ``:204-271`` ``measure_state``)."""
    if n_qubits > MAX_DENSITY_QUBITS:
        raise NotImplementedError(
            f"density-matrix simulation supports at most {MAX_DENSITY_QUBITS} qubits "
            f"(got {n_qubits})")
    if type == "state":
        raise ValueError(
            "Measurement type 'state' is not defined for mixed (noisy) circuits. "
            "Use 'density' instead."
        )
    N.require_gpu()
    if x64:
        return _density_measure_x64(_evolve_density(tape, n_qubits, B, True), n_qubits, type, obs)
    return measure_density_vec(_evolve_density(tape, n_qubits, B), n_qubits, type, obs)


def measure_density_vec(rho_vec, n_qubits: int, type: str, obs):
    """Measurement of vec(rho) [B, 4^n] (complex64): the density matrices themselves, their diagonals, or
    Tr(O rho) per observable (``simulation.py:274-317`` ``measure_density``)."""
    torch = N.require_gpu()
    B, D = rho_vec.shape[0], 1 << n_qubits
    if type == "density":
        return rho_vec.view(B, D, D)
    if type == "probs":
        return N.density_probs(rho_vec, n_qubits)
    obs = list(obs)
    if not obs:
        return torch.empty((B, 0), dtype=torch.float32, device=rho_vec.device)
    return _density_expval(rho_vec, n_qubits, obs)


class LoweredTape:
    """Engine view of a tape: op list, per-slot values, const blob, structure key."""

    def __init__(self, tape: Sequence[Operation], n_qubits: int):
        self.ops, self.values, blobs, blobs64 = [], [], [], []
        self.ops_periodic = []  # per slot: the gate is 4 pi-periodic in this angle
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
                self.ops_periodic.append(name != "DIAG_ALL")  # Golomb: exp(-i marks x), any marks
            off = -1
            if blob is not None:
                off = const_len
                blobs64.append(np.asarray(blob, dtype=np.float64).reshape(-1))
                blobs.append(blobs64[-1].astype(np.float32))
                const_len += blobs[-1].size
                h.update(blobs64[-1].tobytes())  # (the complex128 engine reads the float64 copy)
            self.ops.append((name, list(wires), slots, off))
            h.update(f"{name}{wires}{len(slots)}{off};".encode())
        self.consts = np.concatenate(blobs) if blobs else np.zeros(0, dtype=np.float32)
        self.consts64 = np.concatenate(blobs64) if blobs64 else np.zeros(0, dtype=np.float64)
        self.n_slots = len(self.values)
        self.n_qubits = n_qubits
        self.key = h.hexdigest()

    def angle_table(self, batch: int, dtype=np.float32) -> np.ndarray:
        table = np.empty((batch, max(1, self.n_slots)), dtype=dtype)
        if self.n_slots == 0:
            table[:] = 0
        four_pi = 4.0 * np.pi
        for j, v in enumerate(self.values):
            col = np.asarray(v, dtype=np.float64)
            if col.ndim and col.shape[0] != batch:
                raise ValueError(f"parameter column has batch {col.shape[0]}, expected {batch}")
            if self.ops_periodic[j] and np.any(np.abs(col) > four_pi):
                # gates depend on angle / 2 only: reduce in fp64 before the float32 cast
                col = col - four_pi * np.rint(col / four_pi)
            table[:, j] = col
        return table[:, : self.n_slots] if self.n_slots else table[:, :0]


def _current_device() -> int:
    try:
        import torch

        return int(torch.cuda.current_device()) if torch.cuda.is_available() else -1
    except Exception:  # pragma: no cover
        return -1


def get_plan(low: LoweredTape, flags: Optional[int] = None) -> N.Plan:
    flags = PLAN_FLAGS if flags is None else flags
    # a plan's device image lives on the GPU of its first run (libqmle_sv refuses another one):
    # the cache is keyed by the current device, so a process that switches torch.cuda devices
    # gets one plan per device instead of QMLE_ERR_UNSUPPORTED from a cached one
    key = (low.key, flags, _current_device())
    plan = _PLAN_CACHE.get(key)
    if plan is None:
        plan = N.Plan(low.ops, low.n_qubits, low.n_slots, low.consts, flags)
        if low.consts64.size:  # explicit matrices at full precision for complex128 runs of this plan
            plan.set_consts64(low.consts64)
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


def sample_shots(probs, n_qubits: int, type: str, obs: Sequence[Operation], shots: int, key,
                 row_offset: int = 0):
    """Exact probabilities [B, 2^n] (device) -> shot estimates (``simulation.py:320-377``):
    ``counts / shots`` or, per observable, ``diag(O) . counts / shots`` (exact for diagonal
    observables, the computational-basis estimate otherwise).  Row ``b`` uses the Philox
    stream ``(key, row_offset + b)`` -- the reference splits the key per batch element
    (``script.py:480-482``)."""
    if type not in ("probs", "expval"):
        raise ValueError(
            f"Shot simulation is only supported for 'probs' and 'expval', got {type!r}.")
    from .utils import key_to_seed

    _, est = N.sample_counts(probs, int(shots), key_to_seed(key), row_offset)
    if type == "probs":
        return est
    torch = N.require_gpu()
    obs = list(obs)
    if not obs:
        return torch.empty((probs.shape[0], 0), dtype=torch.float32, device=probs.device)
    specs = []
    for ob in obs:
        if z_parity_mask(ob) is not None:
            specs.append((ob.wires, None))
        else:
            specs.append((ob.wires, np.real(np.diag(np.asarray(ob.matrix)))))
    return N.probs_diag_expval(est, specs)


def simulate_and_measure(tape: Sequence[Operation], n_qubits: int, type: str,
                         obs: Sequence[Operation] = (), use_density: bool = False,
                         shots: Optional[int] = None, key=None, batch: Optional[int] = None,
                         as_tensor: bool = False, row_offset: int = 0):
    """Run the tape from |0..0> and measure.  Returns ``(B, ...)`` (numpy unless
    ``as_tensor``); the caller strips the batch axis for un-batched execution.  With
    ``shots`` the exact probabilities are sampled on the device (``probs`` / ``expval``
    only; other types stay exact, ``simulation.py:191-201``)."""
    if type not in MEAS_TYPES:
        raise ValueError(f"Unknown measurement type: {type!r}")
    torch = N.require_gpu()
    B = int(batch) if batch is not None else _tape_batch(tape)
    sampled = shots is not None and type in ("probs", "expval")
    if any(isinstance(o, KrausChannel) for o in tape):
        from .utils import x64_enabled

        res = _simulate_mixed(tape, n_qubits, "probs" if sampled else type, obs, B, x64=x64_enabled())
        if sampled:  # (x64: the sampler takes the float64 probabilities rounded once to float32)
            res = sample_shots(res.float() if res.dtype != torch.float32 else res, n_qubits, type, obs,
                               shots, key, row_offset)
        return res if as_tensor else N.to_host(res)
    low = LoweredTape(tape, n_qubits)
    from .utils import x64_enabled

    if x64_enabled():
        # gate by gate, like the reference (no products of neighbouring 1-qubit gates): the
        # accuracy mode keeps the reference's operation order as well as its precision
        # (ONE compile: the default-flag plan of the complex64 engine is not built here)
        plan = get_plan(low, (PLAN_FLAGS or 0) | N.PLAN_NO_MERGE)
        if sampled:  # shots: drawn from the complex128 probabilities, rounded once to float32
            res = sample_shots(_simulate_x64(plan, low, B, n_qubits, "probs", []).float(), n_qubits, type,
                               obs, shots, key, row_offset)
        else:
            res = _simulate_x64(plan, low, B, n_qubits, type, list(obs))
        return res if as_tensor else N.to_host(res)
    plan = get_plan(low)
    angles = torch.from_numpy(low.angle_table(B)).cuda()
    if sampled:
        res = sample_shots(plan.run(angles, "probs"), n_qubits, type, obs, shots, key, row_offset)
        return res if as_tensor else N.to_host(res)
    if type == "expval":
        obs = list(obs)
        masks = [z_parity_mask(o) for o in obs]
        if obs and all(m is not None and len(m) == 1 for m in masks):
            res = plan.run(angles, "expval", [m[0] for m in masks])
        elif not obs:
            res = torch.empty((B, 0), dtype=torch.float32, device=angles.device)
        elif all(m is not None for m in masks) and len(obs) <= 32:
            res = plan.run_parity(angles, masks)  # Z (x) Z ..: out of the last pass as well
        else:
            res = _general_expval(plan.run(angles, "state"), n_qubits, obs)
    else:
        res = plan.run(angles, type)
    if as_tensor:
        return res
    return N.to_host(res)


def _simulate_x64(plan: N.Plan, low: "LoweredTape", B: int, n_qubits: int, type: str, obs):
    """complex128 execution (``utils.enable_x64`` -- the reference's ``jax_enable_x64`` mode):
    float64 angle table, ``qmle_run_batch_f64``.  Z / Z-parity observables are measured by the
    engine; any other observable is applied to the complex128 state with torch (dense ``einsum``
    over the observable's wires, ``simulation.py:263-269``)."""
    torch = N.require_gpu()
    angles = torch.from_numpy(low.angle_table(B, dtype=np.float64)).cuda()
    if type != "expval":
        return plan.run64(angles, type)
    if not obs:
        return torch.empty((B, 0), dtype=torch.float64, device=angles.device)
    masks = [z_parity_mask(o) for o in obs]
    if all(m is not None for m in masks) and len(obs) <= 32:
        return plan.run64(angles, "expval", masks)
    return _x64_observables(plan.run64(angles, "state"), n_qubits, obs, masks)


def _x64_observables(states, n_qubits: int, obs, masks):
    """<psi|O|psi> of complex128 states [B, 2^n] that are already in hand: Z / Z-parity observables as
    signed sums of |psi|^2 (no second run of the circuit per observable), anything else as a dense
    contraction over the observable's wires."""
    torch = N.require_gpu()
    B = states.shape[0]
    psi = states.reshape((B,) + (2,) * n_qubits)
    prob = None
    cols = []
    for ob, m in zip(obs, masks):
        if m is not None:
            if prob is None:
                prob = (states.real ** 2 + states.imag ** 2).reshape((B,) + (2,) * n_qubits)
            # (-1)^(bit_w) along every wire of the parity: contract those axes with (1, -1)
            sgn = torch.tensor([1.0, -1.0], dtype=torch.float64, device=states.device)
            t = prob
            for w in sorted(m, reverse=True):
                t = torch.tensordot(t, sgn, dims=([1 + w], [0]))
            cols.append(t.reshape(B, -1).sum(dim=1))
            continue
        k = len(ob.wires)
        M = torch.from_numpy(np.asarray(ob.matrix, dtype=np.complex128)).to(psi.device).reshape((2,) * (2 * k))
        axes = [1 + w for w in ob.wires]
        moved = torch.movedim(psi, axes, list(range(1, 1 + k)))
        flat = moved.reshape(B, 2**k, -1)
        applied = torch.einsum("rc,bcx->brx", M.reshape(2**k, 2**k), flat)
        cols.append(torch.sum(torch.conj(flat) * applied, dim=(1, 2)).real)
    return torch.stack(cols, dim=1)


def run_expval_table(plan: N.Plan, table: np.ndarray, obs: Sequence[Operation], n_qubits: int,
                     max_rows: int = 1 << 16) -> np.ndarray:
    """Expectation values for every row of an explicit angle table (parameter-shift batches)."""
    torch = N.require_gpu()
    masks = [z_parity_mask(o) for o in obs]
    single_z = bool(obs) and all(m is not None and len(m) == 1 for m in masks)
    parity = bool(obs) and all(m is not None for m in masks) and len(obs) <= 32
    out = []
    from . import memory
    if table.dtype == np.float64:  # the complex128 engine (x64 mode)
        chunk = max(1, memory.compute_chunk_size(n_qubits, min(max_rows, table.shape[0]), "expval", False,
                                                 len(obs), n_ops=plan.n_ops, x64=True, general_obs=not parity))
        for r0 in range(0, table.shape[0], chunk):
            ang = torch.from_numpy(np.ascontiguousarray(table[r0:r0 + chunk])).cuda()
            if parity:
                out.append(plan.run64(ang, "expval", masks).cpu().numpy())
            else:  # PauliX / Hermitian ...: contracted with the complex128 states, like _simulate_x64
                out.append(_x64_observables(plan.run64(ang, "state"), n_qubits, list(obs), masks).cpu().numpy())
        return np.concatenate(out, axis=0)
    chunk = memory.compute_chunk_size(n_qubits, min(max_rows, table.shape[0]),
                                      "expval" if (single_z or parity) else "state", False, len(obs),
                                      n_ops=plan.n_ops)
    for r0 in range(0, table.shape[0], chunk):
        ang = torch.from_numpy(np.ascontiguousarray(table[r0:r0 + chunk])).cuda()
        if single_z:
            res = plan.run(ang, "expval", [m[0] for m in masks])
        elif parity:
            res = plan.run_parity(ang, masks)
        else:
            res = _general_expval(plan.run(ang, "state"), n_qubits, list(obs))
        out.append(res.cpu().numpy())
    return np.concatenate(out, axis=0)
