"""Circuit container + batch dispatcher -- the drop-in boundary of this build.

API mirror of ``qml_essentials/script.py``: ``Script(f, n_qubits)`` and
``Script.execute(type, obs, args=..., kwargs=..., in_axes=..., shots, key)``
(:137-219) with the batched path of ``_execute_batched`` (:399-553).  Where the
reference builds ``jit(vmap(single_execute))``, this class records the circuit
once with :class:`batching.Batched` arguments and hands the tape + angle table to
``libqmle_sv``.  Drawing (``script.py:555-627``) is out of scope.
"""
from __future__ import annotations

from typing import Any, Callable, List, Optional, Tuple

import numpy as np

from . import _native as N
from . import distributed, memory, simulation
from .batching import Batched, to_numpy
from .operations import KrausChannel, Operation, z_parity_mask
from .tape import batch_context, recording
from .utils import PRNGKey


class NotAffine(Exception):
    """A gate angle is not an affine function of the device-resident arguments."""


class CompiledCall:
    """One traced circuit whose gate angles are affine in its device-resident arguments.

    Counterpart of the reference's cached ``jit(vmap(...))`` executable
    (``script.py:272-329,469-553``): tracing happens once on a two-row probe; afterwards a
    call is three kernels' worth of work on the GPU -- ``qmle_build_angles`` (angle table
    from the resident ``params`` / ``inputs`` tensors), the plan's passes, the measurement --
    and nothing per sample on the host.
    """

    def __init__(self, script: "Script", type: str, obs, args: tuple, leaf_ids: Tuple[int, ...],
                 kwargs: dict):
        import torch

        self.type, self.obs = type, list(obs)
        rng = np.random.default_rng(12345)
        probes, wrapped = {}, list(args)
        for k in leaf_ids:  # args[k]: one host row of the device tensor, per-sample shape
            row = np.asarray(args[k], dtype=np.float64)
            pr = np.stack([row, row + rng.uniform(0.1, 1.0, row.shape)])
            probes[k] = pr
            wrapped[k] = Batched.leaf(pr, k)
        tape = script._record(*wrapped, **kwargs)
        self.n_qubits = script._n_qubits or simulation.infer_n_qubits(tape, obs)
        # A tape with noise channels runs on vec(rho), a pure "state" of the doubled register
        # (simulation.doubled_tape): the same plan / angle-map machinery, measured as a density matrix.
        # Channels on 3 or 4 wires (applied Kraus operator by Kraus operator) keep the recorded path.
        self.density = any(isinstance(o, KrausChannel) for o in tape)
        n_reg = self.n_qubits
        if self.density:
            if type == "state" or self.n_qubits > simulation.MAX_DENSITY_QUBITS:
                raise NotAffine("noisy tape")  # (the recorded path raises the reference's messages)
            tape = simulation.doubled_tape(tape, self.n_qubits)
            if any(isinstance(it, simulation._WideChannel) for it in tape):
                raise NotAffine("wide channel")
            n_reg = 2 * self.n_qubits
        self.n_register = n_reg
        low = simulation.LoweredTape(tape, n_reg)
        self.plan = simulation.get_plan(low)
        self.n_slots = low.n_slots
        values = np.stack([np.broadcast_to(np.asarray(v, dtype=np.float64), (2,))
                           for v in low.values], axis=1) if low.values else np.zeros((2, 0))
        ptr, arg, idx, coef, const = [0], [], [], [], []
        slot = 0
        order = {k: i for i, k in enumerate(leaf_ids)}
        for op_ in tape:
            if op_.lower(n_reg) is None:
                continue
            tans = op_.parameter_tangents
            for j in range(len(op_.lower(n_reg)[2])):  # one slot per lowered parameter
                t = tans[j] if j < len(tans) else []
                if t is None:
                    raise NotAffine(op_.name)
                c0 = values[0, slot]
                c1 = values[1, slot]
                for lid, flat, cf in t:
                    cf = np.broadcast_to(np.asarray(cf, dtype=np.float64), (2,))
                    if abs(cf[0] - cf[1]) > 1e-12 * max(1.0, abs(cf[0])):
                        raise NotAffine(op_.name)
                    arg.append(order[lid]); idx.append(int(flat)); coef.append(float(cf[0]))
                    c0 -= cf[0] * probes[lid][0].reshape(-1)[flat]
                    c1 -= cf[0] * probes[lid][1].reshape(-1)[flat]
                if abs(c0 - c1) > 1e-9 * max(1.0, abs(c0), abs(values[0, slot])):
                    raise NotAffine(op_.name)
                const.append(c0)
                ptr.append(len(arg))
                slot += 1
        dev = torch.device("cuda", torch.cuda.current_device())
        i32 = lambda v: torch.tensor(v if len(v) else [0], dtype=torch.int32, device=dev)
        f32 = lambda v: torch.tensor(v if len(v) else [0.0], dtype=torch.float32, device=dev)
        self.d_ptr, self.d_arg, self.d_idx = i32(ptr), i32(arg), i32(idx)
        self.d_coef, self.d_const = f32(coef), f32(const)
        # rotation angles wrap at 4 pi (the gates depend on angle / 2); a DIAG_ALL scale does not
        self.d_period = torch.tensor([4.0 * np.pi if per else 0.0 for per in low.ops_periodic] or [0.0],
                                     dtype=torch.float64, device=dev)
        self.leaf_ids = leaf_ids
        self._low, self._map = low, (ptr, arg, idx, coef)   # host copies for the adjoint path
        self._leaf_sizes = {order[k]: int(np.prod(np.shape(args[k]))) for k in leaf_ids}
        self._adj = None

    def run(self, leaves, divs, mods, batch: int, batch_offset: int = 0, meas: Optional[str] = None,
            shots: Optional[int] = None, key=None):
        """leaves: contiguous float32 CUDA tensors [rows_k, ...] in ``leaf_ids`` order.  ``meas``
        overrides the call's measurement with another one of the same circuit ("mw": Meyer-Wallach
        out of the producing pass, for a call compiled for "state").  ``shots``: "probs" / "expval"
        are estimated from that many draws of the exact probabilities (row b on the Philox stream
        ``(key, batch_offset + b)``, ``simulation.sample_shots``)."""
        if shots is not None and meas is None and self.type in ("probs", "expval"):
            probs = self.run(leaves, divs, mods, batch, batch_offset, meas="probs")
            return simulation.sample_shots(probs, self.n_qubits, self.type, self.obs, shots, key, batch_offset)
        strides = [t[0].numel() if t.shape[0] else 0 for t in leaves]
        if self.density:
            if meas not in (None, self.type, "probs"):
                raise NotImplementedError(f"measurement {meas!r} of a noisy circuit")
            if self.n_slots:
                am = getattr(self, "_amap", None)
                if am is None:
                    am = self._amap = N.AngleMapArgs(len(self.leaf_ids), self.d_ptr, self.d_arg, self.d_idx,
                                                     self.d_coef, self.d_const, self.d_period)
                rho = self.plan.run_map(am.set(leaves, strides, divs, mods, batch_offset), batch, "state", ())
            else:
                import torch
                rho = self.plan.run(torch.zeros((batch, 0), dtype=torch.float32, device=self.d_const.device), "state")
            return simulation.measure_density_vec(rho, self.n_qubits, meas or self.type, self.obs)
        kind = meas if meas is not None and meas != self.type else self.type
        arg = ()
        if kind == "expval":
            how, arg = self._measure()
            if how != "z":
                kind = None
        if kind is not None and self.n_slots:
            # one native call: angle table (scratch) + the batch -- no host time between the two launches
            am = getattr(self, "_amap", None)
            if am is None:
                am = self._amap = N.AngleMapArgs(len(self.leaf_ids), self.d_ptr, self.d_arg, self.d_idx, self.d_coef,
                                                 self.d_const, self.d_period)
            return self.plan.run_map(am.set(leaves, strides, divs, mods, batch_offset), batch, kind, arg)
        angles = N.build_angles(leaves, strides, divs, mods, self.d_ptr, self.d_arg, self.d_idx,
                                self.d_coef, self.d_const, self.n_slots, batch, batch_offset,
                                d_period=self.d_period)
        if self.n_slots == 0:
            import torch
            angles = torch.zeros((batch, 0), dtype=torch.float32, device=self.d_const.device)
        if meas is not None and meas != self.type:
            return self.plan.run(angles, meas)
        if self.type == "expval":
            how, arg = self._measure()
            if how == "z":
                return self.plan.run(angles, "expval", arg)
            if how == "parity":
                return self.plan.run_parity(angles, arg)
            return simulation._general_expval(self.plan.run(angles, "state"), self.n_qubits, self.obs)
        return self.plan.run(angles, self.type)

    def _measure(self):
        """How the observables are measured (fixed per compiled call): ("z", wires) for plain Z's,
        ("parity", wire groups) for Z-parities, ("general", None) otherwise."""
        m = getattr(self, "_meas", None)
        if m is None:
            masks = [z_parity_mask(o) for o in self.obs]
            if self.obs and all(k is not None and len(k) == 1 for k in masks):
                m = ("z", [k[0] for k in masks])
            elif self.obs and all(k is not None for k in masks) and len(masks) <= 32:
                m = ("parity", masks)
            else:
                m = ("general", None)
            self._meas = m
        return m

    def _adjoint_setup(self):
        """Reverse tape, generator terms and the chain-rule matrices -- built once."""
        import torch

        from . import adjoint

        ptr, arg, idx, coef = self._map
        low = self._low
        want = [ptr[s_ + 1] > ptr[s_] for s_ in range(low.n_slots)]
        rev_ops, terms, rev_src = adjoint.build_reverse(low, adjoint._op_blobs(low), want)
        rev = simulation.LoweredTape(rev_ops, self.n_qubits)
        dev = self.d_const.device
        mats = []
        for k in range(len(self.leaf_ids)):  # angle_s = const_s + sum_t coef_t * leaf[arg_t][idx_t]
            m = np.zeros((max(1, low.n_slots), self._leaf_sizes[k]), dtype=np.float32)
            for s_ in range(low.n_slots):
                for t in range(ptr[s_], ptr[s_ + 1]):
                    if arg[t] == k:
                        m[s_, idx[t]] += coef[t]
            mats.append(torch.from_numpy(m).to(dev))
        self._adj = dict(
            rev=rev,
            terms=adjoint.patch_marks(rev, terms),
            perm=torch.tensor(rev_src if rev_src else [0], dtype=torch.int64, device=dev),
            n_rev_slots=rev.n_slots, mats=mats)
        return self._adj

    def vjp(self, leaves, divs, mods, batch: int, weights):
        """Adjoint gradient of ``sum_k weights[b, k] <obs_k>_b`` with respect to the device
        leaves -> list of CUDA tensors shaped like ``leaves`` (a leaf shared by the whole batch
        receives the sum over the batch).  Angle table, forward pass, backward sweep and the
        chain rule all stay on the GPU."""
        import torch

        if self.density:
            raise NotImplementedError("adjoint differentiation of a noisy circuit")
        masks = [z_parity_mask(o) for o in self.obs]
        if self.type != "expval" or not self.obs or any(m is None for m in masks):
            raise NotImplementedError("adjoint differentiation needs Z / Z-parity observables")
        adj = self._adj or self._adjoint_setup()
        strides = [int(np.prod(t.shape[1:], dtype=np.int64)) for t in leaves]
        angles = N.build_angles(leaves, strides, divs, mods, self.d_ptr, self.d_arg, self.d_idx,
                                self.d_coef, self.d_const, self.n_slots, batch, 0,
                                d_period=self.d_period)
        if adj["n_rev_slots"]:
            rev_angles = (-angles.index_select(1, adj["perm"])).contiguous()
        else:
            rev_angles = torch.zeros((batch, 1), dtype=torch.float32, device=angles.device)
        from . import adjoint

        d = adjoint.run_sweep(self.plan, adj["rev"], angles, rev_angles, weights, masks,
                              adj["terms"], max(1, self.n_slots))               # [B, n_slots]
        out = []
        for k, leaf in enumerate(leaves):
            g = d @ adj["mats"][k]                                               # [B, leaf size]
            rows = int(leaf.shape[0])
            if rows == 1:
                g = g.sum(dim=0, keepdim=True)
            elif not (divs[k] == 1 and rows == batch):
                row = (torch.arange(batch, device=g.device) // int(divs[k])) % int(mods[k])
                g = torch.zeros((rows, g.shape[1]), dtype=g.dtype, device=g.device).index_add_(0, row, g)
            out.append(g.reshape(tuple(leaf.shape)))
        return out


class Script:
    """Wraps a circuit function ``f(*args, **kwargs)`` that instantiates operations."""

    def __init__(self, f: Callable[..., None], n_qubits: Optional[int] = None) -> None:
        self.f = f
        self._n_qubits = n_qubits
        self._compiled: dict = {}  # structure key -> CompiledCall (device-resident path)

    def compiled(self, key, type: str, obs, args: tuple, leaf_ids: Tuple[int, ...],
                 kwargs: Optional[dict] = None) -> CompiledCall:
        """Fetch / build the :class:`CompiledCall` for ``key`` (raises :class:`NotAffine`)."""
        cc = self._compiled.get(key)
        if cc is None:
            if callable(args):  # probe values are only needed to build the call
                args = args()
            cc = CompiledCall(self, type, obs, args, leaf_ids, kwargs or {})
            if len(self._compiled) > 64:
                self._compiled.pop(next(iter(self._compiled)))
            self._compiled[key] = cc
        return cc

    def _record(self, *args, **kwargs) -> List[Operation]:
        with recording() as tape:
            self.f(*args, **kwargs)
        return tape

    @staticmethod
    def _batch_size(args: tuple, in_axes: Tuple) -> int:
        for a, ax in zip(args, in_axes):
            if ax is not None:
                return int(np.shape(a)[ax])
        return 1

    def execute(self, type: str = "expval", obs: Optional[List[Operation]] = None, *,
                args: tuple = (), kwargs: Optional[dict] = None, in_axes: Optional[Tuple] = None,
                shots: Optional[int] = None, key=None, as_tensor: bool = False):
        """Execute the circuit; with ``in_axes`` the result has a leading batch axis."""
        obs = [] if obs is None else obs
        kwargs = {} if kwargs is None else kwargs
        if shots is not None and key is None:
            key = PRNGKey(0)  # script.py:189-190
        if in_axes is not None:
            return self._execute_batched(type, obs, args, kwargs, in_axes, shots, key, as_tensor)
        tape = self._record(*[to_numpy(a) for a in args], **kwargs)
        n_qubits = self._n_qubits or simulation.infer_n_qubits(tape, obs)
        res = simulation.simulate_and_measure(
            tape, n_qubits, type, obs, simulation.uses_density(tape, type), shots=shots, key=key,
            as_tensor=as_tensor,
        )
        if simulation._tape_batch(tape) == 1:
            res = res[0]
        return res

    def _execute_batched(self, type, obs, args, kwargs, in_axes, shots=None, key=None,
                         as_tensor: bool = False):
        if len(in_axes) != len(args):
            raise ValueError(
                f"in_axes has {len(in_axes)} entries but args has {len(args)}. "
                "Provide one in_axes entry per positional argument."
            )
        args = tuple(to_numpy(a) for a in args)
        batch_size = self._batch_size(args, in_axes)
        n_obs = len(obs)

        def run(start: int, end: int):
            wrapped = []
            for a, ax in zip(args, in_axes):
                if ax is None or a is None:
                    wrapped.append(a)
                else:
                    wrapped.append(Batched(np.moveaxis(np.asarray(a), ax, 0)[start:end]))
            with batch_context(end - start):
                tape = self._record(*wrapped, **kwargs)
            n_qubits = self._n_qubits or simulation.infer_n_qubits(tape, obs)
            return simulation.simulate_and_measure(
                tape, n_qubits, type, obs, simulation.uses_density(tape, type), shots=shots,
                key=key, batch=end - start, as_tensor=as_tensor, row_offset=start,
            ), n_qubits, len(tape)

        # memory-aware chunking needs n_qubits / n_ops: probe with one sample's structure
        n_qubits = self._n_qubits
        n_ops = 1
        use_density = False
        if n_qubits is None or kwargs.get("noise_params"):
            probe = [a if ax is None or a is None else Batched(np.moveaxis(np.asarray(a), ax, 0)[:1])
                     for a, ax in zip(args, in_axes)]
            tape = self._record(*probe, **kwargs)
            n_qubits = n_qubits or simulation.infer_n_qubits(tape, obs)
            n_ops = len(tape)
            use_density = any(isinstance(o, KrausChannel) for o in tape)
        # multi-GPU: this rank simulates one contiguous block of the batch; one
        # all-gather returns the full result everywhere (script.py:443-453)
        lo, hi, sharded = distributed.my_block(batch_size, *args)
        local = hi - lo
        from .utils import x64_enabled

        x64 = x64_enabled() and shots is None
        # (complex128 mode: 16-byte amplitudes, and observables the engine does not measure itself --
        # anything but Z / Z-parities -- keep every sample's state for the contraction)
        general = x64 and type == "expval" and any(z_parity_mask(o) is None for o in (obs or []))
        chunk = memory.compute_chunk_size(n_qubits, local, type, use_density, n_obs, n_ops=n_ops,
                                          x64=x64, general_obs=general)
        if chunk >= local:
            res = run(lo, hi)[0]
        else:
            res = memory.execute_chunked(lambda s, e: run(lo + s, lo + e)[0], local, chunk)
        if sharded:
            res = distributed.all_gather_rows(res, batch_size)
        return res

    # ------------------------------------------------------------------ gradients
    _SHIFT_RULES = {
        # (shift, coefficient) terms:  df/dtheta = sum_k coef_k * f(theta + shift_k)
        "two": ((np.pi / 2, 0.5), (-np.pi / 2, -0.5)),
        # controlled rotations (generator spectrum {0, +-1/2}): four-term rule
        "four": ((np.pi / 2, (np.sqrt(2) + 1) / (4 * np.sqrt(2))),
                 (-np.pi / 2, -(np.sqrt(2) + 1) / (4 * np.sqrt(2))),
                 (3 * np.pi / 2, -(np.sqrt(2) - 1) / (4 * np.sqrt(2))),
                 (-3 * np.pi / 2, (np.sqrt(2) - 1) / (4 * np.sqrt(2)))),
    }

    def _trace_for_gradient(self, obs, args, kwargs, in_axes, argnums):
        """Record the tape with the ``argnums`` arguments as differentiable leaves.  Returns
        ``(tape, lowered tape, n_qubits, B, slots, leaf_shapes, batched)`` where ``slots`` lists
        ``(angle slot, shift rule, tangent terms)`` for every angle that depends on a leaf."""
        kwargs = {} if kwargs is None else kwargs
        args = tuple(to_numpy(a) for a in args)
        batched = in_axes is not None
        if in_axes is None:
            in_axes = (None,) * len(args)
        if len(in_axes) != len(args):
            raise ValueError(
                f"in_axes has {len(in_axes)} entries but args has {len(args)}. "
                "Provide one in_axes entry per positional argument."
            )
        B = self._batch_size(args, in_axes) if batched else 1
        wrapped, leaf_shapes = [], {}
        for k, (a, ax) in enumerate(zip(args, in_axes)):
            if a is None or not hasattr(a, "shape"):
                wrapped.append(a)
                continue
            arr = np.asarray(a, dtype=np.float64)
            data = np.moveaxis(arr, ax, 0) if ax is not None else np.broadcast_to(arr, (B,) + arr.shape)
            if k in argnums:
                wrapped.append(Batched.leaf(np.array(data), k))
                leaf_shapes[k] = data.shape[1:]
            elif ax is not None:
                wrapped.append(Batched(data, []))  # batched, but a constant for this gradient
            else:
                wrapped.append(a)
        tape = self._record(*wrapped, **kwargs)
        n_qubits = self._n_qubits or simulation.infer_n_qubits(tape, obs)
        low = simulation.LoweredTape(tape, n_qubits)

        # differentiable slots: (slot, rule, tangent terms)
        slots, s = [], 0
        for op_ in tape:
            lowered = op_.lower(n_qubits)
            if lowered is None:
                continue
            tans = op_.parameter_tangents
            for j in range(len(lowered[2])):  # one angle slot per lowered parameter
                t = tans[j] if j < len(tans) else []
                if t is None:
                    raise NotImplementedError(
                        f"{op_.name}: parameter is a non-linear function of the arguments; "
                        "cannot apply the chain rule")
                if t:
                    slots.append((s, op_._shift_rule, t, op_.name))
                s += 1
        return tape, low, n_qubits, B, slots, leaf_shapes, batched

    def vjp(self, obs: List[Operation], cotangent, *, args: tuple = (),
            kwargs: Optional[dict] = None, in_axes: Optional[Tuple] = None,
            argnums: Tuple[int, ...] = (0,)):
        """Gradient of ``sum_k cotangent[b, k] * <obs_k>_b`` with respect to the array arguments
        ``argnums`` by ADJOINT differentiation: one backward sweep for all angles
        (:mod:`adjoint`).  ``obs`` must be Z / Z-parity observables; ``cotangent`` has shape
        ``(B, n_obs)`` (``(n_obs,)`` without ``in_axes``).  Returns one array per ``argnums``
        entry of shape ``(B, *arg_shape)`` (no ``B`` axis without ``in_axes``).  A ``cotangent`` with a
        leading axis of K cotangents, ``(K, B, n_obs)``, is served by one trace and one sweep over K * B
        states and returns ``(K, B, *arg_shape)`` (a Jacobian: K = n_obs one-hot rows)."""
        from . import adjoint

        masks = [z_parity_mask(o) for o in obs]
        if not obs or any(m is None for m in masks):
            raise adjoint.AdjointUnsupported("adjoint differentiation needs Z / Z-parity observables")
        tape, low, n_qubits, B, slots, leaf_shapes, batched = self._trace_for_gradient(
            obs, args, kwargs, in_axes, argnums)
        if any(isinstance(o, KrausChannel) for o in tape):
            raise adjoint.AdjointUnsupported("adjoint differentiation of noisy circuits")
        from .utils import x64_enabled

        x64 = x64_enabled()  # complex128 sweep, float64 tables (jax.grad with jax_enable_x64, test_jaqsi.py:57)
        w = np.asarray(cotangent, dtype=np.float64 if x64 else np.float32)
        K = int(w.size // (B * len(obs)))  # K > 1: ``cotangent`` is (K, B, n_obs) -- K cotangents, ONE trace and sweep
        stacked = K > 1 or w.ndim == 3
        w = w.reshape(K * B, len(obs))
        want = [False] * low.n_slots
        for s_, _rule, _t, _name in slots:
            want[s_] = True
        grads = {k: np.zeros((K, B) + tuple(shp)) for k, shp in leaf_shapes.items()}
        if slots:
            d = adjoint.adjoint_slot_gradient(low, n_qubits, B, masks, w, want, x64=x64)  # [K * B, n_slots]
            d = d.reshape(K, B, -1)
            for s_, _rule, tangent, _name in slots:
                for lid, flat, coef in tangent:  # gate angles are scalars: one leaf element each
                    g = grads[lid].reshape(K, B, -1)
                    g[:, :, int(flat)] += d[:, :, s_] * np.asarray(coef, dtype=np.float64).reshape(1, B)
        if not stacked:
            grads = {k: g[0] for k, g in grads.items()}
            return tuple(grads[k] if batched else grads[k][0] for k in argnums)
        return tuple(grads[k] if batched else grads[k][:, 0] for k in argnums)

    def gradient(self, obs: List[Operation], *, args: tuple = (), kwargs: Optional[dict] = None,
                 in_axes: Optional[Tuple] = None, argnums: Tuple[int, ...] = (0,)):
        """Jacobian of ``execute(type="expval", obs=obs, ...)`` with respect to the array
        arguments ``argnums`` by the parameter-shift rule (exact, no finite differences).

        What ``jax.grad`` through ``Script.execute`` provides in the reference
        (``tests/test_jaqsi.py:131-141,764-786``), done the way this engine is good at:
        every shifted circuit is one more row of the batch, so one engine call evaluates
        all ``2 x (#rotation gates)`` (4 per controlled rotation) shifted copies of every
        sample.  Gate angles may be sums / products of the arguments (``inputs * enc_params``);
        the chain rule uses the tangents tracked by :class:`batching.Batched`.

        Returns a tuple with one array per entry of ``argnums`` of shape
        ``(B, n_obs, *arg_shape)`` (``arg_shape`` without its batch axis); ``B = 1`` and the
        axis is dropped when ``in_axes`` is None.
        """
        tape, low, n_qubits, B, slots, leaf_shapes, batched = self._trace_for_gradient(
            obs, args, kwargs, in_axes, argnums)
        from .utils import x64_enabled

        base = low.angle_table(B, dtype=np.float64) if x64_enabled() else low.angle_table(B).astype(np.float64)
        grads = {k: np.zeros((B, len(obs)) + tuple(shp)) for k, shp in leaf_shapes.items()}
        if slots:
            rows = []
            for slot, rule_name, _t, op_name in slots:
                if rule_name is None:
                    raise NotImplementedError(f"{op_name} has no parameter-shift rule")
                for shift, _c in self._SHIFT_RULES[rule_name]:
                    t = base.copy()
                    t[:, slot] += shift
                    rows.append(t)
            if x64_enabled():  # shifted circuits on the complex128 engine, gate by gate like simulate()
                table = np.concatenate(rows, axis=0)
                plan = simulation.get_plan(low, (simulation.PLAN_FLAGS or 0) | N.PLAN_NO_MERGE)
            else:
                table = np.concatenate(rows, axis=0).astype(np.float32)
                plan = simulation.get_plan(low)
            vals = simulation.run_expval_table(plan, table, obs, n_qubits)  # (rows*B, n_obs)
            vals = vals.reshape(-1, B, len(obs))
            r = 0
            for slot, rule_name, tangent, _op_name in slots:
                d = np.zeros((B, len(obs)))
                for _shift, coef in self._SHIFT_RULES[rule_name]:
                    d += coef * vals[r]
                    r += 1
                for lid, flat, coef in tangent:
                    g = grads[lid].reshape(B, len(obs), -1)
                    g[:, :, flat] += d * np.asarray(coef).reshape(-1, 1)
        out = tuple(grads[k] if batched else grads[k][0] for k in argnums)
        return out

    def draw(self, *a, **k):
        raise NotImplementedError("circuit drawing is outside the MI355X hot path")
