"""Entangling capability: Meyer-Wallach measure on statevectors.

API mirror of ``Entanglement.meyer_wallach`` / ``_compute_meyer_wallach_meas``
(``qml_essentials/entanglement.py:17-103``).  The reference traces out one qubit
at a time from full density matrices (``jaqsi.partial_trace``) and squares
2^(n-1)-dimensional matrices.  For pure states ``Tr rho_{not j}^2 = Tr rho_j^2``
(Schmidt decomposition) and ``rho_j`` is 2x2, so the HIP kernel only needs, per
wire, the two populations and one cross term: ``a^2 + d^2 + 2|c|^2``.

``bell_measurements`` (``entanglement.py:106-219``) and ``concentratable_entanglement``
(``:471-576``) build 2n- / 3n-qubit circuits from shifted copies of the model circuit
(``tape.copy_to_tape``) and read marginal probabilities -- same kernels at a larger
register.  ``relative_entropy`` / ``entanglement_of_formation``
need matrix logarithms / eigendecompositions of density matrices and stay out of scope.
"""
from __future__ import annotations

import logging
from typing import Any, Optional

import numpy as np

from . import _native as N
from . import distributed
from . import jaqsi as js
from . import operations as op
from .model import Model
from .tape import copy_to_tape

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

    # ------------------------------------------------------------------ multi-register
    @staticmethod
    def _sample_params(model: Model, n_samples, random_key):
        if n_samples is not None and n_samples > 0:
            model.initialize_params(random_key, repeat=int(n_samples))
        params = np.asarray(model.params)
        return params.reshape(1, *params.shape) if params.ndim <= 2 else params

    @staticmethod
    def _register_states(script: js.Script, params, inputs, kwargs):
        """States of the multi-register circuit for every parameter set: (S, 2^m) on device."""
        kwargs = {k: v for k, v in kwargs.items() if k not in ("inputs", "execution_type")}
        with distributed.local_only():
            if params.shape[0] > 1:
                return script.execute(type="state", args=(params, inputs, None, None),
                                      kwargs=kwargs, in_axes=(0, None, None, None),
                                      as_tensor=True)
            return script.execute(type="state", args=(params, inputs, None, None), kwargs=kwargs,
                                  as_tensor=True).reshape(1, -1)

    @classmethod
    def bell_measurements(cls, model: Model, n_samples: int, random_key=None,
                          scale: bool = False, **kwargs: Any) -> float:
        """Entangling capability from Bell measurements on two copies of the state
        (``entanglement.py:106-219``): CX(q, q+n), H(q), then 1 - 2 P(|11>) per pair."""
        if kwargs.get("noise_params"):
            log.warning("Bell Measurements not suitable for noisy circuits. "
                        "Consider 'concentratable entanglement' instead.")
        n = model.n_qubits
        if scale:
            n_samples = (2**n) * n_samples

        def bell_circuit(params, inputs, pulse_params=None, random_key=None, **kw):
            def vari():
                model._variational(params, inputs, **kw)

            vari()
            copy_to_tape(vari, offset=n)
            for q in range(n):
                op.CX(wires=[q, q + n])
                op.H(wires=q)

        params = cls._sample_params(model, n_samples, random_key)
        inputs = model._inputs_validation(kwargs.get("inputs", None))
        states = cls._register_states(js.Script(f=bell_circuit, n_qubits=2 * n), params, inputs,
                                      kwargs)
        # P(|11>) of wires (q, q+n) = last entry of the 2-wire marginal (jaqsi.py:141-146)
        exp = np.stack([1 - 2 * N.marginal_probs(states, [q, q + n])[:, -1].cpu().numpy()
                        for q in range(n)], axis=-1)  # (S, n)
        measure = 2 * (1 - exp.mean(axis=0))
        return min(max(float(measure.mean()), 0.0), 1.0)

    @classmethod
    def concentratable_entanglement(cls, model: Model, n_samples: int, random_key=None,
                                    scale: bool = False, **kwargs: Any) -> float:
        """Concentratable entanglement (arXiv:2104.06923) by a swap test on 3n qubits
        (``entanglement.py:471-576``): 1 - P(ancilla register = 0...0)."""
        n = model.n_qubits
        if scale:
            n_samples = (2**n) * n_samples

        def swap_test(params, inputs, pulse_params=None, random_key=None, **kw):
            def vari():
                model._variational(params, inputs, **kw)

            copy_to_tape(vari, offset=n)
            copy_to_tape(vari, offset=2 * n)
            for i in range(n):
                op.H(wires=i)
            for i in range(n):
                op.CSWAP(wires=[i, i + n, i + 2 * n])
            for i in range(n):
                op.H(wires=i)

        params = cls._sample_params(model, n_samples, random_key)
        inputs = model._inputs_validation(kwargs.get("inputs", None))
        states = cls._register_states(js.Script(f=swap_test, n_qubits=3 * n), params, inputs,
                                      kwargs)
        p0 = N.marginal_probs(states, list(range(n)))[:, 0].cpu().numpy()
        return float((1 - p0).mean())

    @classmethod
    def concentratable_entanglement_estimation(cls, model: Model, n_samples: int, random_key=None,
                                               scale: bool = False, **kwargs: Any) -> float:
        """Concentratable entanglement from Bell-basis measurements on TWO copies (2n qubits
        instead of the swap test's 3n; ``entanglement.py:579-684``):
        ``1 - <(1/2^n) prod_i (I + SWAP_i)>``.  After CX(i, i+n) H(i) every ``I + SWAP_i`` is
        ``diag(2, 2, 2, 0)``, so the expectation value is the probability that no pair
        ``(i, i+n)`` reads ``11`` -- evaluated matrix-free on the 4^n probabilities."""
        n = model.n_qubits
        if scale:
            n_samples = (2**n) * n_samples

        def bell_basis(params, inputs, pulse_params=None, random_key=None, **kw):
            def vari():
                model._variational(params, inputs, **kw)

            copy_to_tape(vari, offset=0)
            copy_to_tape(vari, offset=n)
            for i in range(n):
                op.CX(wires=[i, i + n])
                op.H(wires=i)

        params = cls._sample_params(model, n_samples, random_key)
        inputs = model._inputs_validation(kwargs.get("inputs", None))
        states = cls._register_states(js.Script(f=bell_basis, n_qubits=2 * n), params, inputs,
                                      kwargs)
        x = np.arange(4**n)
        no_11 = (((x >> n) & x & (2**n - 1)) == 0).astype(np.float32)
        vals = []
        for b0 in range(0, states.shape[0], 4096):
            probs = N.probs(states[b0:b0 + 4096])
            vals.append(N.probs_diag_expval(probs, [(list(range(2 * n)), no_11)])[:, 0])
        torch = N.require_gpu()
        ent = 1.0 - torch.cat(vals).double()
        log.debug("Variance of measure: %s", float(ent.var()) if ent.numel() > 1 else 0.0)
        return float(ent.mean())

