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
register.  ``relative_entropy`` (``:222-372``) and
``entanglement_of_formation`` (``:375-468``) take their density matrices from the engine
(``execution_type="density"``) and do what the reference does with them on the host: SciPy's
matrix logarithm per matrix, ``eigh`` + Meyer-Wallach of the eigenvectors.
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
        kwargs.setdefault("inputs", None)
        kwargs.pop("execution_type", None)
        torch = N.require_gpu()
        if n_samples is not None and n_samples > 0:
            # (the compiled call is resolved before the sampler is launched: expressibility.py)
            prep = None
            if not any(v is not None for v in kwargs.values()) and not distributed.enabled():
                prep = model.prepared_state_call(int(n_samples))
            random_key = model.initialize_params(random_key, repeat=int(n_samples))
            if prep is not None and model.device_params() is not None:
                cc, divs, mods, B = prep
                # Meyer-Wallach out of the pass that produces the state (QMLE_MEAS_MEYER_WALLACH): for
                # n <= 14 no statevector is ever stored, above the first of the three reads is saved
                out = cc.run([model.device_params()], divs, mods, B, 0, meas="mw")
                return float(out[:, 0].mean(dtype=torch.float64))
        # sampled parameters stay on the GPU (the reference's are jax device arrays,
        # entanglement.py:52-60); small draws / user-set parameters are host arrays
        params = model.device_params()
        if params is None:
            params = np.asarray(model.params)
        total = params.shape[0]
        lo, hi, sharded = distributed.my_block(total, params, kwargs.get("inputs"))
        local = params[lo:hi] if sharded else params
        ent = cls._fused_meyer_wallach(model, local, kwargs)
        if ent is None:  # noisy / shot / complex128 models: states first, then the stand-alone kernels
            with distributed.local_only():
                states = model._forward(params=local, execution_type="state", as_tensor=True, **kwargs)
            ent = cls._compute_meyer_wallach_meas(states, model.n_qubits)
        model.params = params
        if sharded:
            b_i = model.batch_shape[0]
            ent = distributed.all_gather_rows(ent.reshape(b_i, -1).transpose(0, 1).contiguous(),
                                              total)
        if log.isEnabledFor(logging.DEBUG):  # (a variance costs a reduction and a device -> host sync)
            log.debug("Variance of measure: %s", float(ent.var()) if ent.numel() > 1 else 0.0)
        return float(ent.mean(dtype=torch.float64))  # one reduction, one sync

    @staticmethod
    def _fused_meyer_wallach(model: Model, params, kwargs):
        """Q per sample through ``QMLE_MEAS_MEYER_WALLACH`` (the circuit's last pass reports the sums
        of its own tile; no first read of the state).  None when the model has no compiled device
        call (noise, shots, complex128 mode, non-affine angles): the caller takes the state route."""
        from .utils import _gpu_present, x64_enabled

        if not model.host_arrays_via_device or not _gpu_present():
            return None
        if set(kwargs) - {"inputs", "enc_params"} or model.noise_params is not None \
                or model.shots is not None or (x64_enabled() if model.x64 is None else model.x64):
            return None
        torch = N.require_gpu()
        if not hasattr(params, "is_cuda"):
            params = torch.from_numpy(np.ascontiguousarray(params, dtype=np.float32)).cuda()
        inputs = kwargs.get("inputs")
        if inputs is not None and not hasattr(inputs, "is_cuda"):
            x = np.asarray(inputs, dtype=np.float32)
            inputs = torch.from_numpy(np.ascontiguousarray(x.reshape(-1, model.n_input_feat))).cuda()
        with distributed.local_only():
            got = model._forward_device(params, inputs, kwargs.get("enc_params"), "state", False,
                                        _want_call=True)
        if got is NotImplemented:
            return None
        cc, leaves, divs, mods, B = got
        return cc.run(leaves, divs, mods, B, 0, meas="mw")[:, 0]

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
        script = js.Script(f=bell_circuit, n_qubits=2 * n)
        # P(|11>) of wires (q, q+n) -- the last entry of the 2-wire marginal (jaqsi.py:141-146) -- is
        # <(1 - Z_q)(1 - Z_{q+n})> / 4: three Z-parities per pair, which the engine sums in the pass that
        # finishes the state (round 5: no 4^n statevector per sample is stored, no marginal pass per pair reads
        # them back).  <= 30 parities per run (the engine's 32-observable limit): 10 pairs.
        kw = {k: v for k, v in kwargs.items() if k not in ("inputs", "execution_type")}
        cols = []
        for q0 in range(0, n, 10):
            pairs = range(q0, min(n, q0 + 10))
            obs = []
            for q in pairs:
                obs += [op.PauliZ(wires=q, record=False), op.PauliZ(wires=q + n, record=False),
                        js.build_parity_observable([q, q + n])]
            with distributed.local_only():
                if params.shape[0] > 1:
                    ev = script.execute(type="expval", obs=obs, args=(params, inputs, None, None), kwargs=kw,
                                        in_axes=(0, None, None, None), as_tensor=True)
                else:
                    ev = script.execute(type="expval", obs=obs, args=(params, inputs, None, None), kwargs=kw,
                                        as_tensor=True).reshape(1, -1)
            ev = ev.double().reshape(-1, len(pairs), 3)
            cols.append((1.0 + ev[..., 0] + ev[..., 1] - ev[..., 2]) / 2.0)  # 1 - 2 P(|11>) per pair
        torch = N.require_gpu()
        exp = torch.cat(cols, dim=1)  # (S, n)
        measure = 2 * (1 - exp.mean(dim=0))
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
        # P(ancilla register = 0...0): the ancillas are wires 0..n-1, the leading bits of the basis index, so the
        # amplitudes in question are the first 4^n of every state -- one slice instead of a marginal over all 8^n
        torch = N.require_gpu()
        head = torch.view_as_real(states.reshape(states.shape[0], -1)[:, : 4**n]).double()
        p0 = (head * head).sum(dim=(1, 2))
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



    # ------------------------------------------------------------------ density-matrix measures
    @classmethod
    def _compute_log_density(cls, model: Model, **kwargs):
        """(rho, log2 rho) of the model's output state (``entanglement.py:307-327``); the matrix
        logarithm is SciPy's, one matrix at a time, as ``math.logm_v`` (``math.py:7-28``)."""
        kwargs.setdefault("inputs", None)
        kwargs.pop("execution_type", None)
        rho = np.asarray(js._host(model(execution_type="density", **kwargs)))
        dim = 2**model.n_qubits
        rho = rho.reshape(-1, dim, dim)
        return rho, logm_v(rho) / np.log(2)

    @classmethod
    def _compute_rel_entropies(cls, rhos, log_rhos, log_sigmas):
        """|Tr rho (log rho - log sigma)| for every (sigma, rho) pair (``entanglement.py:330-372``):
        shape (n_rhos,) for a single sigma, (n_sigmas, n_rhos) for a batch of them."""
        rhos, log_rhos, log_sigmas = (np.asarray(a) for a in (rhos, log_rhos, log_sigmas))
        single = log_sigmas.ndim == 2
        ls = log_sigmas[None] if single else log_sigmas
        # Tr(A B) = sum_ij A_ij B_ji
        out = np.abs(np.einsum("rij,srji->sr", rhos, log_rhos[None] - ls[:, None]))
        return out[0] if single else out

    @classmethod
    def relative_entropy(cls, model: Model, n_samples: int, n_sigmas: int, random_key=None,
                         scale: bool = False, **kwargs: Any) -> float:
        """Relative entropy of entanglement against ``n_sigmas`` random separable states (an upper
        bound: the nearest separable state is not searched for), normalised by the GHZ state's
        (``entanglement.py:222-304``)."""
        from .utils import safe_random_split

        if scale:
            n_samples = (2**model.n_qubits) * n_samples
            n_sigmas = (2**model.n_qubits) * n_sigmas
        if random_key is None:
            random_key = model.random_key
        log_sigmas = sample_random_separable_states(model.n_qubits, n_samples=int(n_sigmas),
                                                    random_key=random_key, take_log=True)
        random_key, _ = safe_random_split(random_key)
        if n_samples is not None and n_samples > 0:
            model.initialize_params(random_key, repeat=int(n_samples))
        elif np.asarray(model.params).ndim <= 2:
            model.params = np.asarray(model.params).reshape(1, *np.asarray(model.params).shape)
        rhos, log_rhos = cls._compute_log_density(model, **kwargs)
        rel = cls._compute_rel_entropies(rhos, log_rhos, log_sigmas)          # (n_sigmas, n_rhos)
        ghz = Model(model.n_qubits, 1, "GHZ", data_reupload=False)
        rho_g, log_rho_g = cls._compute_log_density(ghz, **kwargs)
        rel_g = cls._compute_rel_entropies(rho_g, log_rho_g, log_sigmas)      # (n_sigmas, 1)
        capability = (rel / rel_g).min(axis=0)                                 # nearest sampled sigma
        log.debug("Variance of measure: %s", float(capability.var()))
        return float(capability.mean())

    @classmethod
    def _compute_entanglement_of_formation(cls, rhos, n_qubits: int, always_decompose: bool):
        """Eigen-decompose every density matrix and weight the Meyer-Wallach measure of the
        eigenvectors by the eigenvalues (``entanglement.py:438-468``)."""
        rhos = np.asarray(js._host(rhos)).astype(np.complex128)
        evals, evecs = np.linalg.eigh(rhos)
        if not always_decompose and np.isclose(evals, 1.0, atol=1e-5).any(axis=-1).all():
            return np.asarray(cls._compute_meyer_wallach_meas(rhos, n_qubits))
        dim = 2**n_qubits
        states = np.swapaxes(evecs, -1, -2).reshape(-1, dim)                   # eigenvector k = row
        measures = cls._compute_meyer_wallach_meas(states.astype(np.complex64), n_qubits)
        measures = np.asarray(js._host(measures), dtype=np.float64).reshape(-1, dim)
        return np.einsum("si,si->s", measures, np.clip(evals, 0.0, None))

    @classmethod
    def entanglement_of_formation(cls, model: Model, n_samples: int, random_key=None,
                                  scale: bool = False, always_decompose: bool = False,
                                  **kwargs: Any) -> float:
        """Entanglement of formation of (possibly mixed) output states for the eigen-decomposition
        into pure states; equals Meyer-Wallach for pure states (``entanglement.py:375-435``)."""
        if scale:
            n_samples = (2**model.n_qubits) * n_samples
        if n_samples is not None and n_samples > 0:
            model.initialize_params(random_key, repeat=int(n_samples))
        elif np.asarray(model.params).ndim <= 2:
            model.params = np.asarray(model.params).reshape(1, *np.asarray(model.params).shape)
        kwargs.setdefault("inputs", None)
        kwargs.pop("execution_type", None)
        rhos = np.asarray(js._host(model(execution_type="density", **kwargs)))
        dim = 2**model.n_qubits
        ent = cls._compute_entanglement_of_formation(rhos.reshape(-1, dim, dim), model.n_qubits,
                                                     always_decompose)
        return float(np.mean(ent))


def logm_v(A):
    """Matrix logarithm of one matrix or of every matrix of a batch (``math.py:7-28``)."""
    import warnings

    from scipy.linalg import logm

    A = np.asarray(A)
    if A.ndim not in (2, 3):
        raise NotImplementedError("Unsupported shape of input matrix")
    with warnings.catch_warnings():
        # pure states are singular: SciPy reports "logm result may be inaccurate" for every one
        # of them (the reference leaves the same warnings unhandled, math.py:19)
        warnings.simplefilter("ignore", RuntimeWarning)
        return logm(A) if A.ndim == 2 else np.stack([logm(a) for a in A])


def sample_random_separable_states(n_qubits: int, n_samples: int, random_key, take_log: bool = False):
    """Density matrices (n_samples, 2^n, 2^n) of random product states: one ``No_Entangling``
    layer with random angles (``entanglement.py:687-715``)."""
    model = Model(n_qubits, 1, "No_Entangling", data_reupload=False)
    model.initialize_params(random_key, repeat=int(n_samples))
    sigmas = np.asarray(js._host(model(execution_type="density", inputs=None)))
    sigmas = sigmas.reshape(-1, 2**n_qubits, 2**n_qubits)
    if take_log:
        sigmas = logm_v(sigmas) / np.log(2.0)
    return sigmas
