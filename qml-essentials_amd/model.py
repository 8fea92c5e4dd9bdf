"""Data-reuploading ``Model`` on the MI355X statevector engine.

API mirror of ``qml_essentials/model.py`` for the noise-free unitary path:
constructor (``:26-210``), batching semantics (``_assimilate_batch`` ``:1414-1483``),
gate order (``_variational`` ``:818-963``, ``_iec`` ``:746-816``), observables
(``_build_obs`` ``:965-998``), result shaping (``_forward`` ``:1572-1737``) and
``initialize_params`` (``:631-722``).  Results are NumPy arrays (the reference
returns ``jnp`` arrays); the 2^n arithmetic runs in ``libqmle_sv`` via
:class:`script.Script`.  Out of scope: noise, pulses, drawing, ``exact_spectrum``.
"""
from __future__ import annotations

import logging
import warnings
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import numpy as np

from . import jaqsi as js
from . import operations as op
from .tape import recording
from .ansaetze import Ansaetze, Circuit, Encoding
from .batching import to_numpy
from .gates import Gates
from ._native import to_host as N_to_host
from .utils import (PRNGKey, as_key, device_sampling, safe_random_split, uniform,
                    uniform_device)

log = logging.getLogger(__name__)

_NOISE_KEYS = ("BitFlip", "PhaseFlip", "Depolarizing", "MultiQubitDepolarizing",
               "AmplitudeDamping", "PhaseDamping", "GateError", "ThermalRelaxation",
               "StatePreparation", "Measurement")


class Model:
    """A parametrised quantum circuit with (re-uploaded) input encoding."""

    def __init__(
        self,
        n_qubits: int,
        n_layers: int,
        circuit_type: Union[str, Circuit] = "No_Ansatz",
        data_reupload: Union[bool, List[List[bool]], List[List[List[bool]]]] = True,
        state_preparation: Union[str, Callable, List[Union[str, Callable]], None] = None,
        encoding: Union[Encoding, str, Callable, List[Union[str, Callable]]] = Gates.RX,
        trainable_frequencies: bool = False,
        initialization: str = "random",
        initialization_domain: List[float] = [0, 2 * np.pi],
        output_qubit: Union[List[int], int] = -1,
        shots: Optional[int] = None,
        random_seed: int = 1000,
        remove_zero_encoding: bool = True,
        repeat_batch_axis: List[bool] = [True, True, True],
        pulse_shape: str = "gaussian",
        x64: Optional[bool] = None,
    ) -> None:
        # x64 (extension): run this model on the complex128 engine regardless of the global
        # ``utils.enable_x64`` switch (None = follow it) -- the reference's ``jax_enable_x64`` mode
        self.x64 = x64
        self._params, self._params_dev = None, None
        self._fast_calls = {}  # fingerprint -> compiled device call (_forward_device)
        self.n_qubits = n_qubits
        self.output_qubit = output_qubit
        self.n_layers = n_layers
        self.noise_params = None
        self.shots = shots
        self.remove_zero_encoding = remove_zero_encoding
        self.trainable_frequencies = trainable_frequencies
        self.execution_type = "expval"
        self.repeat_batch_axis = list(repeat_batch_axis)
        self.gate_mode = "unitary"

        try:
            self._sp = Gates.parse_gates(state_preparation, Gates)
        except ValueError as e:
            raise ValueError(f"Error parsing encodings: {e}")

        self._enc = encoding if isinstance(encoding, Encoding) else Encoding("hamming", encoding)
        if self._enc.is_golomb:
            self._enc._n_qubits = n_qubits
        self.n_input_feat = len(self._enc)

        # trainable frequencies start at one (arXiv:2309.03279v2), model.py:150
        self.enc_params = np.ones((n_layers, n_qubits, self.n_input_feat), dtype=np.float32)
        self._zero_inputs = False
        self.data_reupload = data_reupload  # also sets degree / frequencies / has_dru

        impl_layers = n_layers + 1 if self.has_dru else n_layers  # Schuld et al.: L+1 blocks

        if isinstance(circuit_type, str):
            self.pqc = getattr(Ansaetze, circuit_type or "No_Ansatz")()
        else:
            self.pqc = circuit_type()
        self._params_shape = (impl_layers, self.pqc.n_params_per_layer(n_qubits))
        self._pulse_params_shape = (impl_layers, 0)

        self._batch_shape = None
        self._inialization_strategy = initialization
        self._initialization_domain = initialization_domain
        self.random_key = self.initialize_params(PRNGKey(random_seed))
        self.pulse_params = np.ones((1, *self._pulse_params_shape), dtype=np.float32)

        self.script = js.Script(f=self._variational, n_qubits=n_qubits)

    # ------------------------------------------------------------------ properties
    @property
    def noise_params(self):
        return self._noise_params

    @noise_params.setter
    def noise_params(self, kvs) -> None:
        if kvs is not None and all(v == 0.0 for v in kvs.values()):
            kvs = None
        if kvs is not None:
            for k in kvs:
                if k not in _NOISE_KEYS:
                    warnings.warn(f"Noise type {k} is not supported by this package", UserWarning)
            for k in _NOISE_KEYS:
                kvs.setdefault(k, None if k == "ThermalRelaxation" else 0.0)
        self._noise_params = kvs

    @property
    def output_qubit(self):
        return self._output_qubit

    @output_qubit.setter
    def output_qubit(self, value) -> None:
        if isinstance(value, list):
            assert len(value) <= self.n_qubits, (
                f"Size of output_qubit {len(value)} cannot be larger than number of qubits "
                f"{self.n_qubits}."
            )
        elif isinstance(value, (int, np.integer)):
            if value == -1:
                value = list(range(self.n_qubits))
            else:
                assert value < self.n_qubits, (
                    f"Output qubit {value} cannot be larger than {self.n_qubits}."
                )
                value = [int(value)]
        self._output_qubit = value
        # the reference derives the result shape in the execution_type setter only
        # (model.py:343-365), so a later change of output_qubit leaves it stale there; refresh it
        if getattr(self, "_execution_type", None) is not None:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                self.execution_type = self._execution_type

    @property
    def execution_type(self) -> str:
        return self._execution_type

    @execution_type.setter
    def execution_type(self, value: str) -> None:
        k = len(self.output_qubit)
        # "state" ignores output_qubit (warning below), so its shape is the full register; the
        # reference sizes it by output_qubit (model.py:365) and then fails in its own reshape
        shapes = {"density": (2**k, 2**k), "expval": (k,), "probs": (2,) * k,
                  "state": (2**self.n_qubits,)}
        if value not in shapes:
            raise ValueError(f"Invalid execution type: {value}.")
        self._result_shape = shapes[value]
        if value == "state" and not self.all_qubit_measurement:
            warnings.warn(
                f"{value} measurement does ignore output_qubit, which is {self.output_qubit}.",
                UserWarning,
            )
        if value == "probs" and self.shots is None:
            warnings.warn("Setting execution_type to probs without specifying shots.", UserWarning)
        if value == "density" and self.shots is not None:
            raise ValueError("Setting execution_type to density with shots not None.")
        self._execution_type = value

    @property
    def shots(self) -> Optional[int]:
        return self._shots

    @shots.setter
    def shots(self, value: Optional[int]) -> None:
        if type(value) is int and value <= 0:
            value = None
        self._shots = value

    @property
    def params(self) -> np.ndarray:
        """Host view of the parameters.  When they were drawn (or set) on the GPU the host
        mirror is materialised on first access -- like a ``jax`` device array, which the
        reference's ``params`` are (``model.py:687-693``), turns into NumPy only when asked."""
        if self._params is None:
            # a read-only snapshot: the live storage is the device tensor, so an in-place edit of
            # this mirror could not take effect -- it raises instead of being silently ignored
            # (assign ``model.params = new`` to change them).  Not cached: the device tensor may
            # be updated in place (an optimizer step) and a cached mirror would go stale.
            mirror = self._params_dev.detach().cpu().numpy().astype(np.float64)
            mirror.setflags(write=False)
            return mirror
        return self._params

    @params.setter
    def params(self, value) -> None:
        if self._is_cuda(value):  # device-resident: the host mirror is made on demand.  The
            # tensor is kept BY REFERENCE (no copy: a training loop's in-place updates are meant
            # to be seen by the next call); ``model.params`` reads it back afresh each time.
            value = value.detach()
            if value.dim() == 2:
                value = value.unsqueeze(0)
            self._params_dev, self._params = value, None
            return
        value = to_numpy(value)
        if len(value.shape) == 2:
            value = value.reshape(1, *value.shape)
        self._params, self._params_dev = value, None

    def device_params(self):
        """The float32 CUDA tensor behind ``params`` when they live on the GPU, else None."""
        return self._params_dev

    @property
    def enc_params(self) -> np.ndarray:
        return self._enc_params

    @enc_params.setter
    def enc_params(self, value) -> None:
        self._enc_params = to_numpy(value)

    @property
    def pulse_params(self) -> np.ndarray:
        return self._pulse_params

    @pulse_params.setter
    def pulse_params(self, value) -> None:
        self._pulse_params = value

    @property
    def data_reupload(self) -> np.ndarray:
        return self._data_reupload

    @data_reupload.setter
    def data_reupload(self, value) -> None:
        """Boolean mask (n_layers, n_qubits, n_input_feat); also derives the naive
        spectrum (``model.py:466-512``).  ``False`` keeps exactly one encoding gate
        (layer 0 / qubit 0)."""
        full = (self.n_layers, self.n_qubits, self.n_input_feat)
        if isinstance(value, (bool, np.bool_)):
            mask = np.ones(full) if value else np.zeros(full)
            if not value:
                mask[0][0] = 1
        else:
            mask = np.asarray(value)
            if mask.ndim == 2:
                assert mask.shape == full[:2], (
                    f"Data reuploading array has wrong shape. Expected {full[:2]} or {full}, "
                    f"got {mask.shape}."
                )
                mask = np.repeat(mask[:, :, None], self.n_input_feat, axis=2)
            assert mask.shape == full, (
                f"Data reuploading array has wrong shape. Expected {full}, got {mask.shape}."
            )
        self._data_reupload = mask.astype(bool)
        uses = [int(np.count_nonzero(self._data_reupload[..., i])) for i in range(self.n_input_feat)]
        self.degree = tuple(self._enc.get_n_freqs(u) for u in uses)
        self.frequencies = tuple(self._enc.get_spectrum(u) for u in uses)
        self._has_dru = bool(max(int(np.max(f)) for f in self.frequencies) > 1)

    @property
    def degree(self) -> Tuple:
        return self._degree

    @degree.setter
    def degree(self, value: Tuple) -> None:
        self._degree = value

    @property
    def frequencies(self) -> Tuple:
        return self._frequencies

    @frequencies.setter
    def frequencies(self, value: Tuple) -> None:
        self._frequencies = value

    @property
    def has_dru(self) -> bool:
        return self._has_dru

    @property
    def all_qubit_measurement(self) -> bool:
        return self.output_qubit == list(range(self.n_qubits))

    @property
    def batch_shape(self) -> Tuple[int, ...]:
        """(B_I, B_P, B_R); (1, 1, 1) before the first call."""
        return (1, 1, 1) if self._batch_shape is None else self._batch_shape

    @property
    def eff_batch_shape(self) -> np.ndarray:
        shape = np.array(self.batch_shape) * self.repeat_batch_axis
        return shape[shape != 0]

    # ------------------------------------------------------------------ parameters
    def initialize_params(self, random_key=None, repeat: int = 1,
                          initialization: Optional[str] = None,
                          initialization_domain: Optional[List[float]] = None):
        """(Re-)draw ``self.params`` with shape ``(repeat, layers, params_per_layer)``;
        returns the advanced key (``model.py:631-722``)."""
        shape = (repeat, *self._params_shape)
        strategy = initialization or self._inialization_strategy
        lo, hi = initialization_domain or self._initialization_domain
        random_key, sub = safe_random_split(
            as_key(random_key) if random_key is not None else self.random_key
        )

        def draw():
            # large draws stay where the GPU sampler wrote them (the analysis loops feed them
            # straight back into the engine); `params` materialises a host mirror on demand
            if device_sampling(int(np.prod(shape))):
                return uniform_device(sub, shape, minval=lo, maxval=hi)
            return uniform(sub, shape, minval=lo, maxval=hi)

        def pin_controlled(params, value: float):
            idx = self.pqc.get_control_indices(self.n_qubits)
            if idx is None:
                warnings.warn(
                    f"Specified {strategy} but circuit does not contain controlled rotation "
                    "gates. Parameters are intialized randomly.",
                    UserWarning,
                )
                return params
            if not self._is_cuda(params):
                params = np.array(params)
            if len(idx) == 3 and None in idx:
                params[:, :, idx[0]:idx[1]:idx[2]] = value
            else:
                params[:, :, idx] = value
            return params

        if strategy == "random":
            self.params = draw()
        elif strategy == "zeros":
            self.params = np.zeros(shape, dtype=np.float32)
        elif strategy == "pi":
            self.params = np.full(shape, np.pi, dtype=np.float32)
        elif strategy == "zero-controlled":
            self.params = pin_controlled(draw(), 0.0)
        elif strategy == "pi-controlled":
            self.params = pin_controlled(draw(), np.pi)
        else:
            raise Exception("Invalid initialization method")
        return random_key

    def transform_input(self, inputs, enc_params):
        """Scale the input by the encoding weight (arXiv:2309.03279v2)."""
        return inputs * enc_params

    # ------------------------------------------------------------------ circuit
    def _iec(self, inputs, data_reupload: np.ndarray, enc: Encoding, enc_params,
             noise_params=None, random_key=None) -> None:
        """Input-encoding layer (``model.py:746-816``)."""
        if self.remove_zero_encoding and self._zero_inputs and self.batch_shape[0] == 1:
            return
        if enc.is_golomb:
            if data_reupload[:, 0].any():
                random_key, sub_key = safe_random_split(random_key)
                scale = np.mean(enc_params[:, 0])
                enc[0](self.transform_input(inputs[..., 0], scale),
                       wires=list(range(self.n_qubits)), noise_params=noise_params,
                       random_key=sub_key)
            return
        for q in range(self.n_qubits):
            for idx in range(inputs.shape[-1]):
                if data_reupload[q, idx]:
                    random_key, sub_key = safe_random_split(random_key)
                    enc[idx](self.transform_input(inputs[..., idx], enc_params[q, idx]),
                             wires=q, noise_params=noise_params, random_key=sub_key)

    def _variational(self, params, inputs, pulse_params=None, random_key=None, enc_params=None,
                     gate_mode: str = "unitary", noise_params=None) -> None:
        """Record the whole circuit: state prep; per layer ansatz then encoding; final
        ansatz layer when re-uploading is active (``model.py:913-959``)."""
        if len(params.shape) > 2 and params.shape[0] == 1:
            params = params[0]
        if len(inputs.shape) > 1 and inputs.shape[0] == 1:
            inputs = inputs[0]
        if enc_params is None:
            if self.trainable_frequencies:
                warnings.warn(
                    "Explicit call to `_circuit` or `_variational` detected: "
                    "`enc_params` is None, using `self.enc_params` instead.",
                    RuntimeWarning,
                )
            enc_params = self.enc_params
        if noise_params is None and self.noise_params is not None:
            warnings.warn(
                "Explicit call to `_circuit` or `_variational` detected: "
                "`noise_params` is None, using `self.noise_params` instead.",
                RuntimeWarning,
            )
            noise_params = self.noise_params
        if noise_params is not None:
            if random_key is None:
                warnings.warn(
                    "Explicit call to `_circuit` or `_variational` detected: "
                    "`random_key` is None, using the model's key instead.",
                    RuntimeWarning,
                )
                random_key = self.random_key
            self._apply_state_prep_noise(noise_params=noise_params)

        for q in range(self.n_qubits):
            for prep in self._sp:
                random_key, sub_key = safe_random_split(random_key)
                prep(wires=q, noise_params=noise_params, random_key=sub_key, gate_mode=gate_mode)
        for layer in range(self.n_layers):
            random_key, sub_key = safe_random_split(random_key)
            self.pqc(params[layer], self.n_qubits, noise_params=noise_params,
                     random_key=sub_key, gate_mode=gate_mode)
            random_key, sub_key = safe_random_split(random_key)
            self._iec(inputs, data_reupload=self.data_reupload[layer], enc=self._enc,
                      enc_params=enc_params[layer], noise_params=noise_params,
                      random_key=sub_key)
        if self.has_dru:
            random_key, sub_key = safe_random_split(random_key)
            self.pqc(params[self.n_layers], self.n_qubits, noise_params=noise_params,
                     random_key=sub_key, gate_mode=gate_mode)
        if noise_params is not None:
            self._apply_general_noise(noise_params=noise_params)

    # ------------------------------------------------------------------ noise
    def _apply_state_prep_noise(self, noise_params) -> None:
        """BitFlip(p = "StatePreparation") on every qubit (``model.py:1000-1020``)."""
        p = noise_params.get("StatePreparation", 0.0)
        if p > 0:
            for q in range(self.n_qubits):
                op.BitFlip(p, wires=q)

    def _apply_general_noise(self, noise_params) -> None:
        """End-of-circuit decoherence channels per qubit: AmplitudeDamping, PhaseDamping,
        Measurement (BitFlip), ThermalRelaxation with gate time = depth * t_factor
        (``model.py:1022-1064``)."""
        amp = noise_params.get("AmplitudeDamping", 0.0)
        phase = noise_params.get("PhaseDamping", 0.0)
        thermal = noise_params.get("ThermalRelaxation", 0.0)
        meas = noise_params.get("Measurement", 0.0)
        for q in range(self.n_qubits):
            if amp > 0:
                op.AmplitudeDamping(amp, wires=q)
            if phase > 0:
                op.PhaseDamping(phase, wires=q)
            if meas > 0:
                op.BitFlip(meas, wires=q)
            if isinstance(thermal, dict):
                tg = self._get_circuit_depth() * thermal["t_factor"]
                op.ThermalRelaxationError(1.0, thermal["t1"], thermal["t2"], tg, q)

    def _get_circuit_depth(self, inputs=None) -> int:
        """Critical-path length of the noise-free circuit: every gate is scheduled at the
        earliest step where all its wires are free (``model.py:1066-1118``); cached."""
        if hasattr(self, "_cached_circuit_depth"):
            return self._cached_circuit_depth
        inputs = self._inputs_validation(inputs)
        saved, self._noise_params = self._noise_params, None
        try:
            with recording() as tape:
                self._variational(self.params[0] if self.params.ndim == 3 else self.params,
                                  inputs[0] if inputs.ndim == 2 else inputs, noise_params=None)
        finally:
            self._noise_params = saved
        busy, depth = {}, 0
        for gate in tape:
            if isinstance(gate, op.KrausChannel):
                continue
            end = max((busy.get(w, 0) for w in gate.wires), default=0) + 1
            for w in gate.wires:
                busy[w] = end
            depth = max(depth, end)
        self._cached_circuit_depth = depth
        return depth

    def _requires_density(self) -> bool:
        """Density request, or any incoherent channel with non-zero strength
        (``model.py:1485-1510``); GateError alone stays on the pure path."""
        if self.execution_type == "density":
            return True
        if self.noise_params is None:
            return False
        for k, v in self.noise_params.items():
            if k == "GateError":
                continue
            if isinstance(v, dict) or (v is not None and v > 0):
                return True
        return False

    def _build_obs(self) -> Tuple[str, List[op.Operation]]:
        et = self.execution_type
        if et in ("density", "state", "probs"):
            return et, []
        if et == "expval":
            key = tuple(int(q) if isinstance(q, (int, np.integer)) else tuple(int(w) for w in q)
                        for q in self.output_qubit)
            cached = getattr(self, "_obs_cache", None)
            if cached is not None and cached[0] == key:  # rebuilt only when output_qubit changes
                return "expval", cached[1]
            obs = []
            for spec in self.output_qubit:
                if isinstance(spec, (int, np.integer)):
                    obs.append(op.PauliZ(wires=int(spec), record=False))
                else:
                    obs.append(js.build_parity_observable(list(spec)))
            self._obs_cache = (key, obs)
            return "expval", obs
        raise ValueError(f"Invalid execution_type: {et}.")

    # ------------------------------------------------------------------ validation
    def _params_validation(self, params) -> np.ndarray:
        if params is None:
            return self.params
        params = to_numpy(params)
        if len(params.shape) == 2:
            params = np.expand_dims(params, axis=0)
        self.params = params
        return params

    def _enc_params_validation(self, enc_params) -> np.ndarray:
        if enc_params is None:
            enc_params = self.enc_params
        else:
            enc_params = to_numpy(enc_params)
            self.enc_params = enc_params
        if len(enc_params.shape) == 1 and self.n_input_feat == 1:
            enc_params = enc_params.reshape(-1, 1)
        elif len(enc_params.shape) == 1 and self.n_input_feat > 1:
            raise ValueError(
                f"Input dimension {self.n_input_feat} >1 but `enc_params` has shape "
                f"{enc_params.shape}"
            )
        return enc_params

    def _inputs_validation(self, inputs) -> np.ndarray:
        """-> ``(batch, n_input_feat)`` float array; flags all-zero inputs."""
        self._zero_inputs = False
        if isinstance(inputs, list):
            inputs = np.array(np.stack([to_numpy(i) for i in inputs]))
        elif isinstance(inputs, (float, int)):
            inputs = np.array([inputs])
        elif inputs is None:
            inputs = np.array([[0] * self.n_input_feat])
        inputs = np.asarray(to_numpy(inputs))
        if not inputs.any():
            self._zero_inputs = True
        if len(inputs.shape) <= 1:
            if self.n_input_feat == 1:
                inputs = inputs.reshape(-1, 1)
            elif inputs.shape and inputs.shape[0] == self.n_input_feat:
                inputs = inputs.reshape(1, -1)
            else:
                inputs = inputs.reshape(-1, 1).repeat(self.n_input_feat, axis=1)
                warnings.warn(
                    f"Expected {self.n_input_feat} inputs, but {inputs.shape[0]} "
                    "was provided, replicating input for all input features.",
                    UserWarning,
                )
        elif inputs.shape[1] != self.n_input_feat:
            raise ValueError(
                f"Wrong number of inputs provided. Expected {self.n_input_feat} "
                f"inputs, but input has shape {inputs.shape}."
            )
        return inputs

    def _assimilate_batch(self, inputs: np.ndarray, params: np.ndarray):
        """Cartesian batch ``B = B_I * B_P`` with inputs slowest (``model.py:1414-1483``);
        axes whose ``repeat_batch_axis`` flag is off are zipped instead of crossed."""
        B_I = inputs.shape[0]
        B_P = 1 if 0 in params.shape else params.shape[0]
        self._batch_shape = (B_I, B_P, 1)
        B = int(np.prod(self.eff_batch_shape))
        rep_i, rep_p, _ = self.repeat_batch_axis
        if B_I > 1 and rep_i:
            x = inputs[:, None, ...]
            if rep_p:
                x = np.repeat(x, B_P, axis=1)
            inputs = x.reshape(B, *inputs.shape[1:])
        if B_P > 1 and rep_p:
            p = params[None, ...]
            if rep_i:
                p = np.repeat(p, B_I, axis=0)
            params = p.reshape(B, *params.shape[1:])
        return inputs, params

    def gradient(self, params=None, inputs=None, enc_params=None, wrt: str = "params",
                 force_mean: bool = False, data_reupload=None,
                 method: str = "parameter-shift", cotangent=None) -> np.ndarray:
        """d<Z_q>/d``wrt`` for every output qubit by the parameter-shift rule
        (:meth:`script.Script.gradient`); ``wrt`` in {"params", "inputs", "enc_params"}.

        The reference differentiates ``model(...)`` with ``jax.grad``
        (``tests/test_model.py:1097-1145``, ``docs/training.md``); without an autodiff
        framework this method returns the Jacobian of the expectation values instead, from
        which any cost gradient follows by the chain rule on the host.  Shape:
        ``(*eff_batch_shape, n_outputs, *shape_of(wrt))`` (batch axes of size 1 dropped;
        ``force_mean`` averages over the outputs like ``__call__``).

        ``method="adjoint"`` uses adjoint differentiation (:mod:`adjoint`): one backward sweep
        per output, or a single sweep for ``force_mean`` / an explicit ``cotangent`` of shape
        ``(B, n_outputs)`` -- then the result is the vector-Jacobian product
        ``sum_k cotangent[b, k] d<out_k>_b / d wrt`` of shape ``(*eff_batch_shape, *shape_of(wrt))``,
        i.e. the gradient of any cost whose derivative with respect to the outputs is
        ``cotangent``.  Same numbers as the parameter-shift rule at O(gates) instead of
        O(gates x angles) cost.
        """
        from .utils import x64_enabled, x64_scope

        if self.x64 is not None and bool(self.x64) != x64_enabled():  # Model(x64=...) scopes the mode
            with x64_scope(self.x64):
                return self.gradient(params=params, inputs=inputs, enc_params=enc_params, wrt=wrt,
                                     force_mean=force_mean, data_reupload=data_reupload,
                                     method=method, cotangent=cotangent)
        if method == "auto":  # measured (profiles/r01_gradients.md): from 14 qubits the fused
            # backward sweep beats the batched parameter shift 5-40x; below, psi and lambda live
            # in LDS and the single-launch sweep wins once there is a batch to spread over the CUs
            n_in = 1 if inputs is None else int(np.asarray(to_numpy(inputs)).reshape(-1, self.n_input_feat).shape[0])
            ok = self.noise_params is None and (force_mean or cotangent is not None)
            method = "adjoint" if ok and (self.n_qubits >= 14 or n_in >= 8) else "parameter-shift"
        if method not in ("parameter-shift", "adjoint"):
            raise ValueError(
                f"method must be 'parameter-shift', 'adjoint' or 'auto', got {method!r}")
        if cotangent is not None and method != "adjoint":
            raise ValueError("cotangent needs method='adjoint'")
        if wrt not in ("params", "inputs", "enc_params"):
            raise ValueError(f"wrt must be 'params', 'inputs' or 'enc_params', got {wrt!r}")
        if data_reupload is not None:  # call-time override, as in __call__ (model.py:1633-1634)
            self.data_reupload = data_reupload
        self.execution_type = "expval"
        params = self._params_validation(params)
        inputs = self._inputs_validation(inputs)
        enc_params = self._enc_params_validation(enc_params)
        inputs, params = self._assimilate_batch(inputs, params)
        if self.remove_zero_encoding and self._zero_inputs and self.batch_shape[0] == 1 \
                and wrt != "params":
            raise ValueError("inputs are all zero and remove_zero_encoding=True: the encoding "
                             "gates are not on the tape; pass remove_zero_encoding=False")
        _, obs = self._build_obs()
        B = int(np.prod(self.eff_batch_shape))
        args = (params, inputs, None, None, np.asarray(enc_params, dtype=np.float64))
        in_axes = (0 if self.batch_shape[1] > 1 else None, 0 if self.batch_shape[0] > 1 else None,
                   None, None, None)
        argnum = {"params": 0, "inputs": 1, "enc_params": 4}[wrt]
        kwargs = dict(noise_params=self.noise_params, gate_mode="unitary")
        if method == "adjoint":  # (x64 mode: Script.vjp runs the complex128 sweep, qmle_adjoint_gradient_f64)
            n_out = len(obs)
            ax = in_axes if B > 1 else None
            if cotangent is not None or force_mean:
                w = (np.full((B, n_out), 1.0 / n_out) if cotangent is None
                     else np.asarray(cotangent, dtype=np.float64).reshape(B, n_out))
                (g,) = self.script.vjp(obs, w, args=args, kwargs=kwargs, in_axes=ax,
                                       argnums=(argnum,))
                jac = (g if B > 1 else g[None])[:, None]       # (B, 1, *leaf)
                force_mean = True                              # the output axis is already reduced
            else:
                # the Jacobian: one one-hot cotangent per output, all of them in ONE trace and ONE sweep over
                # n_out * B states (round 5; before: n_out calls, each re-recording the tape)
                w = np.zeros((n_out, B, n_out))
                w[np.arange(n_out), :, np.arange(n_out)] = 1.0
                (g,) = self.script.vjp(obs, w, args=args, kwargs=kwargs, in_axes=ax, argnums=(argnum,))
                jac = np.moveaxis(g if B > 1 else g[:, None], 0, 1)   # (B, n_out, *leaf)
        else:
            (jac,) = self.script.gradient(obs, args=args, kwargs=kwargs,
                                          in_axes=in_axes if B > 1 else None, argnums=(argnum,))
            if B == 1:
                jac = jac[None]
        leaf = jac.shape[2:]
        if wrt in ("params", "inputs") and in_axes[argnum] is None and leaf and leaf[0] == 1:
            jac = jac.reshape(jac.shape[:2] + leaf[1:])  # drop the dummy batch axis of the arg
        jac = jac.reshape(*self.eff_batch_shape, *jac.shape[1:])
        jac = jac.reshape([d for i, d in enumerate(jac.shape)
                           if not (i < len(self.eff_batch_shape) and d == 1)])
        if force_mean:
            jac = jac.mean(axis=len([d for d in self.eff_batch_shape if d != 1]))
        return jac

    def record_tape(self, params=None, inputs=None, enc_params=None):
        """Validate + batch the arguments exactly like ``__call__`` and return
        ``(tape, batch)`` without executing (host only; used by tests and tools)."""
        from .batching import Batched
        from .tape import recording

        params = self._params_validation(params)
        inputs = self._inputs_validation(inputs)
        enc_params = self._enc_params_validation(enc_params)
        inputs, params = self._assimilate_batch(inputs, params)
        B = int(np.prod(self.eff_batch_shape))
        if B > 1:
            if self.batch_shape[1] > 1:
                params = Batched(params)
            if self.batch_shape[0] > 1:
                inputs = Batched(inputs)
        with recording() as tape:
            self._variational(params, inputs, enc_params=enc_params)
        return tape, B

    # ------------------------------------------------------------------ execution
    def __call__(self, params=None, inputs=None, pulse_params=None, enc_params=None,
                 data_reupload=None, noise_params=None, execution_type: Optional[str] = None,
                 force_mean: bool = False, gate_mode: str = "unitary") -> np.ndarray:
        return self._forward(params=params, inputs=inputs, pulse_params=pulse_params,
                             enc_params=enc_params, data_reupload=data_reupload,
                             noise_params=noise_params, execution_type=execution_type,
                             force_mean=force_mean, gate_mode=gate_mode)

    # ------------------------------------------------------------------ device-resident path
    @staticmethod
    def _is_cuda(x) -> bool:
        return hasattr(x, "is_cuda") and bool(x.is_cuda)

    def _call_fingerprint(self, params, inputs, enc_params, et):
        """Cheap key of everything that decides the compiled call of :meth:`_forward_device` when
        the per-sample arguments are CUDA tensors (shapes only) and the rest are the model's own
        small host arrays (their bytes): the sampling loops call with the same structure thousands
        of times, and re-deriving it (validation, blake2b over float64 copies, batch bookkeeping)
        cost more host time than the GPU needed for the circuits.  None: take the general path."""
        if enc_params is not None:
            return None
        if self._is_cuda(params):
            if params.dim() not in (2, 3):
                return None
            pk = ("d", tuple(params.shape[-2:]), params.dim() == 3 and int(params.shape[0]) > 1)
        elif params is None and self._params is not None and self._params.shape[0] == 1:
            pk = ("h", hash(self._params.tobytes()), self._params.shape)
        else:
            return None
        if self._is_cuda(inputs):
            if inputs.dim() != 2 or inputs.shape[1] != self.n_input_feat:
                return None
            xk = ("d", int(inputs.shape[0]) > 1)
        elif inputs is None:
            xk = ("z",)
        else:
            return None
        return (et, pk, xk, hash(self._enc_params.tobytes()), hash(self._data_reupload.tobytes()),
                self.remove_zero_encoding, tuple(self.repeat_batch_axis), self._oq_key(), self._noise_key())

    def _noise_key(self):
        """Hashable view of the active noise parameters (they are constants of a compiled call)."""
        if self.noise_params is None:
            return None
        return tuple(sorted((k, tuple(sorted(v.items())) if isinstance(v, dict) else v)
                            for k, v in self.noise_params.items()))

    def prepared_state_call(self, n_param_sets: int):
        """The compiled device call that turns a CUDA tensor of ``n_param_sets`` parameter sets
        (inputs None) into statevectors, looked up BEFORE the parameters exist: the sampling loops
        resolve it first, launch the sampler, and then have nothing left to do on the host between
        the sampler's launch and the circuit's.  -> ``(cc, divs, mods, B)`` or None (no such call
        has been compiled yet, noise / shots / complex128 mode: take the general path)."""
        from .utils import x64_enabled

        if self.noise_params is not None or self.shots is not None or self.gate_mode != "unitary":
            return None
        if (x64_enabled() if self.x64 is None else self.x64):
            return None
        fp = ("state", ("d", tuple(self._params_shape), n_param_sets > 1), ("z",),
              hash(self._enc_params.tobytes()), hash(self._data_reupload.tobytes()),
              self.remove_zero_encoding, tuple(self.repeat_batch_axis), self._oq_key(), None)
        rec = self._fast_calls.get(fp)
        if rec is None:
            return None
        cc, p_leaf, x_leaf, _b_i, _b_p, _cross, zero_inputs = rec
        if not p_leaf or x_leaf:
            return None
        if self._execution_type != "state":  # through the setter: it derives _result_shape
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                self.execution_type = "state"
        self._zero_inputs = zero_inputs
        self._batch_shape = (1, n_param_sets, 1)
        return cc, [1], [n_param_sets if n_param_sets > 1 else 1], n_param_sets

    def _oq_key(self):
        oq = self.output_qubit
        return tuple(oq) if oq and isinstance(oq[0], (int, np.integer)) else repr(oq)

    def _forward_device(self, params, inputs, enc_params, execution_type, force_mean,
                        _want_call: bool = False, raw: bool = False):
        """``__call__`` for CUDA-tensor ``params`` / ``inputs``: nothing per sample happens on
        the host and the result stays on the GPU (a ``torch`` tensor).  Falls back to the host
        path (returns ``NotImplemented``) for partial-wire density / probs or non-affine angles."""
        import torch

        from . import distributed

        if execution_type is not None and execution_type != self._execution_type:
            self.execution_type = execution_type
        et = self.execution_type
        if et in ("density", "probs") and not self.all_qubit_measurement:
            return NotImplemented
        fp = self._call_fingerprint(params, inputs, enc_params, et)
        rec = self._fast_calls.get(fp) if fp is not None else None
        if rec is not None:  # same structure as an earlier call: only the leaves are new
            cc, p_leaf, x_leaf, B_I0, B_P0, cross, zero_inputs = rec
            leaves, divs, mods = [], [], []
            B_P, B_I = B_P0, B_I0
            if p_leaf:
                p = params.to(torch.float32)
                p = (p.unsqueeze(0) if p.dim() == 2 else p).contiguous()
                B_P = 1 if 0 in p.shape else int(p.shape[0])  # (model.py:1444: an ansatz without parameters)
                leaves.append(p); divs.append(1); mods.append(B_P if B_P > 1 else 1)
            if x_leaf:
                x = inputs.to(torch.float32).contiguous()
                B_I = int(x.shape[0])
                leaves.append(x); divs.append(B_P if cross else 1); mods.append(B_I if B_I > 1 else 1)
            if B_I > 1 and B_P > 1 and not cross and B_I != B_P:
                raise ValueError("zipped batch axes must have equal length")
            self._zero_inputs = zero_inputs
            self._batch_shape = (B_I, B_P, 1)
            B = (B_I * B_P if cross else max(B_I, B_P))
        else:
            got = self._device_call(params, inputs, enc_params, et)
            if got is NotImplemented:
                return NotImplemented
            cc, leaves, divs, mods, B, rec = got
            if fp is not None and rec is not None:
                if len(self._fast_calls) > 32:
                    self._fast_calls.clear()
                self._fast_calls[fp] = rec
        if _want_call:  # (Model.vjp_device) hand the compiled call + its leaves to the caller
            return cc, leaves, divs, mods, B
        lo, hi, sharded = distributed.my_block(B, *leaves)
        shot_key = None
        if self.shots is not None and et in ("probs", "expval"):
            # the key schedule of the recorded path (model.py:1670-1675): same draws either way
            self.random_key, sub_key = safe_random_split(self.random_key)
            _, shot_key = safe_random_split(sub_key)
        result = cc.run(leaves, divs, mods, hi - lo, lo, shots=self.shots if shot_key is not None else None,
                        key=shot_key)
        if sharded:
            result = distributed.all_gather_rows(result, B)
        if raw:  # the analysis loops' (B, ...) device tensor, batch axis flat
            return result
        result = result.reshape((*[int(d) for d in self.eff_batch_shape], *self._result_shape))
        result = result.squeeze()
        if et in ("expval", "probs") and force_mean and result.dim() > 0 and self._result_shape[0] > 1:
            result = result.mean(dim=-1)
        return result

    def _device_call(self, params, inputs, enc_params, et):
        """The general derivation of a compiled device call: argument validation, batch
        bookkeeping, cache key over the VALUES of every host argument, tape recording on a miss.
        -> ``(cc, leaves, divs, mods, B, record for the fingerprint cache)`` or NotImplemented."""
        import hashlib

        import torch

        from .script import NotAffine

        enc = self._enc_params_validation(enc_params)
        dev_params, dev_inputs = self._is_cuda(params), self._is_cuda(inputs)
        uploaded = False
        # --- shapes (mirrors _params_validation / _inputs_validation, shapes only) ----------
        if dev_params:
            p = params.to(torch.float32)
            p = p.unsqueeze(0) if p.dim() == 2 else p
            p = p.contiguous()
            B_P = 1 if 0 in p.shape else int(p.shape[0])  # (model.py:1444: an ansatz without parameters)
        else:
            p = self._params_validation(params)
            B_P = 1 if 0 in p.shape else int(p.shape[0])
            if B_P > 1:
                p, dev_params = torch.from_numpy(np.ascontiguousarray(p, dtype=np.float32)).cuda(), True
                uploaded = True
        if dev_inputs:
            x = inputs.to(torch.float32)
            if x.dim() <= 1:
                if self.n_input_feat == 1:
                    x = x.reshape(-1, 1)
                elif x.numel() == self.n_input_feat:
                    x = x.reshape(1, -1)
                else:
                    return NotImplemented
            if x.shape[1] != self.n_input_feat:
                raise ValueError(
                    f"Wrong number of inputs provided. Expected {self.n_input_feat} "
                    f"inputs, but input has shape {tuple(x.shape)}."
                )
            x = x.contiguous()
            B_I = int(x.shape[0])
            self._zero_inputs = False
        else:
            x = self._inputs_validation(inputs)
            B_I = int(x.shape[0])
            if B_I > 1:
                x, dev_inputs = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda(), True
                uploaded = True
        self._batch_shape = (B_I, B_P, 1)
        B = int(np.prod(self.eff_batch_shape))
        rep_i, rep_p, _ = self.repeat_batch_axis
        cross = bool(B_I > 1 and B_P > 1 and rep_i and rep_p)
        if B_I > 1 and B_P > 1 and not cross and B_I != B_P:
            raise ValueError("zipped batch axes must have equal length")
        # --- compiled call ------------------------------------------------------------------
        meas_type, obs = self._build_obs()
        leaf_ids = tuple(k for k, d in ((0, dev_params), (1, dev_inputs)) if d)
        # device leaves enter the cache key by shape only: their values are read back (a
        # device -> host sync) just once, as probe values when the call is first compiled
        host_p = None if dev_params else np.asarray(p)
        host_x = None if dev_inputs else np.asarray(x)
        if host_p is not None and host_p.ndim == 3 and host_p.shape[0] == 1:
            host_p = host_p[0]
        if host_x is not None and host_x.ndim == 2 and host_x.shape[0] == 1:
            host_x = host_x[0]
        shape_p = tuple(p.shape[1:]) if dev_params else tuple(host_p.shape)
        shape_x = tuple(x.shape[1:]) if dev_inputs else tuple(host_x.shape)
        h = hashlib.blake2b(digest_size=12)
        for a in (np.asarray(enc, dtype=np.float64), self.data_reupload, host_p, host_x):
            h.update(b"-" if a is None else np.ascontiguousarray(a).tobytes())
        key = (meas_type, tuple((type(o).__name__, tuple(o.wires)) for o in obs), leaf_ids,
               shape_p, shape_x, self._zero_inputs, B_I == 1,
               self.remove_zero_encoding, h.hexdigest(), self._noise_key())
        noisy = self.noise_params is not None

        def probe_args():
            hp = p[0].detach().cpu().numpy() if dev_params else host_p
            hx = x[0].detach().cpu().numpy() if dev_inputs else host_x
            # (the channels take no randomness; the key only satisfies the noisy gates' signature)
            return (hp, hx, None, PRNGKey(0) if noisy else None, enc)

        try:
            cc = self.script.compiled(key, meas_type, obs, probe_args, leaf_ids,
                                      dict(noise_params=dict(self.noise_params) if noisy else None,
                                           gate_mode="unitary"))
        except NotAffine:
            return NotImplemented
        leaves, divs, mods = [], [], []
        if dev_params:
            leaves.append(p); divs.append(1); mods.append(B_P if B_P > 1 else 1)
        if dev_inputs:
            leaves.append(x); divs.append(B_P if cross else 1); mods.append(B_I if B_I > 1 else 1)
        # (a call whose leaves were uploaded host batches is keyed by their values: no shape-only record)
        rec = None if uploaded else (cc, dev_params, dev_inputs, B_I, B_P, cross, self._zero_inputs)
        return cc, leaves, divs, mods, B, rec

    def vjp_device(self, params, inputs, cotangent, enc_params=None, force_mean: bool = False):
        """Device-resident adjoint gradient: ``params`` / ``inputs`` are CUDA tensors (as for
        ``__call__``), ``cotangent`` a CUDA tensor ``(B, n_outputs)`` (``(B,)`` with
        ``force_mean``).  Returns ``(grad_params, grad_inputs)`` as CUDA tensors shaped like the
        arguments (None for an argument that is not a CUDA tensor): the gradient of
        ``sum_b sum_k cotangent[b, k] * out[b, k]``.  Nothing per sample happens on the host:
        angle table, forward pass, backward sweep and the chain rule all run on the GPU."""
        import torch

        got = self._forward_device(params, inputs, enc_params, "expval", force_mean,
                                   _want_call=True)
        if got is NotImplemented:
            raise NotImplementedError("this call has no compiled device path")
        cc, leaves, divs, mods, B = got
        n_out = len(cc.obs)
        cot = cotangent.to(device=leaves[0].device, dtype=torch.float32)
        if force_mean and n_out > 1:
            cot = (cot.reshape(B, 1) / n_out).expand(B, n_out)
        cot = cot.reshape(B, n_out).contiguous()
        grads = cc.vjp(leaves, divs, mods, B, cot)
        out = {k: g for k, g in zip(cc.leaf_ids, grads)}
        gp = out.get(0)
        if gp is not None and self._is_cuda(params):
            gp = gp.reshape(tuple(params.shape))
        gx = out.get(1)
        if gx is not None and self._is_cuda(inputs):
            gx = gx.reshape(tuple(inputs.shape))
        return (gp if self._is_cuda(params) else None, gx if self._is_cuda(inputs) else None)

    # Host arrays take the compiled device path too: the circuit is recorded once per
    # structure (what the reference's jit cache does, script.py:475-490), later calls upload
    # the (tiny) parameter / input arrays and download the result.
    host_arrays_via_device = True

    def _forward_host_via_device(self, params, inputs, enc_params, execution_type, force_mean,
                                 data_reupload, as_tensor: bool = False):
        import torch

        if data_reupload is not None:
            self.data_reupload = data_reupload
        p = self._params_validation(params)
        x = self._inputs_validation(inputs)
        zero_special = self.remove_zero_encoding and self._zero_inputs and x.shape[0] == 1
        # batches that do not fit in free HBM in one engine call stay on the chunking path
        from . import memory

        B_P = 1 if 0 in p.shape else int(p.shape[0])
        B = B_P * int(x.shape[0]) if (self.repeat_batch_axis[0] and self.repeat_batch_axis[1]) \
            else max(B_P, int(x.shape[0]))
        et = execution_type or self.execution_type
        if B > 1 and memory.compute_chunk_size(self.n_qubits, B, et, self.noise_params is not None,
                                               self.n_qubits, n_ops=64) < B:
            return NotImplemented
        if 0 not in p.shape:
            p = torch.from_numpy(np.ascontiguousarray(p, dtype=np.float32)).cuda()
        if not zero_special:
            x = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
        out = self._forward_device(p, x, enc_params, execution_type, force_mean)
        if out is NotImplemented:
            return NotImplemented
        return out if as_tensor else N_to_host(out)

    def _forward(self, params=None, inputs=None, pulse_params=None, enc_params=None,
                 data_reupload=None, noise_params=None, execution_type: Optional[str] = None,
                 force_mean: bool = False, gate_mode: str = "unitary",
                 as_tensor: bool = False):
        from .utils import x64_enabled, x64_scope

        with x64_scope(self.x64):
            if x64_enabled():  # complex128: the recorded-tape path only (float64 angle table)
                params = params.detach().cpu().numpy() if self._is_cuda(params) else params
                inputs = inputs.detach().cpu().numpy() if self._is_cuda(inputs) else inputs
            return self._forward_impl(params, inputs, pulse_params, enc_params, data_reupload,
                                      noise_params, execution_type, force_mean, gate_mode, as_tensor,
                                      x64_enabled())

    def _forward_impl(self, params, inputs, pulse_params, enc_params, data_reupload, noise_params,
                      execution_type, force_mean, gate_mode, as_tensor, x64=False):
        user_cuda = self._is_cuda(params) or self._is_cuda(inputs)
        own_dev = params is None and self._params_dev is not None and not x64
        if own_dev:  # the model's own parameters live on the GPU (a large initialize_params draw)
            params = self._params_dev
        if noise_params is not None:
            self.noise_params = noise_params
        # incoherent channels compile like any other tape (vec(rho) on the doubled register, round 5);
        # GateError draws fresh random angles per call and 'state' of a noisy circuit is an error: both
        # keep the recorded path
        compilable = self.noise_params is None or (
            not self.noise_params.get("GateError") and (execution_type or self.execution_type) != "state")
        if (self._is_cuda(params) or self._is_cuda(inputs)) \
                and (not as_tensor or (execution_type or self.execution_type) == "state") \
                and compilable and gate_mode == "unitary" \
                and pulse_params is None:
            if data_reupload is not None:
                self.data_reupload = data_reupload
            out = self._forward_device(params, inputs, enc_params, execution_type, force_mean,
                                       raw=as_tensor)
            if out is not NotImplemented:
                # CUDA tensors in -> CUDA tensor out; host arguments get a host array back
                return out if (user_cuda or as_tensor) else N_to_host(out)
        if own_dev:
            params = None  # (-> the lazily materialised host mirror, self.params)
        params = params.detach().cpu().numpy() if self._is_cuda(params) else params
        inputs = inputs.detach().cpu().numpy() if self._is_cuda(inputs) else inputs
        # (as_tensor: the analysis loops -- Expressibility, Meyer-Wallach -- ask for the raw device
        # tensor of states; the compiled call serves them too: no tape re-recording per call)
        if (compilable
                and gate_mode == "unitary" and pulse_params is None
                and self.host_arrays_via_device and not self._is_cuda(params)
                and not self._is_cuda(inputs) and not x64
                and (not as_tensor or (execution_type or self.execution_type) == "state")):
            out = self._forward_host_via_device(params, inputs, enc_params, execution_type,
                                                force_mean, data_reupload, as_tensor=as_tensor)
            if out is not NotImplemented:
                return out
        if execution_type is not None:
            self.execution_type = execution_type
        self.gate_mode = gate_mode
        if pulse_params is not None and gate_mode != "pulse":
            raise ValueError(
                "pulse_params were provided but gate_mode is not 'pulse'. "
                "Either switch gate_mode='pulse' or do not pass pulse_params."
            )
        if gate_mode == "pulse":
            raise NotImplementedError("gate_mode='pulse' is outside the MI355X hot path")
        if data_reupload is not None:
            self.data_reupload = data_reupload

        params = self._params_validation(params)
        inputs = self._inputs_validation(inputs)
        enc_params = self._enc_params_validation(enc_params)
        inputs, params = self._assimilate_batch(inputs, params)
        self.random_key, sub_key = safe_random_split(self.random_key)

        meas_type, obs = self._build_obs()
        B = int(np.prod(self.eff_batch_shape))
        kwargs = dict(noise_params=self.noise_params, gate_mode=self.gate_mode)
        shot_key = None
        if self.shots is not None:  # model.py:1670-1675
            sub_key, shot_key = safe_random_split(sub_key)

        # probs on a subset of wires: marginalise on the GPU (jaqsi.py:106-146) instead of
        # shipping B x 2^n probabilities to the host
        native_marginal = (meas_type == "probs" and not self.all_qubit_measurement
                           and not as_tensor and self.n_qubits > 10 and self.shots is None)
        if native_marginal:
            meas_type, as_tensor_call = "state", True
        else:
            as_tensor_call = as_tensor

        # one key for the whole batch: GateError draws B values per gate from it (the
        # reference vmaps over B split keys, model.py:1678-1691)
        args = (params, inputs, None, sub_key if self.noise_params is not None else None,
                enc_params)
        if B > 1:
            in_axes = (0 if self.batch_shape[1] > 1 else None,
                       0 if self.batch_shape[0] > 1 else None, None, None, None)
            result = self.script.execute(type=meas_type, obs=obs, args=args, kwargs=kwargs,
                                         in_axes=in_axes, as_tensor=as_tensor_call,
                                         shots=self.shots, key=shot_key)
        else:
            result = self.script.execute(type=meas_type, obs=obs, args=args, kwargs=kwargs,
                                         as_tensor=as_tensor_call, shots=self.shots,
                                         key=shot_key)
        if as_tensor:
            return result  # raw (B, ...) device tensor for the analysis loops
        if native_marginal:
            from . import _native

            states = result.reshape(-1, 2**self.n_qubits)
            groups = (self.output_qubit if isinstance(self.output_qubit[0], (list, tuple))
                      else [self.output_qubit])
            parts = [_native.marginal_probs(states, list(g)).cpu().numpy() for g in groups]
            result = np.stack(parts) if len(parts) > 1 else parts[0]
            result = np.asarray(result)
            result = result.reshape((*self.eff_batch_shape, *self._result_shape)).squeeze()
            if force_mean and len(result.shape) > 0 and self._result_shape[0] > 1:
                result = result.mean(axis=-1)
            return result

        if self.execution_type == "density" and not self.all_qubit_measurement:
            result = js.partial_trace(result, self.n_qubits, self.output_qubit)
        if self.execution_type == "probs" and not self.all_qubit_measurement:
            if isinstance(self.output_qubit[0], (list, tuple)):
                result = np.stack([js.marginalize_probs(result, self.n_qubits, list(g))
                                   for g in self.output_qubit])
            else:
                result = js.marginalize_probs(result, self.n_qubits, self.output_qubit)

        result = np.asarray(result)
        result = result.reshape((*self.eff_batch_shape, *self._result_shape)).squeeze()
        if (self.execution_type in ("expval", "probs") and force_mean and len(result.shape) > 0
                and self._result_shape[0] > 1):
            result = result.mean(axis=-1)
        return result
