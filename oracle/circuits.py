"""Gate-list generator of the reference Model, restated (oracle; test infra only).

Produces the *tape* ``[(name, wires, params), ...]`` that
``Model._variational`` (``qml_essentials/model.py:818-963``) records for one
sample, so that the product front-end's op-list can be checked gate by gate.

Follows
  * ``topologies.py:21-121``   (stairs / bricks / all_to_all)
  * ``ansaetze.py:215-221``    (DeclarativeCircuit.build: block, then Barrier)
  * ``ansaetze.py:323-371``    (Block.apply: weight consumption order)
  * ``ansaetze.py:408-756``    (the 23 structures)
  * ``model.py:746-816``       (_iec), ``:913-959`` (layer order), ``:469-512``
    (data_reupload mask, degree, has_dru)
  * ``ansaetze.py:815-848,915-961`` (hamming / binary / ternary encodings)
"""
import numpy as np

from . import gates as G

ROTATIONAL = {"RX", "RY", "RZ", "Rot", "CRX", "CRY", "CRZ", "CPhase"}  # gates.py:209-221
ENTANGLING = {"CX", "CY", "CZ", "CRX", "CRY", "CRZ", "CPhase"}  # gates.py:223-225


# --- topologies.py ----------------------------------------------------------
def stairs(n_qubits, offset=0, wrap=False, reverse=True, mirror=True, span=1,
           stride=1, modulo=True):
    """topologies.py:21-100."""
    ctrls, targets = [], []
    n_gates = n_qubits if wrap else n_qubits - 1
    _offset = offset(n_qubits) if callable(offset) else offset
    _span = span(n_qubits) if callable(span) else span
    for q in range(0, n_gates, stride):
        t = q + _offset + _span
        if t >= n_qubits and not modulo:
            continue
        c = q + _offset
        if c < 0 and not modulo:
            continue
        t %= n_qubits
        c %= n_qubits
        if t == c:
            continue
        targets.append(t)
        ctrls.append(c)
    if reverse:
        ctrls, targets = ctrls[::-1], targets[::-1]
    if mirror:
        ctrls, targets = targets, ctrls
    return list(zip(ctrls, targets))


def bricks(n_qubits, **kw):
    """topologies.py:102-106."""
    kw.setdefault("stride", 2)
    kw.setdefault("modulo", False)
    return stairs(n_qubits, **kw)


def all_to_all(n_qubits):
    """topologies.py:108-121."""
    return [
        (n_qubits - ql - 1, (n_qubits - q - 1) % n_qubits)
        for ql in range(n_qubits)
        for q in range(n_qubits)
        if q != ql
    ]


_nm1 = lambda n: n - 1  # noqa: E731
_nm2 = lambda n: n - 2  # noqa: E731
_nh = lambda n: n // 2  # noqa: E731

# (gate, topology fn or None, kwargs) per block -- ansaetze.py:408-756
STRUCTURES = {
    "No_Ansatz": [],
    "Circuit_1": [("RX",), ("RZ",)],
    "Circuit_2": [("RX",), ("RZ",), ("CX", stairs, {})],
    "Circuit_3": [("RX",), ("RZ",), ("CRZ", stairs, {})],
    "Circuit_4": [("RX",), ("RZ",), ("CRX", stairs, {})],
    "Circuit_5": [("RX",), ("RZ",), ("CRZ", all_to_all, {}), ("RX",), ("RZ",)],
    "Circuit_6": [("RX",), ("RZ",), ("CRX", all_to_all, {}), ("RX",), ("RZ",)],
    "Circuit_7": [("RX",), ("RZ",), ("CRZ", bricks, {}), ("RX",), ("RZ",),
                  ("CRZ", bricks, dict(offset=1))],
    "Circuit_8": [("RX",), ("RZ",), ("CRX", bricks, {}), ("RX",), ("RZ",),
                  ("CRX", bricks, dict(offset=1))],
    "Circuit_9": [("H",), ("CZ", stairs, {}), ("RX",)],
    "Circuit_10": [("RY",), ("CZ", stairs, dict(offset=-1, wrap=True)), ("RY",)],
    "Circuit_13": [("RY",), ("CRZ", stairs, dict(wrap=True, reverse=True, mirror=False)),
                   ("RY",),
                   ("CRZ", stairs, dict(reverse=False, mirror=False, offset=_nm1,
                                        span=3, wrap=True))],
    "Circuit_14": [("RY",), ("CRX", stairs, dict(wrap=True, reverse=True, mirror=False)),
                   ("RY",),
                   ("CRX", stairs, dict(reverse=False, mirror=False, offset=_nm1,
                                        span=3, wrap=True))],
    "Circuit_15": [("RY",), ("CX", stairs, dict(wrap=True, reverse=True, mirror=False)),
                   ("RY",),
                   ("CX", stairs, dict(reverse=False, mirror=False, offset=_nm1,
                                       span=3, wrap=True))],
    "Circuit_16": [("RX",), ("RZ",), ("CRZ", bricks, {}), ("CRZ", bricks, dict(offset=1))],
    "Circuit_17": [("RX",), ("RZ",), ("CRX", bricks, {}), ("CRX", bricks, dict(offset=1))],
    "Circuit_18": [("RX",), ("RZ",), ("CRZ", stairs, dict(wrap=True, mirror=False))],
    "Circuit_19": [("RX",), ("RZ",), ("CRX", stairs, dict(wrap=True, mirror=False))],
    "Circuit_20": [("RY",), ("CX", stairs, dict(wrap=True, reverse=True, mirror=False)),
                   ("RY",),
                   ("CX", stairs, dict(reverse=False, offset=_nm2, span=1, wrap=True))],
    "No_Entangling": [("Rot",)],
    "Hardware_Efficient": [("RY",), ("RZ",), ("RY",), ("CX", bricks, dict(mirror=False)),
                           ("CX", bricks, dict(offset=-1, modulo=True, wrap=True,
                                               mirror=False))],
    "Strongly_Entangling": [("Rot",),
                            ("CX", stairs, dict(wrap=True, reverse=False, mirror=False)),
                            ("Rot",),
                            ("CX", stairs, dict(reverse=False, span=_nh, wrap=True,
                                                mirror=False))],
}


def _enough_qubits(block, n):
    """ansaetze.py:271-283."""
    if block[0] in ENTANGLING:
        span = block[2].get("span", 1)
        if callable(span):
            span = span(n)
        return n >= 2 and n > span
    return n >= 1


def block_n_params(block, n):
    """ansaetze.py:285-303."""
    gate = block[0]
    if gate in ROTATIONAL:
        if gate in ENTANGLING:
            return len(block[1](n, **block[2])) if _enough_qubits(block, n) else 0
        return 3 * n if gate == "Rot" else n
    return 0


def n_params_per_layer(ansatz, n):
    """ansaetze.py:173-175."""
    if ansatz == "GHZ":
        return 0
    return sum(block_n_params(b, n) for b in STRUCTURES[ansatz])


def ansatz_layer(ansatz, w, n):
    """One ansatz layer: ansaetze.py:215-221 + :323-371 (GHZ: :423-427)."""
    tape = []
    if ansatz == "GHZ":
        tape.append(("H", [0], ()))
        for q in range(n - 1):
            tape.append(("CX", [q, q + 1], ()))
        return tape
    w_idx = 0
    for block in STRUCTURES[ansatz]:
        gate = block[0]
        ent = gate in ENTANGLING
        it = block[1](n, **block[2]) if ent else range(n)
        for wires in it:
            if ent and not _enough_qubits(block, n):
                continue
            wl = list(wires) if ent else [wires]
            if gate in ROTATIONAL:
                if gate == "Rot":
                    tape.append((gate, wl, tuple(float(x) for x in w[w_idx:w_idx + 3])))
                    w_idx += 3
                else:
                    tape.append((gate, wl, (float(w[w_idx]),)))
                    w_idx += 1
            else:
                tape.append((gate, wl, ()))
        tape.append(("Barrier", list(range(n)), ()))
    return tape


# --- Encoding (ansaetze.py:759-1000) ----------------------------------------
def enc_n_freqs(strategy, omegas, n_qubits=None):
    """ansaetze.py:815-848."""
    if strategy == "hamming":
        return int(2 * omegas + 1)
    if strategy == "binary":
        return int(2 ** (omegas + 1) - 1)
    if strategy == "ternary":
        return int(3**omegas)
    if strategy == "golomb":
        return int(2 * omegas * max(G.golomb_ruler(2**n_qubits)) + 1)
    raise NotImplementedError


def enc_spectrum(strategy, omegas, n_qubits=None):
    """ansaetze.py:850-893."""
    if strategy == "hamming":
        return np.arange(-omegas, omegas + 1)
    if strategy == "binary":
        return np.arange(-(2**omegas) + 1, 2**omegas)
    if strategy == "ternary":
        lim = int(np.floor(3**omegas / 2))
        return np.arange(-lim, lim + 1)
    if strategy == "golomb":
        lim = omegas * max(G.golomb_ruler(2**n_qubits))
        return np.arange(-lim, lim + 1)
    raise NotImplementedError


class ModelSpec:
    """Structural subset of ``Model.__init__`` (model.py:26-210) the tape depends on."""

    def __init__(self, n_qubits, n_layers, circuit_type="No_Ansatz", data_reupload=True,
                 encoding="RX", strategy="hamming", state_preparation=None,
                 remove_zero_encoding=True):
        self.n_qubits, self.n_layers, self.ansatz = n_qubits, n_layers, circuit_type
        self.enc_gates = [encoding] if isinstance(encoding, str) else list(encoding)
        self.strategy = strategy
        self.n_input_feat = 1 if strategy == "golomb" else len(self.enc_gates)
        self.sp = ([] if state_preparation is None else
                   [state_preparation] if isinstance(state_preparation, str)
                   else list(state_preparation))
        self.remove_zero_encoding = remove_zero_encoding
        F = self.n_input_feat
        if isinstance(data_reupload, bool):  # model.py:492-499
            if data_reupload:
                dru = np.ones((n_layers, n_qubits, F))
            else:
                dru = np.zeros((n_layers, n_qubits, F))
                dru[0][0] = 1
        else:  # model.py:471-489
            dru = np.array(data_reupload)
            if dru.ndim == 2:
                dru = np.repeat(dru.reshape(*dru.shape, 1), F, axis=2)
        self.data_reupload = dru.astype(bool)
        cnt = [int(np.count_nonzero(self.data_reupload[..., i])) for i in range(F)]
        self.degree = tuple(enc_n_freqs(strategy, c, n_qubits) for c in cnt)  # :500-503
        self.frequencies = tuple(enc_spectrum(strategy, c, n_qubits) for c in cnt)
        self.has_dru = bool(max(int(np.max(f)) for f in self.frequencies) > 1)  # :512
        self.impl_n_layers = n_layers + 1 if self.has_dru else n_layers  # :163-166
        self.params_shape = (self.impl_n_layers, n_params_per_layer(circuit_type, n_qubits))
        self.enc_params = np.ones((n_layers, n_qubits, F))  # model.py:150


def model_tape(spec, params, inputs, enc_params=None, zero_inputs_batch1=None):
    """Tape of one sample: model.py:913-959 with _iec (model.py:746-816).

    ``params`` has shape ``params_shape``; ``inputs`` shape ``(n_input_feat,)``.
    ``zero_inputs_batch1``: emulate ``_zero_inputs and batch_shape[0]==1``
    (model.py:782-783); default = inputs are all zero.
    """
    n = spec.n_qubits
    params = np.asarray(params, dtype=np.float64)
    inputs = np.asarray(inputs, dtype=np.float64).reshape(-1)
    enc_params = spec.enc_params if enc_params is None else np.asarray(enc_params)
    if zero_inputs_batch1 is None:
        zero_inputs_batch1 = not inputs.any()
    tape = []
    for q in range(n):  # model.py:914-923
        for g in spec.sp:
            tape.append((g, [q], ()))

    def iec(layer):
        if spec.remove_zero_encoding and zero_inputs_batch1:  # :782-783
            return
        dru = spec.data_reupload[layer]
        ep = enc_params[layer]
        if spec.strategy == "golomb":  # :786-802
            if dru[:, 0].any():
                x = inputs[0] * np.mean(ep[:, 0])
                tape.append(("DiagU", list(range(n)), (G.golomb_diag(x, n),)))
            return
        for q in range(n):  # :804-816
            for idx in range(inputs.shape[-1]):
                if dru[q, idx]:
                    ang = inputs[idx] * ep[q, idx]
                    if spec.strategy == "binary":  # ansaetze.py:933-936
                        ang = ang * (2**q)
                    elif spec.strategy == "ternary":  # ansaetze.py:958-961
                        ang = ang * (3**q)
                    tape.append((spec.enc_gates[idx], [q], (float(ang),)))

    for layer in range(spec.n_layers):  # model.py:926-947
        tape += ansatz_layer(spec.ansatz, params[layer], n)
        iec(layer)
    if spec.has_dru:  # model.py:950-959
        tape += ansatz_layer(spec.ansatz, params[spec.n_layers], n)
    return tape
