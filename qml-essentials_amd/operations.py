"""Gate / observable records of the front-end (host side, NumPy only).

API mirror of the gate classes in ``qml_essentials/operations.py`` (wires,
parameters, ``matrix``, self-registration on the active tape, wire validation
``operations.py:140-146``).  Unlike the reference, an operation here is a *record*:
the statevector arithmetic is done by ``libqmle_sv`` (HIP), which receives the
record through :meth:`Operation.lower`.  Parameters may be floats or per-sample
columns (see :mod:`batching`), so one tape serves a whole batch.

Noise channels (``KrausChannel`` and subclasses, ``operations.py:1490-1929``) are records
too: on the engine a channel is the superoperator ``sum_k K_k (x) conj(K_k)`` acting on the
ket and bra copies of its wires in the vectorised density matrix (see ``simulation.py``).

Out of scope here: ``ParametrizedHamiltonian``, ``PauliWord``
algebra, pulse evolution.
"""
from __future__ import annotations

from functools import reduce
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np

from .batching import Batched, as_param, param_tangent
from .tape import active_tape, recording  # noqa: F401  (re-export, like the reference)

CDTYPE = np.complex64  # reference default: JAX x64 off (operations.py:12-16)


def _cdtype():
    return CDTYPE


_I2 = np.eye(2, dtype=np.complex128)
_X = np.array([[0, 1], [1, 0]], dtype=np.complex128)
_Y = np.array([[0, -1j], [1j, 0]], dtype=np.complex128)
_Z = np.array([[1, 0], [0, -1]], dtype=np.complex128)
_PAULI = {"I": _I2, "X": _X, "Y": _Y, "Z": _Z}
_P0 = np.diag([1.0, 0.0]).astype(np.complex128)
_P1 = np.diag([0.0, 1.0]).astype(np.complex128)


def _wire_list(wires) -> List[int]:
    if isinstance(wires, (list, tuple, np.ndarray, range)):
        return [int(w) for w in wires]
    return [int(wires)]


def _pauli_word_matrix(word: str) -> np.ndarray:
    return reduce(np.kron, [_PAULI[c] for c in word])


def _rotation(theta, generator: np.ndarray) -> np.ndarray:
    """cos(t/2) I - i sin(t/2) P; ``theta`` float or (B,) column -> (d,d) or (B,d,d)."""
    th = np.asarray(theta, dtype=np.float64)
    c = np.cos(th / 2)[..., None, None]
    s = np.sin(th / 2)[..., None, None]
    return c * np.eye(generator.shape[0]) - 1j * s * generator


def _controlled(block: np.ndarray, n_controls: int = 1) -> np.ndarray:
    """Identity with ``block`` in the all-controls-set corner (batched blocks ok)."""
    d_t = block.shape[-1]
    dim = (2**n_controls) * d_t
    out = np.zeros(block.shape[:-2] + (dim, dim), dtype=np.complex128)
    idx = np.arange(dim - d_t)
    out[..., idx, idx] = 1.0
    out[..., dim - d_t:, dim - d_t:] = block
    return out


class Operation:
    """A gate or observable acting on ``wires``.

    Created inside a circuit function it appends itself to the active tape
    (``operations.py:151-155``).  ``matrix`` overrides the class default.
    """

    is_controlled = False
    is_clifford = False
    _matrix: Optional[np.ndarray] = None
    _num_wires: Optional[int] = None
    _param_names: Tuple[str, ...] = ()
    _native: Optional[str] = None  # engine opcode name (include/qmle_sv.h)
    # parameter-shift rule of the gate parameters: "two" = generator spectrum {-1/2, +1/2} or
    # {0, 1}; "four" = controlled rotation (spectrum {0, +-1/2}); None = not differentiable
    _shift_rule: Optional[str] = None

    def _store_param(self, name: str, value) -> None:
        """Keep the numeric value (float / (B,) column) and, when the value came from a
        differentiable leaf, its tangent terms (see :mod:`batching`)."""
        setattr(self, name, as_param(value))
        if "_tangents" not in self.__dict__:
            self._tangents = {}
        self._tangents[name] = param_tangent(value)

    @property
    def parameter_tangents(self) -> list:
        t = self.__dict__.get("_tangents", {})
        return [t.get(n, []) for n in self._param_names]

    def __init__(self, wires: Union[int, Sequence[int]] = 0, matrix=None, record: bool = True,
                 name: Optional[str] = None) -> None:
        self.name = name or type(self).__name__
        self._wires = _wire_list(wires)
        k = self._num_wires
        if k is not None and len(self._wires) != k:
            raise ValueError(
                f"{self.name} expects {k} wire(s), got {len(self._wires)}: {self._wires}"
            )
        if len(set(self._wires)) != len(self._wires):
            raise ValueError(f"{self.name} received duplicate wires: {self._wires}")
        if matrix is not None:
            self._matrix = np.asarray(matrix)
        if record:
            tape = active_tape()
            if tape is not None:
                tape.append(self)

    # ---- structure ------------------------------------------------------------------
    @property
    def wires(self) -> List[int]:
        return self._wires

    @wires.setter
    def wires(self, value) -> None:
        self._wires = _wire_list(value)

    @property
    def parameters(self) -> list:
        return [getattr(self, n) for n in self._param_names]

    @property
    def is_batched(self) -> bool:
        return any(isinstance(p, np.ndarray) and p.ndim > 0 for p in self.parameters)

    def __repr__(self) -> str:
        ps = self.parameters
        if ps:
            txt = ", ".join(
                f"{float(p):.4f}" if np.ndim(p) == 0 else f"<batch {np.shape(p)[0]}>" for p in ps
            )
            return f"{self.name}({txt}, wires={self.wires})"
        return f"{self.name}(wires={self.wires})"

    # ---- numerics (host, small matrices only) -------------------------------------------
    def _build_matrix(self) -> Optional[np.ndarray]:
        return None

    @property
    def matrix(self) -> np.ndarray:
        m = self._matrix
        if m is None:
            m = self._build_matrix()
        if m is None:
            raise NotImplementedError(f"{type(self).__name__} does not define a matrix.")
        return m

    def decompose(self) -> List["Operation"]:
        raise NotImplementedError(f"{type(self).__name__} does not define a decomposition.")

    def _replace_on_tape(self, new: "Operation") -> None:
        tape = active_tape()
        if tape is not None:
            if tape and tape[-1] is self:
                tape[-1] = new
            else:
                tape.append(new)

    def dagger(self) -> "Operation":
        """U -> U^dagger; replaces ``self`` on the tape when chained right after creation."""
        m = self.matrix
        new = Operation(wires=self.wires, matrix=np.conj(np.swapaxes(m, -1, -2)), record=False)
        self._replace_on_tape(new)
        return new

    def power(self, power: int) -> "Operation":
        new = Operation(wires=self.wires, matrix=np.linalg.matrix_power(self.matrix, power),
                        record=False)
        self._replace_on_tape(new)
        return new

    def __mul__(self, other):
        if isinstance(other, Operation):
            return self.__matmul__(other)
        new = Operation(wires=self.wires, matrix=other * self.matrix, record=False)
        self._replace_on_tape(new)
        return new

    __rmul__ = __mul__

    def __add__(self, other: "Operation") -> "Operation":
        if sorted(self.wires) != sorted(other.wires):
            raise ValueError(
                "Can only add operations acting on the same set of wires, "
                f"got {self.wires} and {other.wires}"
            )
        return Operation(wires=self.wires, matrix=self.matrix + other.matrix, record=False)

    def prod(self, *ops: "Operation") -> "Operation":
        """Kronecker / matrix product on the union of the wire sets."""
        if not ops:
            return self
        every = (self,) + ops
        wires: List[int] = []
        for o in every:
            for w in o.wires:
                if w not in wires:
                    wires.append(w)
        mat = np.eye(2 ** len(wires), dtype=np.complex128)
        for o in every:
            mat = mat @ embed_matrix(o.matrix, o.wires, wires)
        return Operation(wires=wires, matrix=mat, name="Prod(" + "*".join(o.name for o in every) + ")",
                         record=False)

    def __matmul__(self, other):
        if not isinstance(other, Operation):
            return NotImplemented
        return self.prod(other)

    def lifted_matrix(self, n_qubits: int) -> np.ndarray:
        """Full 2^n x 2^n embedding (small n only; observables of the general path)."""
        return embed_matrix(self.matrix, self.wires, list(range(n_qubits)))

    # ---- hand-off to the HIP engine ------------------------------------------------------
    def lower(self, n_qubits: int):
        """-> (opcode name, wires, [params], const float32 blob or None)."""
        if self._native is not None:
            return self._native, self.wires, list(self.parameters), None
        m = np.asarray(self.matrix)
        if m.ndim != 2:
            raise NotImplementedError(
                f"{self.name}: explicit matrices must be batch-constant for the HIP engine"
            )
        k = len(self.wires)
        if m.shape != (2**k, 2**k):
            raise ValueError(f"{self.name}: matrix shape {m.shape} does not match {k} wire(s)")
        if k > 2:
            raise NotImplementedError(f"{self.name}: generic {k}-qubit matrices are not supported")
        # (float64: LoweredTape keeps this copy for the complex128 engine and casts for the complex64 one)
        blob = np.stack([m.real, m.imag], axis=-1).astype(np.float64).reshape(-1)
        return ("MAT1" if k == 1 else "MAT2"), self.wires, [], blob


def _neg(x):
    return -x if not isinstance(x, np.ndarray) else -x


# how U* (the gate acting on the bra wires of vec(rho), operations.py:505-510) is obtained
# from U for the named gates: negate these parameter positions, keep the others
_CONJ_NEGATE = {"RX": (0,), "RZ": (0,), "CRX": (0,), "CRZ": (0,), "CPhase": (0,), "RXX": (0,),
                "RYY": (0,), "RZZ": (0,), "RZX": (0,), "Rot": (0, 2), "RY": (), "CRY": (),
                "H": (), "PauliX": (), "PauliZ": (), "CX": (), "CZ": (), "SWAP": (), "CCX": (),
                "CSWAP": (), "Id": ()}


def conj_lower(op_: "Operation", n_qubits: int, offset: int):
    """Lowered form of ``conj(U)`` on wires shifted by ``offset`` (the bra register)."""
    low = op_.lower(n_qubits)
    if low is None:
        return None
    name, wires, params, blob = low
    wires = [w + offset for w in wires]
    if name in _CONJ_NEGATE:
        neg = _CONJ_NEGATE[name]
        return name, wires, [(-p if j in neg else p) for j, p in enumerate(params)], None
    if name in ("MAT1", "MAT2", "MAT4"):
        b = np.array(blob, dtype=np.float64).reshape(-1, 2)
        b[:, 1] *= -1
        return name, wires, [], b.reshape(-1)
    if name == "DIAG_ALL":
        raise NotImplementedError("full-register diagonal encodings in density-matrix mode")
    # PauliY, CY, S ...: conjugate the explicit matrix
    m = np.conj(np.asarray(op_.matrix))
    if m.ndim != 2 or len(wires) > 2:
        raise NotImplementedError(f"conj({op_.name}) is not available on the engine")
    blob = np.stack([m.real, m.imag], axis=-1).astype(np.float64).reshape(-1)
    return ("MAT1" if len(wires) == 1 else "MAT2"), wires, [], blob


def prod(*ops: "Operation") -> "Operation":
    """Generalised product of operations: Kronecker product on disjoint wires, matrix product
    where wires overlap (module-level form of :meth:`Operation.prod`)."""
    if not ops:
        raise ValueError("prod() needs at least one operation")
    return ops[0].prod(*ops[1:])


def embed_matrix(mat: np.ndarray, wires: Sequence[int], all_wires: Sequence[int]) -> np.ndarray:
    """Embed ``mat`` (on ``wires``) into the space spanned by ``all_wires`` (first = MSB)."""
    n, k = len(all_wires), len(wires)
    pos = [list(all_wires).index(w) for w in wires]
    t = np.asarray(mat, dtype=np.complex128).reshape((2,) * (2 * k))
    full = np.eye(2**n, dtype=np.complex128).reshape((2,) * (2 * n))
    # contract gate "in" legs with the row legs of the identity
    full = np.tensordot(t, full, axes=(list(range(k, 2 * k)), pos))
    full = np.moveaxis(full, list(range(k)), pos)
    return full.reshape(2**n, 2**n)


class Hermitian(Operation):
    """Generic Hermitian observable / explicit-matrix gate (``operations.py:515``)."""

    def __init__(self, matrix, wires=0, record: bool = True, **kw) -> None:
        super().__init__(wires=wires, matrix=matrix, record=record, **kw)


class Id(Operation):
    _matrix = _I2
    _native = "Id"
    is_clifford = True

    def __init__(self, wires=0, **kw) -> None:
        k = len(_wire_list(wires))
        if k > 1:
            kw["matrix"] = np.eye(2**k, dtype=np.complex128)
        super().__init__(wires=wires, **kw)

    def lower(self, n_qubits: int):
        return "Id", self.wires[:1], [], None


def _fixed_gate(cls_name: str, mat: np.ndarray, n_wires: int, native: str, clifford=True,
                controlled=False, doc: str = ""):
    def __init__(self, wires=list(range(n_wires)) if n_wires > 1 else 0, **kw):
        Operation.__init__(self, wires=wires, **kw)

    return type(cls_name, (Operation,), {
        "_matrix": mat, "_num_wires": n_wires, "_native": native, "is_clifford": clifford,
        "is_controlled": controlled, "__init__": __init__, "__doc__": doc,
    })


PauliX = _fixed_gate("PauliX", _X, 1, "PauliX", doc="Pauli-X gate / observable.")
PauliY = _fixed_gate("PauliY", _Y, 1, "PauliY", doc="Pauli-Y gate / observable.")
PauliZ = _fixed_gate("PauliZ", _Z, 1, "PauliZ", doc="Pauli-Z gate / observable.")
H = _fixed_gate("H", np.array([[1, 1], [1, -1]], dtype=np.complex128) / np.sqrt(2), 1, "H",
                doc="Hadamard gate.")
S = _fixed_gate("S", np.diag([1, 1j]).astype(np.complex128), 1, "S", doc="Phase gate sqrt(Z).")
SWAP = _fixed_gate("SWAP", np.eye(4, dtype=np.complex128)[[0, 2, 1, 3]], 2, "SWAP", doc="SWAP.")
CX = _fixed_gate("CX", _controlled(_X), 2, "CX", controlled=True, doc="wires=[control,target]")
CY = _fixed_gate("CY", _controlled(_Y), 2, "CY", controlled=True, doc="wires=[control,target]")
CZ = _fixed_gate("CZ", _controlled(_Z), 2, "CZ", controlled=True, doc="wires=[control,target]")
CCX = _fixed_gate("CCX", _controlled(_X, 2), 3, "CCX", clifford=False, controlled=True,
                  doc="Toffoli, wires=[c0,c1,target]")
CSWAP = _fixed_gate("CSWAP", _controlled(np.eye(4, dtype=np.complex128)[[0, 2, 1, 3]]), 3, "CSWAP",
                    clifford=False, controlled=True, doc="Fredkin, wires=[control,t0,t1]")


def _cz_decompose(self):
    c, t = self.wires
    return [H(wires=t, record=False), CX(wires=[c, t], record=False), H(wires=t, record=False)]


CZ.decompose = _cz_decompose


class Barrier(Operation):
    """No-op separator: recorded, skipped by the simulator (``simulation.py:93-94``)."""

    def __init__(self, wires=0) -> None:
        super().__init__(wires=wires)

    def lower(self, n_qubits: int):
        return None


class _Rotation(Operation):
    """exp(-i theta/2 P), P in {X,Y,Z}  (``operations.py:1002-1045``)."""

    _num_wires = 1
    _param_names = ("theta",)
    _axis = "X"

    _shift_rule = "two"

    def __init__(self, theta, wires=0, **kw) -> None:
        self._store_param("theta", theta)
        super().__init__(wires=wires, **kw)

    def _build_matrix(self):
        return _rotation(self.theta, _PAULI[self._axis])

    def generator(self) -> Operation:
        return {"X": PauliX, "Y": PauliY, "Z": PauliZ}[self._axis](wires=self.wires[0], record=False)


RX = type("RX", (_Rotation,), {"_axis": "X", "_native": "RX"})
RY = type("RY", (_Rotation,), {"_axis": "Y", "_native": "RY"})
RZ = type("RZ", (_Rotation,), {"_axis": "Z", "_native": "RZ"})


class Rot(Operation):
    """Rot(phi, theta, omega) = RZ(omega) RY(theta) RZ(phi)  (``operations.py:1204-1243``)."""

    _num_wires = 1
    _param_names = ("phi", "theta", "omega")
    _native = "Rot"
    _shift_rule = "two"

    def __init__(self, phi, theta, omega, wires=0, **kw) -> None:
        self._store_param("phi", phi)
        self._store_param("theta", theta)
        self._store_param("omega", omega)
        super().__init__(wires=wires, **kw)

    def _build_matrix(self):
        return _rotation(self.omega, _Z) @ _rotation(self.theta, _Y) @ _rotation(self.phi, _Z)

    def decompose(self):
        w = self.wires[0]
        return [RZ(self.phi, wires=w, record=False), RY(self.theta, wires=w, record=False),
                RZ(self.omega, wires=w, record=False)]


class ControlledPhaseShift(Operation):
    """diag(1, 1, 1, e^{i phi}); wires=[control,target]  (``operations.py:1171-1201``)."""

    _num_wires = 2
    _param_names = ("phi",)
    _native = "CPhase"
    is_controlled = True
    _shift_rule = "two"  # generator |11><11| has spectrum {0, 1}

    def __init__(self, phi, wires=(0, 1), **kw) -> None:
        self._store_param("phi", phi)
        super().__init__(wires=wires, **kw)

    def _build_matrix(self):
        ph = np.exp(1j * np.asarray(self.phi, dtype=np.float64))
        blk = np.zeros(np.shape(ph) + (2, 2), dtype=np.complex128)
        blk[..., 0, 0] = 1.0
        blk[..., 1, 1] = ph
        return _controlled(blk)


class PauliRot(Operation):
    """exp(-i theta/2 P) for a Pauli word P  (``operations.py:1255-1315``)."""

    _param_names = ("theta",)
    _NATIVE_WORDS = {"X": "RX", "Y": "RY", "Z": "RZ", "XX": "RXX", "YY": "RYY", "ZZ": "RZZ",
                     "ZX": "RZX"}

    _shift_rule = "two"

    def __init__(self, theta, pauli_word: str, wires=0, **kw) -> None:
        self._store_param("theta", theta)
        self.pauli_word = pauli_word
        super().__init__(wires=wires, **kw)
        if len(self.wires) != len(pauli_word):
            raise ValueError(
                f"{self.name} expects {len(pauli_word)} wire(s), got {len(self.wires)}: {self.wires}"
            )

    def _build_matrix(self):
        return _rotation(self.theta, _pauli_word_matrix(self.pauli_word))

    def generator(self) -> Operation:
        return Hermitian(_pauli_word_matrix(self.pauli_word), wires=self.wires, record=False)

    def lower(self, n_qubits: int):
        native = self._NATIVE_WORDS.get(self.pauli_word)
        if native is not None:
            return native, self.wires, [self.theta], None
        return Operation.lower(self, n_qubits)


def _pauli_rot_subclass(name: str, word: str):
    def __init__(self, theta, wires=None, **kw):
        PauliRot.__init__(self, theta, word, wires=list(range(len(word))) if wires is None else wires,
                          **kw)

    return type(name, (PauliRot,), {"_num_wires": len(word), "__init__": __init__,
                                    "__doc__": f"{name}(theta) = exp(-i theta/2 {' x '.join(word)})"})


RXX = _pauli_rot_subclass("RXX", "XX")
RYY = _pauli_rot_subclass("RYY", "YY")
RZZ = _pauli_rot_subclass("RZZ", "ZZ")
RZX = _pauli_rot_subclass("RZX", "ZX")


class ControlledPauliRot(Operation):
    """PauliRot on the targets, conditioned on all controls = 1
    (``operations.py:1357-1427``); wires = [controls..., targets...]."""

    _param_names = ("theta",)
    is_controlled = True

    _shift_rule = "four"

    def __init__(self, theta, pauli_word: str, wires, n_controls: int = 1, **kw) -> None:
        self._store_param("theta", theta)
        self.pauli_word = pauli_word
        self.n_controls = n_controls
        wl = _wire_list(wires)
        if len(wl) != n_controls + len(pauli_word):
            raise ValueError(
                f"ControlledPauliRot expects {n_controls + len(pauli_word)} wires "
                f"({n_controls} control + {len(pauli_word)} target), got {len(wl)}."
            )
        super().__init__(wires=wl, **kw)

    def _build_matrix(self):
        return _controlled(_rotation(self.theta, _pauli_word_matrix(self.pauli_word)),
                           self.n_controls)

    def generator(self) -> Operation:
        P = _pauli_word_matrix(self.pauli_word)
        dim = (2**self.n_controls) * P.shape[0]
        gen = np.zeros((dim, dim), dtype=np.complex128)
        gen[dim - P.shape[0]:, dim - P.shape[0]:] = P
        return Hermitian(gen, wires=self.wires, record=False)

    def lower(self, n_qubits: int):
        if self.n_controls == 1 and self.pauli_word in ("X", "Y", "Z"):
            return "CR" + self.pauli_word, self.wires, [self.theta], None
        return Operation.lower(self, n_qubits)


def _controlled_rotation(name: str, axis: str):
    def __init__(self, theta, wires=(0, 1), **kw):
        ControlledPauliRot.__init__(self, theta, axis, wires=wires, n_controls=1, **kw)

    def decompose(self):
        c, t = self.wires
        th = self.theta
        core = [RZ(th / 2, wires=t, record=False), CX(wires=[c, t], record=False),
                RZ(-th / 2, wires=t, record=False), CX(wires=[c, t], record=False)]
        if axis == "Z":
            return core
        if axis == "X":
            return [H(wires=t, record=False)] + core + [H(wires=t, record=False)]
        return [RX(-np.pi / 2, wires=t, record=False), RZ(th / 2, wires=t, record=False),
                CX(wires=[c, t], record=False), RZ(-th / 2, wires=t, record=False),
                RX(np.pi / 2, wires=t, record=False)]

    return type(name, (ControlledPauliRot,), {"_num_wires": 2, "__init__": __init__,
                                              "decompose": decompose})


CRX = _controlled_rotation("CRX", "X")
CRY = _controlled_rotation("CRY", "Y")
CRZ = _controlled_rotation("CRZ", "Z")


class DiagonalQubitUnitary(Operation):
    """diag(d_0 .. d_{2^k-1}) on ``wires``  (``operations.py:881-942``).

    :meth:`from_phases` builds ``exp(-i * marks * x)`` with a per-sample ``x`` -- the
    Golomb data encoding (``unitary.py:661-701``) -- which the engine applies as one
    full-register diagonal pass.
    """

    def __init__(self, diag, wires=0, **kw) -> None:
        wl = _wire_list(wires)
        diag = np.asarray(diag)
        if diag.shape != (2 ** len(wl),):
            raise ValueError(
                f"DiagonalQubitUnitary expects {2 ** len(wl)} diagonal entries "
                f"for {len(wl)} wire(s), got shape {diag.shape}"
            )
        self.diag = diag
        self._marks, self._scale = None, None
        kw.setdefault("name", "DiagU")
        super().__init__(wires=wl, **kw)

    @classmethod
    def from_phases(cls, marks, x, wires, **kw) -> "DiagonalQubitUnitary":
        marks = np.asarray(marks, dtype=np.float64)
        scale = as_param(x)
        self = cls.__new__(cls)
        wl = _wire_list(wires)
        if marks.shape != (2 ** len(wl),):
            raise ValueError(
                f"DiagonalQubitUnitary expects {2 ** len(wl)} diagonal entries "
                f"for {len(wl)} wire(s), got shape {marks.shape}"
            )
        self._marks = marks
        self._param_names = ("_scale",)   # instance-level: x is this gate's (differentiable) angle
        self._store_param("_scale", x)
        self.diag = None if np.ndim(scale) else np.exp(-1j * marks * scale)
        kw.setdefault("name", "DiagU")
        Operation.__init__(self, wires=wl, **kw)
        return self

    def _build_matrix(self):
        if self.diag is not None:
            return np.diag(self.diag)
        ph = np.exp(-1j * self._marks[None, :] * self._scale[:, None])
        out = np.zeros(ph.shape + (ph.shape[1],), dtype=np.complex128)
        idx = np.arange(ph.shape[1])
        out[:, idx, idx] = ph
        return out

    def lower(self, n_qubits: int):
        k = len(self.wires)
        if k == n_qubits and self.wires == list(range(n_qubits)):
            if self._marks is not None:
                return "DIAG_ALL", [], [self._scale], np.asarray(self._marks, dtype=np.float64)
            return "DIAG_ALL", [], [1.0], (-np.angle(self.diag)).astype(np.float64)
        if self.diag is None:
            raise NotImplementedError("batched diagonal on a wire subset")
        return Operation.lower(self, n_qubits)


# ---- observables --------------------------------------------------------------------------
def z_parity_mask(ob: Operation) -> Optional[List[int]]:
    """Wires of a Z / Z(x)Z(x).. observable, else None (fast path of
    ``simulation.py:241-261`` generalised to parities, ``jaqsi.py:149-167``)."""
    if isinstance(ob, PauliZ):
        return list(ob.wires)
    label = getattr(ob, "_pauli_label", None)
    if label is not None and set(label) == {"Z"} and len(label) == len(ob.wires):
        return list(ob.wires)
    return None


# ---- noise channels ---------------------------------------------------------------------------
_SUPEROPERATORS: dict = {}  # KrausChannel.superoperator


class KrausChannel(Operation):
    """phi(rho) = sum_k K_k rho K_k^dagger  (``operations.py:1490-1578``).

    Channels cannot act on a pure statevector; a tape that contains one is simulated on
    the vectorised density matrix, where the channel is the dense operator
    :meth:`superoperator` on ``[wires..., wires + n_qubits...]``.
    """

    def kraus_matrices(self) -> List[np.ndarray]:
        raise NotImplementedError

    @property
    def matrix(self) -> np.ndarray:
        raise TypeError(
            f"{type(self).__name__} is a noise channel and has no single "
            "unitary matrix. Use apply_to_density() instead."
        )

    def superoperator(self) -> np.ndarray:
        """sum_k K_k (x) conj(K_k): row/col index = (ket bits of the wires, bra bits).

        A noisy model applies the same few channels after every gate (``unitary.py:150-197``): the
        matrix of a channel given by plain numbers is computed once per (class, parameters, wire
        count) and handed out read-only."""
        key = self._superoperator_key()
        if key is not None:
            hit = _SUPEROPERATORS.get(key)
            if hit is not None:
                return hit
        ks = [np.asarray(k, dtype=np.complex128) for k in self.kraus_matrices()]
        S = sum(np.kron(k, np.conj(k)) for k in ks)
        if key is not None:
            if len(_SUPEROPERATORS) >= 4096:
                _SUPEROPERATORS.clear()
            S.setflags(write=False)
            _SUPEROPERATORS[key] = S
        return S

    def _superoperator_key(self):
        names = self._param_names
        if not names:
            return None  # (QubitChannel: explicit matrices)
        vals = tuple(getattr(self, n_) for n_ in names)
        if not all(isinstance(v, float) for v in vals):
            return None
        return type(self), vals, len(self.wires)

    def lower(self, n_qubits: int):
        raise TypeError(
            f"{type(self).__name__} is a noise channel and cannot be "
            "applied to a pure statevector. Use execute(type='density') instead."
        )


def _check_unit(name: str, v: float) -> float:
    v = float(v)
    if not 0.0 <= v <= 1.0:
        raise ValueError(f"{name} must be in [0, 1].")
    return v


class BitFlip(KrausChannel):
    """K0 = sqrt(1-p) I, K1 = sqrt(p) X."""

    _num_wires = 1
    _param_names = ("p",)

    def __init__(self, p: float, wires=0) -> None:
        self.p = _check_unit("p", p)
        super().__init__(wires=wires)

    def kraus_matrices(self):
        return [np.sqrt(1 - self.p) * _I2, np.sqrt(self.p) * _X]


class PhaseFlip(KrausChannel):
    """K0 = sqrt(1-p) I, K1 = sqrt(p) Z."""

    _num_wires = 1
    _param_names = ("p",)

    def __init__(self, p: float, wires=0) -> None:
        self.p = _check_unit("p", p)
        super().__init__(wires=wires)

    def kraus_matrices(self):
        return [np.sqrt(1 - self.p) * _I2, np.sqrt(self.p) * _Z]


class DepolarizingChannel(KrausChannel):
    """K0 = sqrt(1-p) I, K1..3 = sqrt(p/3) {X, Y, Z}."""

    _num_wires = 1
    _param_names = ("p",)

    def __init__(self, p: float, wires=0) -> None:
        self.p = _check_unit("p", p)
        super().__init__(wires=wires)

    def kraus_matrices(self):
        q = np.sqrt(self.p / 3)
        return [np.sqrt(1 - self.p) * _I2, q * _X, q * _Y, q * _Z]


class AmplitudeDamping(KrausChannel):
    """K0 = diag(1, sqrt(1-g)), K1 = sqrt(g) |0><1|."""

    _num_wires = 1
    _param_names = ("gamma",)

    def __init__(self, gamma: float, wires=0) -> None:
        self.gamma = _check_unit("gamma", gamma)
        super().__init__(wires=wires)

    def kraus_matrices(self):
        g = self.gamma
        return [np.array([[1, 0], [0, np.sqrt(1 - g)]], dtype=np.complex128),
                np.array([[0, np.sqrt(g)], [0, 0]], dtype=np.complex128)]


class PhaseDamping(KrausChannel):
    """K0 = diag(1, sqrt(1-g)), K1 = diag(0, sqrt(g))."""

    _num_wires = 1
    _param_names = ("gamma",)

    def __init__(self, gamma: float, wires=0) -> None:
        self.gamma = _check_unit("gamma", gamma)
        super().__init__(wires=wires)

    def kraus_matrices(self):
        g = self.gamma
        return [np.array([[1, 0], [0, np.sqrt(1 - g)]], dtype=np.complex128),
                np.array([[0, 0], [0, np.sqrt(g)]], dtype=np.complex128)]


class ThermalRelaxationError(KrausChannel):
    """T1 relaxation + T2 dephasing over a gate time ``tg`` (``operations.py:1790-1893``):
    six reset / phase-flip operators for T2 <= T1, Choi-matrix eigen-operators otherwise."""

    _num_wires = 1
    _param_names = ("pe", "t1", "t2", "tg")

    def __init__(self, pe: float, t1: float, t2: float, tg: float, wires=0) -> None:
        if not 0.0 <= pe <= 1.0:
            raise ValueError("pe must be in [0, 1].")
        if t1 <= 0:
            raise ValueError("t1 must be > 0.")
        if t2 <= 0:
            raise ValueError("t2 must be > 0.")
        if t2 > 2 * t1:
            raise ValueError("t2 must be <= 2*t1.")
        if tg < 0:
            raise ValueError("tg must be >= 0.")
        self.pe, self.t1, self.t2, self.tg = float(pe), float(t1), float(t2), float(tg)
        super().__init__(wires=wires)

    def kraus_matrices(self):
        pe, t1, t2, tg = self.pe, self.t1, self.t2, self.tg
        e1, e2 = np.exp(-tg / t1), np.exp(-tg / t2)
        p_reset = 1.0 - e1
        if t2 <= t1:
            pz = (1.0 - p_reset) * (1.0 - e2 / e1) / 2.0
            pr0, pr1 = (1.0 - pe) * p_reset, pe * p_reset
            pid = 1.0 - pz - pr0 - pr1
            E = lambda r, c: np.array([[float((r, c) == (i, j)) for j in range(2)]  # noqa: E731
                                       for i in range(2)], dtype=np.complex128)
            return [np.sqrt(pid) * _I2, np.sqrt(pz) * _Z, np.sqrt(pr0) * E(0, 0),
                    np.sqrt(pr0) * E(0, 1), np.sqrt(pr1) * E(1, 0), np.sqrt(pr1) * E(1, 1)]
        choi = np.array([[1 - pe * p_reset, 0, 0, e2], [0, pe * p_reset, 0, 0],
                         [0, 0, (1 - pe) * p_reset, 0], [e2, 0, 0, 1 - (1 - pe) * p_reset]],
                        dtype=np.complex128)
        lam, vec = np.linalg.eigh(choi)
        return [np.sqrt(abs(lam[i])) * vec[:, i].reshape(2, 2, order="F") for i in range(4)]


class QubitChannel(KrausChannel):
    """Channel given by an explicit list of Kraus matrices (``operations.py:1896-1929``)."""

    def __init__(self, kraus_ops, wires=0) -> None:
        self._kraus_ops = [np.asarray(k, dtype=np.complex128) for k in kraus_ops]
        super().__init__(wires=wires)

    def kraus_matrices(self):
        return self._kraus_ops
