"""Test helpers: oracle-tape <-> native op-list conversion (test infrastructure)."""
import numpy as np

from oracle import gates as G

NAME_MAP = {"CPhase": "CPhase", "Matrix": None, "DiagU": None}


def tape_to_native(tape, n_qubits):
    """[(name, wires, params)] -> (ops, angle_row, consts) for qml_essentials_amd._native.Plan.

    Every float parameter gets its own angle slot; matrices / diagonals go to the
    const blob."""
    ops, angles, consts = [], [], []
    for name, wires, params in tape:
        if name == "Barrier":
            continue
        if name == "Matrix":
            m = np.asarray(params[0], dtype=np.complex64)
            off = len(consts)
            consts.extend(np.stack([m.real, m.imag], axis=-1).reshape(-1).tolist())
            ops.append(("MAT1" if m.shape[0] == 2 else "MAT2", wires, [], off))
        elif name == "DiagU":
            raise ValueError("use golomb_op() for diagonal encodings")
        elif name == "Golomb":  # params = (x,)
            off = len(consts)
            consts.extend([float(v) for v in G.golomb_ruler(2**n_qubits)])
            angles.append(float(params[0]))
            ops.append(("DIAG_ALL", [], [len(angles) - 1], off))
        else:
            slots = []
            for p in params:
                angles.append(float(p))
                slots.append(len(angles) - 1)
            ops.append((name, wires, slots, -1))
    return ops, np.array(angles, dtype=np.float32), np.array(consts, dtype=np.float32)


def oracle_tape(tape, n_qubits):
    """Replace test-only 'Golomb' entries by the oracle's DiagU entry."""
    out = []
    for name, wires, params in tape:
        if name == "Golomb":
            out.append(("DiagU", list(range(n_qubits)), (G.golomb_diag(params[0], n_qubits),)))
        else:
            out.append((name, wires, params))
    return out


ONE_Q = ["RX", "RY", "RZ", "H", "PauliX", "PauliY", "PauliZ", "S", "Rot"]
TWO_Q = ["CX", "CY", "CZ", "CRX", "CRY", "CRZ", "CPhase", "SWAP", "RXX", "RYY", "RZZ", "RZX"]
THREE_Q = ["CCX", "CSWAP"]
N_PARAMS = {"RX": 1, "RY": 1, "RZ": 1, "Rot": 3, "CRX": 1, "CRY": 1, "CRZ": 1, "CPhase": 1,
            "RXX": 1, "RYY": 1, "RZZ": 1, "RZX": 1}


def random_tape(n, n_gates, rng, three_q=True):
    tape = []
    for _ in range(n_gates):
        r = rng.random()
        if n >= 3 and three_q and r < 0.08:
            name = THREE_Q[rng.integers(len(THREE_Q))]
            wires = [int(x) for x in rng.choice(n, 3, replace=False)]
        elif n >= 2 and r < 0.5:
            name = TWO_Q[rng.integers(len(TWO_Q))]
            wires = [int(x) for x in rng.choice(n, 2, replace=False)]
        else:
            name = ONE_Q[rng.integers(len(ONE_Q))]
            wires = [int(rng.integers(n))]
        params = tuple(float(x) for x in rng.uniform(0, 2 * np.pi, N_PARAMS.get(name, 0)))
        tape.append((name, wires, params))
    return tape


def lowered_to_oracle(lowered, n_qubits, sample=0):
    """Engine-level op tuples ``(name, wires, params, blob)`` (what ``LoweredTape`` consumes,
    e.g. from ``simulation.doubled_tape``) -> oracle tape on the same register, for one
    batch element."""
    out = []
    for low in lowered:
        name, wires, params, blob = low.lower(n_qubits)
        vals = [float(np.asarray(p, dtype=np.float64).reshape(-1)[sample]
                      if np.ndim(p) else p) for p in params]
        if name in ("MAT1", "MAT2", "MAT4"):
            b = np.asarray(blob, dtype=np.float64).reshape(-1, 2)
            d = 2 ** len(wires)
            out.append(("Matrix", list(wires), ((b[:, 0] + 1j * b[:, 1]).reshape(d, d),)))
        elif name == "DIAG_ALL":
            out.append(("DiagU", list(range(n_qubits)),
                        (np.exp(-1j * np.asarray(blob, dtype=np.float64) * vals[0]),)))
        else:
            out.append((name, list(wires), tuple(vals)))
    return out


def frontend_to_oracle(tape, sample=0):
    """Front-end ``Operation`` objects (gates and noise channels) -> oracle tape entries."""
    from qml_essentials_amd import operations as op

    out = []
    for o in tape:
        if isinstance(o, op.Barrier):
            continue
        if isinstance(o, op.QubitChannel):
            out.append(("QubitChannel", list(o.wires), (o.kraus_matrices(),)))
        elif isinstance(o, op.KrausChannel):
            out.append((type(o).__name__, list(o.wires),
                        tuple(getattr(o, k) for k in o._param_names)))
        elif isinstance(o, op.DiagonalQubitUnitary):
            m = np.asarray(o.matrix)
            m = m[sample] if m.ndim == 3 else m
            out.append(("DiagU", list(o.wires), (np.diag(m),)))
        else:
            low = o.lower(max(o.wires) + 1)
            name, _, params, blob = low
            if name in ("MAT1", "MAT2", "MAT4"):
                out.append(("Matrix", list(o.wires), (np.asarray(o.matrix),)))
            else:
                out.append((name, list(o.wires), tuple(
                    float(p if np.ndim(p) == 0 else np.asarray(p)[sample]) for p in params)))
    return out
