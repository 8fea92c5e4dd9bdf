"""CPU tests of the density-matrix / noise row: oracle pinned by
closed-form channel actions (the published semantics the reference checks against
PennyLane's default.mixed in ``tests/test_jaqsi.py:587-696``), front-end channel classes,
the doubled-register lowering and the Model's noise plumbing.  No GPU compute."""
import numpy as np
import pytest

from oracle import einsum_sim as ES
from oracle import noise as ON
from qml_essentials_amd import operations as op
from qml_essentials_amd import simulation
from qml_essentials_amd.model import Model
from qml_essentials_amd.tape import batch_context, recording
from qml_essentials_amd.unitary import UnitaryGates
from qml_essentials_amd.utils import key

from helpers import frontend_to_oracle, lowered_to_oracle


def rx_rho(theta):
    c, s = np.cos(theta / 2), np.sin(theta / 2)
    return np.array([[c * c, 1j * s * c], [-1j * s * c, s * s]])


# ---- oracle known answers -------------------------------------------------------------
@pytest.mark.parametrize("name,param,theta", [
    ("BitFlip", 0.15, 0.8), ("PhaseFlip", 0.2, 1.1), ("DepolarizingChannel", 0.12, 0.6),
    ("AmplitudeDamping", 0.25, 1.3), ("PhaseDamping", 0.3, 0.9)])
def test_oracle_channel_closed_forms(name, param, theta):
    """Same circuits / parameters as test_jaqsi.py:587-640; expected values in closed form."""
    rho0 = rx_rho(theta)
    rho = ON.simulate_mixed([("RX", [0], (theta,)), (name, [0], (param,))], 1)
    X = ON.X
    want = {
        "BitFlip": (1 - param) * rho0 + param * X @ rho0 @ X,
        "PhaseFlip": rho0 * np.array([[1, 1 - 2 * param], [1 - 2 * param, 1]]),
        "DepolarizingChannel": (1 - 4 * param / 3) * rho0 + (2 * param / 3) * np.eye(2),
        "AmplitudeDamping": np.array([[rho0[0, 0] + param * rho0[1, 1],
                                       np.sqrt(1 - param) * rho0[0, 1]],
                                      [np.sqrt(1 - param) * rho0[1, 0],
                                       (1 - param) * rho0[1, 1]]]),
        "PhaseDamping": rho0 * np.array([[1, np.sqrt(1 - param)], [np.sqrt(1 - param), 1]]),
    }[name]
    assert np.allclose(rho, want, atol=1e-12)


@pytest.mark.parametrize("t1,t2", [(1e-4, 5e-5), (1e-4, 1.5e-4)])
def test_oracle_thermal_relaxation(t1, t2):
    """test_jaqsi.py:642-662 circuit; T1 decay of the population, T2 decay of the coherence."""
    pe, tg, theta = 0.0, 1e-6, 1.0
    rho0 = rx_rho(theta)
    rho = ON.simulate_mixed([("RX", [0], (theta,)),
                             ("ThermalRelaxationError", [0], (pe, t1, t2, tg))], 1)
    assert np.isclose(rho[1, 1], np.exp(-tg / t1) * rho0[1, 1], atol=1e-12)
    assert np.isclose(rho[0, 1], np.exp(-tg / t2) * rho0[0, 1], atol=1e-12)
    assert np.isclose(np.trace(rho), 1.0, atol=1e-12)


def test_oracle_noisy_bell_is_valid_density():
    """test_jaqsi.py:677-696."""
    tape = [("H", [0], ()), ("CX", [0, 1], ()), ("DepolarizingChannel", [0], (0.05,)),
            ("DepolarizingChannel", [1], (0.05,))]
    rho = ON.simulate_mixed(tape, 2)
    assert np.isclose(np.trace(rho), 1.0) and np.allclose(rho, rho.conj().T)
    purity = np.real(np.trace(rho @ rho))
    assert purity < 1 - 1e-6 and np.linalg.eigvalsh(rho).min() > -1e-12
    probs = ON.measure_density(rho, 2, "probs")
    assert np.isclose(probs.sum(), 1.0)
    with pytest.raises(ValueError, match="not defined for mixed"):
        ON.measure_density(rho, 2, "state")


# ---- front-end channel classes ----------------------------------------------------------
CHANNELS = [
    (op.BitFlip, (0.15,)), (op.PhaseFlip, (0.2,)), (op.DepolarizingChannel, (0.12,)),
    (op.AmplitudeDamping, (0.25,)), (op.PhaseDamping, (0.3,)),
    (op.ThermalRelaxationError, (0.1, 1e-4, 5e-5, 1e-6)),
    (op.ThermalRelaxationError, (0.2, 1e-4, 1.5e-4, 2e-5)),
]


@pytest.mark.parametrize("cls,params", CHANNELS)
def test_channel_kraus_match_oracle_and_are_complete(cls, params):
    ch = cls(*params, wires=0)
    ks = ch.kraus_matrices()
    assert np.allclose(sum(k.conj().T @ k for k in ks), np.eye(2), atol=1e-12)
    # same channel (not necessarily the same Kraus set): compare superoperators
    want = sum(np.kron(k, k.conj()) for k in ON.kraus(cls.__name__, params))
    assert np.allclose(ch.superoperator(), want, atol=1e-12)
    with pytest.raises(TypeError, match="noise channel"):
        ch.matrix
    with pytest.raises(TypeError, match="cannot be"):
        ch.lower(1)


def test_channel_validation_messages():
    for cls in (op.BitFlip, op.PhaseFlip, op.DepolarizingChannel):
        with pytest.raises(ValueError, match=r"p must be in \[0, 1\]"):
            cls(1.5, wires=0)
    for cls in (op.AmplitudeDamping, op.PhaseDamping):
        with pytest.raises(ValueError, match=r"gamma must be in \[0, 1\]"):
            cls(-0.1, wires=0)
    with pytest.raises(ValueError, match="pe must be"):
        op.ThermalRelaxationError(2.0, 1, 1, 1)
    with pytest.raises(ValueError, match="t1 must be"):
        op.ThermalRelaxationError(0.0, 0, 1, 1)
    with pytest.raises(ValueError, match="t2 must be <="):
        op.ThermalRelaxationError(0.0, 1, 3, 1)
    with pytest.raises(ValueError, match="tg must be"):
        op.ThermalRelaxationError(0.0, 1, 1, -1)
    with pytest.raises(ValueError, match="Probability p"):
        UnitaryGates.NQubitDepolarizingChannel(1.5, [0, 1])
    with pytest.raises(ValueError, match="Number of qubits"):
        UnitaryGates.NQubitDepolarizingChannel(0.5, [0])


def test_n_qubit_depolarizing_matches_oracle():
    ch = UnitaryGates.NQubitDepolarizingChannel(0.3, [0, 1])
    ks = ch.kraus_matrices()
    assert len(ks) == 16
    assert np.allclose(sum(k.conj().T @ k for k in ks), np.eye(4), atol=1e-12)
    want = sum(np.kron(k, k.conj()) for k in ON.n_qubit_depolarizing_kraus(0.3, 2))
    assert np.allclose(ch.superoperator(), want)


# ---- doubled-register lowering -----------------------------------------------------------
def _noisy_tape(rng):
    with recording() as tape:
        op.H(wires=0)
        op.RX(rng.normal(), wires=1)
        op.BitFlip(0.1, wires=1)
        op.RY(rng.normal(), wires=2)
        op.Rot(*rng.normal(size=3), wires=0)
        op.PauliY(wires=2)
        op.S(wires=1)
        op.CX(wires=[0, 1])
        UnitaryGates.NQubitDepolarizingChannel(0.2, [2, 0])   # descending wires
        op.CRX(rng.normal(), wires=[1, 2])
        op.CRY(rng.normal(), wires=[2, 0])
        op.CRZ(rng.normal(), wires=[0, 2])
        op.ControlledPhaseShift(rng.normal(), wires=[1, 0])
        op.AmplitudeDamping(0.3, wires=0)
        op.CY(wires=[2, 1])
        op.RXX(rng.normal(), wires=[0, 1])
        op.RYY(rng.normal(), wires=[1, 2])
        op.RZZ(rng.normal(), wires=[0, 2])
        op.RZX(rng.normal(), wires=[2, 1])
        op.ThermalRelaxationError(0.1, 1.0, 1.5, 0.3, wires=2)
        op.SWAP(wires=[0, 2])
        op.CCX(wires=[0, 1, 2])
        op.CSWAP(wires=[2, 0, 1])
        op.PhaseDamping(0.2, wires=1)
        op.RZ(rng.normal(), wires=1)
        op.DepolarizingChannel(0.05, wires=2)
        UnitaryGates.GolombEncoding(rng.normal(), wires=[0, 1, 2])
        op.Operation(wires=[1, 2], matrix=np.linalg.qr(rng.normal(size=(4, 4))
                                                       + 1j * rng.normal(size=(4, 4)))[0])
        op.PhaseFlip(0.15, wires=0)
    return tape


def test_doubled_tape_reproduces_oracle_density():
    """U (x) conj(U) and the Kraus superoperators on the 2n-wire register give exactly the
    reference's rho -> U rho U^+ / sum K rho K^+ (checked with the oracle's pure engine)."""
    rng = np.random.default_rng(11)
    n = 3
    tape = _noisy_tape(rng)
    assert simulation.uses_density(tape, "expval")
    want = ON.simulate_mixed(frontend_to_oracle(tape), n)
    doubled = simulation.doubled_tape(tape, n)
    vec = ES.simulate_pure(lowered_to_oracle(doubled, 2 * n), 2 * n, dtype=np.complex128)
    assert np.allclose(vec.reshape(2**n, 2**n), want, atol=1e-10)
    assert np.isclose(np.trace(want), 1.0)


def test_doubled_tape_multiplies_channels_that_share_their_wires():
    """The channels a noisy model stacks behind every gate become ONE superoperator per wire tuple (their
    product, later channel on the left), emitted before the next operation that touches one of the wires;
    channels on other wires and gates on other wires pass in between.  Same density as the oracle's
    channel-by-channel evolution; equal channels share one cached matrix."""
    n = 3
    with recording() as tape:
        op.RX(0.3, wires=0)
        op.BitFlip(0.1, wires=0)
        op.PhaseFlip(0.2, wires=1)          # another wire: does not end wire 0's run
        op.AmplitudeDamping(0.3, wires=0)
        op.RY(0.7, wires=2)                  # a gate on another wire does not either
        op.DepolarizingChannel(0.05, wires=0)
        op.PhaseDamping(0.25, wires=1)
        op.CX(wires=[0, 1])                  # ends both runs
        UnitaryGates.NQubitDepolarizingChannel(0.2, [0, 1])
        UnitaryGates.NQubitDepolarizingChannel(0.1, [0, 1])
        op.BitFlip(0.1, wires=1)             # shares a wire with the pair: the pair is emitted first
        UnitaryGates.NQubitDepolarizingChannel(0.3, [1, 0])  # other wire order: a run of its own
        op.ThermalRelaxationError(0.1, 1.0, 1.5, 0.3, wires=2)
    doubled = simulation.doubled_tape(tape, n)
    names = [d.lower(2 * n)[0] for d in doubled]
    wires = [tuple(d.lower(2 * n)[1]) for d in doubled]
    assert names.count("MAT2") == 4 and names.count("MAT4") == 2, names
    assert wires[names.index("MAT2")] in ((0, 3), (1, 4))
    i_cx = names.index("CX")
    assert sorted(wires[i] for i in range(i_cx) if names[i] == "MAT2") == [(0, 3), (1, 4)]
    assert [wires[i] for i in range(len(names)) if names[i] == "MAT4"] == [(0, 1, 3, 4), (1, 0, 4, 3)]
    want = ON.simulate_mixed(frontend_to_oracle(tape), n)
    vec = ES.simulate_pure(lowered_to_oracle(doubled, 2 * n), 2 * n, dtype=np.complex128)
    assert np.allclose(vec.reshape(2**n, 2**n), want, atol=1e-12)
    a, b = op.BitFlip(0.1, wires=0).superoperator(), op.BitFlip(0.1, wires=2).superoperator()
    assert a is b and not a.flags.writeable
    assert op.BitFlip(0.11, wires=0).superoperator() is not a


def test_noise_free_tape_stays_pure():
    with recording() as tape:
        op.H(wires=0)
        op.CX(wires=[0, 1])
    assert not simulation.uses_density(tape, "expval")
    assert simulation.uses_density(tape, "density")


def test_wide_channels_become_kraus_sums():
    """3-wire channels (test_ansaetze.py:152-176) are applied operator by operator; the
    padded 4-wire operators reproduce sum_K K rho K^+."""
    n = 3
    with recording() as tape:
        op.H(wires=1)
        UnitaryGates.NQubitDepolarizingChannel(0.4, [2, 0, 1])
    items = simulation.doubled_tape(tape, n)
    assert isinstance(items[-1], simulation._WideChannel) and len(items[-1].kraus) == 64
    head = lowered_to_oracle(items[:-1], 2 * n)
    acc = 0
    for pair in items[-1].kraus_plans(n):
        acc = acc + ES.simulate_pure(head + lowered_to_oracle(pair, 2 * n), 2 * n,
                                     dtype=np.complex128)
    want = ON.simulate_mixed(frontend_to_oracle(tape), n)
    assert np.allclose(acc.reshape(8, 8), want, atol=1e-12)
    with recording() as tape:
        UnitaryGates.NQubitDepolarizingChannel(0.1, [0, 1, 2, 3, 4])
    with pytest.raises(NotImplementedError, match="more than 4 wires"):
        simulation.doubled_tape(tape, 5)


# ---- UnitaryGates.Noise / GateError ---------------------------------------------------
def test_noise_channels_follow_each_gate():
    """unitary.py:150-197: BitFlip, PhaseFlip, Depolarizing per wire, then the multi-qubit
    depolarizing channel after a two-wire gate."""
    noise = {"BitFlip": 0.1, "PhaseFlip": 0.2, "Depolarizing": 0.3,
             "MultiQubitDepolarizing": 0.05}
    with recording() as tape:
        UnitaryGates.RX(0.3, wires=0, noise_params=noise)
        UnitaryGates.CX(wires=[0, 1], noise_params=noise)
    names = [(o.name, o.wires) for o in tape]
    assert names == [
        ("RX", [0]), ("BitFlip", [0]), ("PhaseFlip", [0]), ("DepolarizingChannel", [0]),
        ("CX", [0, 1]), ("BitFlip", [0]), ("PhaseFlip", [0]), ("DepolarizingChannel", [0]),
        ("BitFlip", [1]), ("PhaseFlip", [1]), ("DepolarizingChannel", [1]),
        ("QubitChannel", [0, 1])]
    with recording() as tape:
        UnitaryGates.RX(0.3, wires=0, noise_params={"BitFlip": 0.0})
    assert [o.name for o in tape] == ["RX"]


def test_gate_error_draws():
    noise = {"GateError": 0.1}
    with pytest.raises(AssertionError, match="random_key must be provided"):
        UnitaryGates.GateError(0.3, noise, None)
    w, k2 = UnitaryGates.GateError(0.3, noise, key(3))
    w_again, _ = UnitaryGates.GateError(0.3, noise, key(3))
    assert w == w_again and w != 0.3 and abs(w - 0.3) < 1.0 and k2 is not None
    assert UnitaryGates.GateError(0.3, None, None) == (0.3, None)
    # one draw per batch element while recording a batch
    with batch_context(64):
        wb, _ = UnitaryGates.GateError(0.3, noise, key(3))
    col = np.asarray(wb.data)
    assert col.shape == (64,) and np.std(col) > 0.03 and abs(np.mean(col) - 0.3) < 0.05
    # batch_gate_error = False: one shared draw from a fixed key
    UnitaryGates.batch_gate_error = False
    try:
        with batch_context(8):
            w1, k_same = UnitaryGates.GateError(0.3, noise, key(3))
            w2, _ = UnitaryGates.GateError(0.3, noise, key(99))
        assert np.ndim(w1) == 0 and w1 == w2
    finally:
        UnitaryGates.batch_gate_error = True


# ---- Model plumbing ---------------------------------------------------------------------
def _record(model, noise):
    model.noise_params = noise
    with recording() as tape:
        model._variational(model.params[0], np.zeros(1), random_key=key(0),
                           noise_params=model.noise_params)
    return tape


def test_model_noise_placement_matches_oracle_placement():
    """model.py:1000-1064 + unitary.py:150-197 placement == oracle.with_gate_noise."""
    model = Model(n_qubits=3, n_layers=1, circuit_type="Hardware_Efficient",
                  remove_zero_encoding=False)
    noise = {"BitFlip": 0.01, "Depolarizing": 0.02, "MultiQubitDepolarizing": 0.03,
             "StatePreparation": 0.04, "AmplitudeDamping": 0.05, "PhaseDamping": 0.06,
             "Measurement": 0.07}
    clean = [t for t in frontend_to_oracle(_record(model, None))]
    noisy = frontend_to_oracle(_record(model, dict(noise)))
    want = ON.with_gate_noise(clean, noise)
    assert len(noisy) == len(want)
    rho_a = ON.simulate_mixed(noisy, 3)
    rho_b = ON.simulate_mixed(
        [(("QubitChannel", w, (ON.kraus(nm, p),)) if nm == "NQubitDepolarizing" else (nm, w, p))
         for nm, w, p in want], 3)
    assert np.allclose(rho_a, rho_b, atol=1e-12)
    assert [t[0] for t in noisy[:3]] == ["BitFlip"] * 3          # state preparation first
    assert noisy[-1][0] == "BitFlip" and noisy[-3][0] == "AmplitudeDamping"


def test_model_requires_density_and_depth():
    model = Model(n_qubits=2, n_layers=1, circuit_type="Circuit_19")
    assert not model._requires_density()
    model.noise_params = {"GateError": 0.1}
    assert not model._requires_density()
    model.noise_params = {"BitFlip": 0.1}
    assert model._requires_density()
    model.noise_params = None
    model.execution_type = "density"
    assert model._requires_density()
    model.execution_type = "expval"
    # depth: RX, RZ on each wire (2), then the CRX ring of 2 -> 2 more; encoding RX; again
    depth = model._get_circuit_depth()
    assert depth == model._cached_circuit_depth and depth >= 5
    model.noise_params = {"ThermalRelaxation": {"t1": 2000.0, "t2": 1000.0, "t_factor": 1.0}}
    with recording() as tape:
        model._variational(model.params[0], np.zeros(1), random_key=key(0),
                           noise_params=model.noise_params)
    th = [o for o in tape if isinstance(o, op.ThermalRelaxationError)]
    assert len(th) == 2 and th[0].tg == depth and th[0].pe == 1.0
    assert [o.wires for o in th] == [[0], [1]]


def test_all_zero_noise_is_none_and_unknown_key_warns():
    model = Model(n_qubits=2, n_layers=1, circuit_type="Circuit_19")
    model.noise_params = {"BitFlip": 0.0, "PhaseFlip": 0.0}
    assert model.noise_params is None
    with pytest.warns(UserWarning, match="not supported"):
        model.noise_params = {"BitFlip": 0.1, "Bogus": 0.2}


def test_doubled_tape_carries_the_tangents_of_its_gates():
    """What a compiled noisy call builds its angle map from: U keeps the source gate's tangent terms,
    conj(U) the same terms with the sign of the parameters that conjugation negates."""
    from qml_essentials_amd.batching import Batched

    leaf = Batched.leaf(np.array([[0.3, 0.7, 1.1], [0.5, 0.9, 1.7]]), 0)
    with recording() as tape:
        op.RX(leaf[0], wires=0)
        op.RY(2.0 * leaf[1], wires=1)
        op.BitFlip(0.1, wires=0)
        op.Rot(leaf[0], leaf[1], leaf[2], wires=1)
        op.PauliY(wires=0)
    items = simulation.doubled_tape(tape, 2)
    by = [(it.name, tuple(it.lower(4)[1]), it.parameter_tangents) for it in items]
    coef = lambda t: [[(lid, flat, float(np.asarray(c).reshape(-1)[0])) for lid, flat, c in p] for p in t]
    assert by[0][:2] == ("RX", (0,)) and coef(by[0][2]) == [[(0, 0, 1.0)]]
    assert by[1][:2] == ("RX", (2,)) and coef(by[1][2]) == [[(0, 0, -1.0)]]
    assert by[2][:2] == ("RY", (1,)) and coef(by[2][2]) == [[(0, 1, 2.0)]]
    assert by[3][:2] == ("RY", (3,)) and coef(by[3][2]) == [[(0, 1, 2.0)]]       # RY is real
    rot = [b for b in by if b[0] == "Rot"]
    assert coef(rot[0][2]) == [[(0, 0, 1.0)], [(0, 1, 1.0)], [(0, 2, 1.0)]]
    assert coef(rot[1][2]) == [[(0, 0, -1.0)], [(0, 1, 1.0)], [(0, 2, -1.0)]]   # phi, omega negated
    assert all(b[2] == [] for b in by if b[0].startswith("MAT"))
