"""GPU: the complex128 engine (``qmle_run_batch_f64``; ``utils.enable_x64`` / ``Model(x64=True)``) --
the reference's ``jax_enable_x64`` mode (``operations.py:12-16``, switched on by
``tests/test_coefficients.py:19``).  Parity bar: 1e-10 against the complex128 oracle."""
import numpy as np
import pytest

from oracle import circuits as OC
from oracle import einsum_sim as OE
from tests.helpers import random_tape, tape_to_native

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _plan_and_angles(tape, n):
    from qml_essentials_amd import _native as N

    ops, angles32, consts = tape_to_native(tape, n)
    # (tape_to_native rounds the angle row to float32; the complex128 run takes the tape's own doubles)
    angles = np.array([float(p) for name, _, params in tape if name != "Barrier" for p in params], dtype=np.float64)
    assert angles.shape == angles32.shape
    plan = N.Plan(ops, n, len(angles), consts, 0)
    ang = torch.from_numpy(np.ascontiguousarray(angles, dtype=np.float64)[None, :]).cuda()
    if ang.shape[1] == 0:
        ang = torch.zeros((1, 1), dtype=torch.float64, device="cuda")[:, :0]
    return plan, ang


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 11, 13, 14, 16])
def test_random_circuits_state_probs_expval_vs_complex128_oracle(n):
    """Every gate kind of the random tape generator; LDS regime (n <= 13) and the streaming
    regime (14, 16)."""
    rng = np.random.default_rng(640 + n)
    tape = [g for g in random_tape(n, 40 if n > 1 else 10, rng) if g[0] not in ("MAT1", "MAT2")]
    want = OE.simulate_pure(tape, n, np.complex128)
    plan, ang = _plan_and_angles(tape, n)
    got = plan.run64(ang, "state").cpu().numpy()[0]
    assert got.dtype == np.complex128
    assert np.abs(got - want).max() < 1e-12
    probs = plan.run64(ang, "probs").cpu().numpy()[0]
    assert probs.dtype == np.float64 and np.abs(probs - np.abs(want) ** 2).max() < 1e-12
    groups = [[q] for q in range(n)] + ([[0, n - 1], list(range(min(n, 3)))] if n > 1 else [])
    ez = plan.run64(ang, "expval", groups).cpu().numpy()[0]
    idx = np.arange(2**n)
    p = np.abs(want) ** 2
    for k, g in enumerate(groups):
        sign = np.ones(2**n)
        for w in g:
            sign *= 1 - 2 * ((idx >> (n - 1 - w)) & 1)
        assert abs(ez[k] - np.dot(p, sign)) < 1e-12, (n, g)
    if n <= 6:
        rho = plan.run64(ang, "density").cpu().numpy()[0]
        assert np.abs(rho - np.outer(want, want.conj())).max() < 1e-12


def test_batch_rows_and_float32_engine_differ_at_float32_level_only():
    """Same plan object, both engines: the complex64 result sits 1e-7 from the complex128 one."""
    from qml_essentials_amd import _native as N
    from tests.test_abi_cpu import he_layer_ops

    n = 10
    ops, slots = he_layer_ops(n)
    ang64 = np.random.default_rng(7).uniform(0, 2 * np.pi, (5, slots))
    plan = N.Plan(ops, n, slots)
    e64 = plan.run64(torch.from_numpy(ang64).cuda(), "expval", [[q] for q in range(n)]).cpu().numpy()
    e32 = plan.run(torch.from_numpy(ang64.astype(np.float32)).cuda(), "expval", list(range(n))).cpu().numpy()
    assert e64.dtype == np.float64 and e32.dtype == np.float32
    assert 0 < np.abs(e64 - e32).max() < 5e-6
    spec = OC.ModelSpec(n, 1, "Hardware_Efficient", data_reupload=False)
    for b in (0, 4):
        tape = [("RY", [q], (ang64[b, q],)) for q in range(n)]  # he_layer_ops: RY, RZ, RY per wire, then the CX ring
        tape = []
        for i, g in enumerate(("RY", "RZ", "RY")):
            tape += [(g, [q], (ang64[b, i * n + q],)) for q in range(n)]
        tape += [("CX", list(w), ()) for w in OC.bricks(n, mirror=False) +
                 OC.bricks(n, offset=-1, modulo=True, wrap=True, mirror=False)]
        want = OE.simulate_and_measure(tape, n, "expval", [("PauliZ", [q]) for q in range(n)], np.complex128)
        assert np.abs(e64[b] - want).max() < 1e-12


def test_model_x64_switch_and_scope():
    """``Model(..., x64=True)`` and the global ``utils.enable_x64``: float64 results equal to the
    complex128 oracle to 1e-10; the default stays complex64."""
    from qml_essentials_amd import utils
    from qml_essentials_amd.model import Model

    rng = np.random.default_rng(11)
    m64 = Model(5, 2, "Circuit_19", x64=True)
    m32 = Model(5, 2, "Circuit_19")
    P = rng.uniform(0, 2 * np.pi, (3, *m64.params.shape[1:]))
    x = np.array([0.3, 1.1])
    out64 = np.asarray(m64(params=P, inputs=x))
    out32 = np.asarray(m32(params=P, inputs=x))
    assert out64.dtype == np.float64 and out32.dtype == np.float32 and out64.shape == out32.shape
    spec = OC.ModelSpec(5, 2, "Circuit_19")
    for i, xi in enumerate(x):
        for j in range(3):
            want = OE.simulate_and_measure(OC.model_tape(spec, P[j], [xi], zero_inputs_batch1=False), 5, "expval",
                                           [("PauliZ", [q]) for q in range(5)], np.complex128)
            assert np.abs(out64[i, j] - want).max() < 1e-10
    assert np.abs(out64 - out32).max() < 5e-6
    assert not utils.x64_enabled()
    utils.enable_x64()
    try:
        assert np.asarray(m32(params=P, inputs=x)).dtype == np.float64
        st = np.asarray(m32(params=P[0], inputs=x[:1], execution_type="state"))
        assert st.dtype == np.complex128
    finally:
        utils.enable_x64(False)
    # (execution_type is sticky, as in the reference: model.py:1584-1585)
    assert np.asarray(m32(params=P, inputs=x, execution_type="expval")).dtype == np.float32


def test_general_observables_in_x64():
    """A non-diagonal observable (PauliX, two-wire Hermitian matrix) on the complex128 state."""
    from qml_essentials_amd import operations as op
    from qml_essentials_amd import utils
    from qml_essentials_amd.script import Script

    def circuit(theta):
        op.RY(theta, wires=0)
        op.CX(wires=[0, 1])
        op.RX(0.3, wires=2)

    th = np.linspace(0.1, 2.9, 4)
    utils.enable_x64()
    try:
        res = Script(circuit, n_qubits=3).execute(
            type="expval", obs=[op.PauliX(0, record=False), op.PauliZ(1, record=False), op.PauliY(2, record=False)],
            args=(th,), in_axes=(0,))
    finally:
        utils.enable_x64(False)
    assert res.dtype == np.float64 and res.shape == (4, 3)
    # <X_0> = 0 (entangled with wire 1), <Z_1> = cos(theta), <Y_2> = -sin(0.3)
    assert np.abs(res[:, 0]).max() < 1e-12
    assert np.abs(res[:, 1] - np.cos(th)).max() < 1e-12
    assert np.abs(res[:, 2] + np.sin(0.3)).max() < 1e-12


@pytest.mark.parametrize("circuit_type,n_qubits,n_layers,output_qubit", [
    ("Circuit_1", 3, 1, [0, 1]), ("Circuit_9", 4, 1, 0), ("Circuit_19", 5, 1, 0),
    ("Hardware_Efficient", 4, 2, -1)])
def test_fourier_series_reproduces_the_model_in_x64(circuit_type, n_qubits, n_layers, output_qubit):
    """`tests/test_coefficients.py:25-70` the way the reference runs it -- that module switches
    `jax_enable_x64` on (`:19`): spectrum and re-evaluated Fourier series in float64 / complex128,
    equal to the model to 1e-10 (1e-5 is all the complex64 engine can promise)."""
    from qml_essentials_amd.coefficients import Coefficients
    from qml_essentials_amd.model import Model
    from qml_essentials_amd.utils import x64_scope

    with x64_scope(True):
        model = Model(n_qubits=n_qubits, n_layers=n_layers, circuit_type=circuit_type,
                      output_qubit=output_qubit)
        coeffs, freqs = Coefficients.get_spectrum(model)
        assert coeffs.shape == model.degree and coeffs.dtype == np.complex128
        ref = np.linspace(-np.pi, np.pi, 10)
        exp_model = np.asarray(model(params=None, inputs=ref, force_mean=True))
        assert exp_model.dtype == np.float64
        exp_fourier = Coefficients.evaluate_Fourier_series(coefficients=coeffs, frequencies=freqs, inputs=ref)
        assert np.allclose(exp_model, exp_fourier, atol=1e-10), np.abs(exp_model - exp_fourier).max()
    # and the complex64 engine on the same model agrees with the float64 values at its own level
    exp32 = np.asarray(model(params=None, inputs=ref, force_mean=True))
    assert exp32.dtype == np.float32 and np.allclose(exp32, exp_model, atol=2e-6)


def test_parameter_shift_gradient_in_x64():
    """`Model.gradient` under `x64_scope`: the shifted circuits run on the complex128 engine --
    central differences of the float64 model (`tests/test_jaqsi.py:131-141`'s jax.grad case runs
    with x64 on) agree to 1e-9, where the complex64 engine stops at ~1e-7."""
    from qml_essentials_amd.model import Model
    from qml_essentials_amd.utils import x64_scope

    m = Model(4, 2, "Hardware_Efficient")
    x = np.array([0.3])
    g32 = np.asarray(m.gradient(inputs=x, force_mean=True))
    with x64_scope(True):
        g64 = np.asarray(m.gradient(inputs=x, force_mean=True))
        gi = np.asarray(m.gradient(inputs=x, wrt="inputs", force_mean=True))
        p = np.asarray(m.params, dtype=np.float64).copy()
        f = lambda q, xx=x: float(np.asarray(m(params=q, inputs=xx, force_mean=True)))
        eps = 1e-5
        fd = np.zeros_like(p)
        for i in np.ndindex(*p.shape):
            a, b = p.copy(), p.copy()
            a[i] += eps
            b[i] -= eps
            fd[i] = (f(a) - f(b)) / (2 * eps)
        fdi = (f(p, x + eps) - f(p, x - eps)) / (2 * eps)
    assert np.abs(g64.reshape(fd.shape) - fd).max() < 1e-9
    assert abs(float(gi.reshape(-1)[0]) - fdi) < 1e-9
    assert 0 < np.abs(g32 - g64).max() < 1e-5


def test_adjoint_method_and_model_scope_in_x64():
    """`method="adjoint"` under x64 (round 5: ONE complex128 backward sweep, `qmle_adjoint_gradient_f64`,
    where rounds 3-4 contracted the parameter-shift Jacobian of 2 P shifted circuits): equal to that
    Jacobian contracted by hand at 1e-13 and to the complex64 adjoint at float32 level;
    `Model(x64=True).gradient` scopes the mode itself."""
    from qml_essentials_amd.model import Model
    from qml_essentials_amd.utils import x64_enabled, x64_scope

    m = Model(4, 2, "Hardware_Efficient")
    x = np.array([0.1, -0.4, 0.9])
    rng = np.random.default_rng(3)
    ct = rng.normal(size=(3, 4))
    a32 = np.asarray(m.gradient(inputs=x, method="adjoint", cotangent=ct))
    with x64_scope(True):
        J = np.asarray(m.gradient(inputs=x))                       # (3, 4, *params)
        a64 = np.asarray(m.gradient(inputs=x, method="adjoint", cotangent=ct))
        mean64 = np.asarray(m.gradient(inputs=x, method="adjoint", force_mean=True))
    assert a64.shape == a32.shape and np.abs(a64 - np.einsum("bk,bk...->b...", ct, J)).max() < 1e-13
    assert np.abs(mean64 - J.mean(axis=1)).max() < 1e-13
    assert 0 < np.abs(a64 - a32).max() < 2e-5
    mx = Model(4, 2, "Hardware_Efficient", x64=True)
    assert not x64_enabled()
    gx = np.asarray(mx.gradient(inputs=x))
    assert np.abs(gx - J).max() < 1e-13 and not x64_enabled()


def test_shots_in_x64_draw_from_the_complex128_probabilities():
    """Shot estimates under x64 (pure and noisy): the sampler takes the float64 probabilities
    rounded once to float32 -- same Philox streams, so the counts equal the complex64 run's except
    where a CDF boundary moved by ~1e-8 (at most a few shots in 10^5)."""
    from qml_essentials_amd import operations as op
    from qml_essentials_amd.script import Script
    from qml_essentials_amd.utils import key, x64_scope

    def circuit(t):
        op.RX(t, wires=0)
        op.RY(0.4, wires=1)
        op.CX(wires=[0, 1])
        op.RZ(0.3, wires=2)
        op.H(wires=2)

    def noisy(t):
        circuit(t)
        op.BitFlip(0.1, wires=1)

    shots = 100_000
    for f in (circuit, noisy):
        s = Script(f=f, n_qubits=3)
        a = s.execute(type="probs", args=(np.array(0.7),), shots=shots, key=key(5))
        with x64_scope(True):
            b = s.execute(type="probs", args=(np.array(0.7),), shots=shots, key=key(5))
            exact = s.execute(type="probs", args=(np.array(0.7),))
        assert exact.dtype == np.float64
        assert np.abs(np.asarray(a) - np.asarray(b)).sum() * shots <= 8  # a few boundary shots
        assert np.abs(np.asarray(b) - exact).max() < 5 / np.sqrt(shots)


def test_x64_calls_chunk_instead_of_running_out_of_memory(monkeypatch):
    """the batch chunker sized x64 calls with the complex64 model.  With free HBM
    'shrunk' to 40 MiB a complex128 batch of 64 x 2^15 amplitudes (32 MiB of states + as much again
    for a general observable) must go through execute_chunked -- and give the unchunked numbers."""
    from qml_essentials_amd import memory, utils
    from qml_essentials_amd import operations as op
    from qml_essentials_amd.script import Script

    n = 15

    def circuit(theta):
        for q in range(n):
            op.RY(theta * (q + 1) / n, wires=q)
        for q in range(n - 1):
            op.CX(wires=[q, q + 1])

    th = np.linspace(0.1, 2.0, 64)
    obs_z = [op.PauliZ(q, record=False) for q in (0, 7, 14)]
    obs_g = [op.PauliX(3, record=False)]
    with utils.x64_scope(True):
        full_state = Script(circuit, n).execute(type="state", args=(th,), in_axes=(0,))
        full_z = Script(circuit, n).execute(type="expval", obs=obs_z, args=(th,), in_axes=(0,))
        full_g = Script(circuit, n).execute(type="expval", obs=obs_g, args=(th,), in_axes=(0,))
        calls = []
        real = memory.execute_chunked
        monkeypatch.setattr(memory, "execute_chunked", lambda run, b, c: (calls.append((b, c)), real(run, b, c))[1])
        monkeypatch.setattr(memory, "available_memory_bytes", lambda: 40 << 20)
        got_state = Script(circuit, n).execute(type="state", args=(th,), in_axes=(0,))
        got_z = Script(circuit, n).execute(type="expval", obs=obs_z, args=(th,), in_axes=(0,))
        got_g = Script(circuit, n).execute(type="expval", obs=obs_g, args=(th,), in_axes=(0,))
    assert len(calls) == 3 and all(c < b for b, c in calls), calls
    assert calls[2][1] < calls[0][1] or calls[2][1] <= calls[1][1]   # the general observable keeps the states
    assert np.asarray(got_state).dtype == np.complex128
    assert np.array_equal(np.asarray(got_state), np.asarray(full_state))
    assert np.array_equal(np.asarray(got_z), np.asarray(full_z))
    assert np.allclose(np.asarray(got_g), np.asarray(full_g), atol=1e-12)
    # and the float32 model would have let the state call through whole: 64 x 2^15 x 8 B = 16 MiB < 32 MiB
    assert memory.compute_chunk_size(n, 64, "state", False) == 64


def test_mixed_observables_and_non_z_gradients_in_x64():
    """a mixed Z / non-Z observable list reads every value off the ONE complex128 state it
    holds (no re-run of the circuit per Z observable), and Script.gradient with a PauliX observable works
    in x64 mode (it raised NotImplementedError): both against central differences / separate calls."""
    from qml_essentials_amd import operations as op, utils
    from qml_essentials_amd.script import Script

    def circuit(a, b):
        op.RY(a, wires=0)
        op.RX(b, wires=1)
        op.CX(wires=[0, 1])
        op.RZ(a * b, wires=1)
        op.H(wires=2)
        op.CRX(b, wires=[2, 0])

    obs = [op.PauliZ(0, record=False), op.PauliX(1, record=False), op.PauliZ(2, record=False), op.PauliY(0, record=False)]
    a, b = 0.7, 1.3
    with utils.x64_scope(True):
        sc = Script(circuit, 3)
        mixed = np.asarray(sc.execute(type="expval", obs=obs, args=(a, b)))
        single = np.array([np.asarray(sc.execute(type="expval", obs=[o], args=(a, b)))[0] for o in obs])
        assert mixed.dtype == np.float64 and np.abs(mixed - single).max() < 1e-14
        (ga, gb) = sc.gradient(obs, args=(np.float64(a), np.float64(b)), argnums=(0, 1))
        h = 1e-6
        fa = (np.asarray(sc.execute(type="expval", obs=obs, args=(a + h, b)))
              - np.asarray(sc.execute(type="expval", obs=obs, args=(a - h, b)))) / (2 * h)
        fb = (np.asarray(sc.execute(type="expval", obs=obs, args=(a, b + h)))
              - np.asarray(sc.execute(type="expval", obs=obs, args=(a, b - h)))) / (2 * h)
    assert np.abs(np.asarray(ga).reshape(-1) - fa).max() < 1e-8
    assert np.abs(np.asarray(gb).reshape(-1) - fb).max() < 1e-8


def _d1(g, eps=1e-3):
    """Fourth-order central difference of the scalar function g at 0: truncation eps^4 / 30 g^(5) ~ 1e-13."""
    return (8.0 * (g(eps) - g(-eps)) - (g(2 * eps) - g(-2 * eps))) / (12.0 * eps)


def _central_differences(f, p):
    fd = np.zeros_like(p)
    for i in np.ndindex(*p.shape):
        def g(t, i=i):
            q = p.copy()
            q[i] += t
            return f(q)
        fd[i] = _d1(g)
    return fd


@pytest.mark.parametrize("circuit", [
    "No_Ansatz", "Circuit_1", "Circuit_2", "Circuit_3", "Circuit_4", "Circuit_6", "Circuit_9", "Circuit_10",
    "Circuit_15", "Circuit_16", "Circuit_17", "Circuit_18", "Circuit_19", "No_Entangling", "Strongly_Entangling",
    "Hardware_Efficient", "Hardware_Efficient_2", "Ghz"])
def test_complex128_adjoint_sweep_every_ansatz_vs_central_differences(circuit):
    """`jax.grad` with x64 on (`/root/reference/tests/test_jaqsi.py:57,131-141,764-786`,
    `tests/test_model.py:1297-1333`): the complex128 adjoint sweep against complex128 central differences
    of the model itself, every ansatz with parameters at 5 qubits, gradient of a random linear cost of the
    <Z> outputs with respect to the parameters AND the inputs: 1e-9."""
    from qml_essentials_amd.ansaetze import Ansaetze
    from qml_essentials_amd.model import Model

    names = {c.__name__ for c in Ansaetze.get_available()}
    if circuit not in names:
        pytest.skip(f"{circuit} is not an ansatz of this front end")
    m = Model(5, 2, circuit, x64=True)
    p = np.asarray(m.params, dtype=np.float64).copy()
    if p.size == 0:
        pytest.skip("no parameters")
    rng = np.random.default_rng(11)
    x = np.array([0.37])
    n_out = int(np.asarray(m(params=p, inputs=x)).reshape(-1).shape[0])
    ct = rng.normal(size=(1, n_out))
    f = lambda q, xx=x: float((np.asarray(m(params=q, inputs=xx)).reshape(-1) * ct[0]).sum())  # noqa: E731
    g = np.asarray(m.gradient(params=p, inputs=x, method="adjoint", cotangent=ct)).reshape(p.shape)
    fd = _central_differences(f, p)
    assert np.abs(g - fd).max() < 1e-9, (circuit, np.abs(g - fd).max())
    gi = np.asarray(m.gradient(params=p, inputs=x, wrt="inputs", method="adjoint", cotangent=ct)).reshape(-1)
    fdi = _d1(lambda t: f(p, x + t))
    assert abs(float(gi[0]) - fdi) < 1e-9, (circuit, float(gi[0]), fdi)


def test_complex128_adjoint_sweep_batches_general_gates_and_streaming_regime():
    """The sweep on a batch (inputs x parameter sets), on gates beyond the ansatz tables (CPhase, RXX, RZZ, an
    explicit matrix, Rot) through Script.vjp, and at 15 qubits (states in HBM, not in LDS): against the
    complex128 parameter-shift Jacobian contracted by hand (1e-12) resp. central differences (1e-9)."""
    from qml_essentials_amd import operations as op
    from qml_essentials_amd.model import Model
    from qml_essentials_amd.script import Script
    from qml_essentials_amd.utils import x64_scope

    rng = np.random.default_rng(5)
    with x64_scope(True):
        m = Model(4, 2, "Circuit_19")
        x = np.array([0.1, -0.4, 0.9])
        P = rng.uniform(0, 6.28, np.asarray(m.params).shape[1:])
        ct = rng.normal(size=(3, 4))
        for wrt in ("params", "inputs"):
            J = np.asarray(m.gradient(params=P, inputs=x, wrt=wrt))
            a = np.asarray(m.gradient(params=P, inputs=x, wrt=wrt, method="adjoint", cotangent=ct))
            assert np.abs(a - np.einsum("bk,bk...->b...", ct, J)).max() < 1e-12, wrt

        U = np.linalg.qr(rng.normal(size=(2, 2)) + 1j * rng.normal(size=(2, 2)))[0]

        def circuit(t, u, v):
            op.H(wires=0)
            op.RX(t, wires=1)
            op.ControlledPhaseShift(u, wires=[0, 1])
            op.RXX(v, wires=[1, 2])
            op.Operation(wires=2, matrix=U)
            op.RZZ(t * u, wires=[0, 2])
            op.Rot(t, u, v, wires=1)
            op.CRY(v, wires=[2, 0])

        sc = Script(circuit, n_qubits=3)
        obs = [op.PauliZ(wires=0), op.PauliZ(wires=1), op.PauliZ(wires=2)]
        args = (np.float64(0.4), np.float64(-0.7), np.float64(1.3))
        w = rng.normal(size=(3,))
        g = sc.vjp(obs, w, args=args, argnums=(0, 1, 2))
        f = lambda a_: float((np.asarray(sc.execute(type="expval", obs=obs, args=tuple(np.float64(v) for v in a_))) * w).sum())  # noqa: E731
        for k in range(3):
            def gk(t, k=k):
                a_ = list(args)
                a_[k] = a_[k] + t
                return f(a_)
            assert abs(float(np.asarray(g[k])) - _d1(gk)) < 1e-9, k

        big = Model(15, 1, "Hardware_Efficient")
        pb = np.asarray(big.params, dtype=np.float64)
        gb = np.asarray(big.gradient(inputs=np.array([0.2]), method="adjoint", force_mean=True)).reshape(pb.shape)
        fb = lambda q: float(np.asarray(big(params=q, inputs=np.array([0.2]), force_mean=True)))  # noqa: E731
        for i in [(0, 0, 0), (0, 1, 7), (0, 0, 44)]:
            def gi_(t, i=i):
                q = pb.copy()
                q[i] += t
                return fb(q)
            assert abs(gb[i] - _d1(gi_)) < 1e-9, i


def test_model_training_smoke_in_x64():
    """`Model(x64=True)` in a training loop (`/root/reference/tests/test_model.py:1297-1333`): gradient descent on a
    mean-squared cost with the complex128 adjoint sweep lowers the cost monotonically over a few steps."""
    from qml_essentials_amd.model import Model

    m = Model(3, 1, "Circuit_19", x64=True)
    x = np.linspace(0, 2 * np.pi, 8, endpoint=False)
    y = 0.5 * np.cos(x)
    p = np.asarray(m.params, dtype=np.float64).copy()
    costs = []
    for _ in range(6):
        out = np.asarray(m(params=p, inputs=x, force_mean=True)).reshape(-1)
        costs.append(float(np.mean((out - y) ** 2)))
        n_out = m.n_qubits if not m.output_qubit or m.output_qubit == -1 else len(np.atleast_1d(m.output_qubit))
        ct = np.repeat((2.0 * (out - y) / x.size)[:, None], n_out, axis=1) / n_out
        g = np.asarray(m.gradient(params=p, inputs=x, method="adjoint", cotangent=ct))
        g = g.reshape(x.size, *p.shape).sum(axis=0) if g.size == x.size * p.size else g.reshape(p.shape)
        p = p - 0.5 * g
    assert all(b <= a + 1e-12 for a, b in zip(costs, costs[1:])) and costs[-1] < 0.9 * costs[0], costs
