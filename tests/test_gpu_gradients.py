"""Parameter-shift gradients against analytic answers held by the
reference's tests and central finite differences of the fp64 oracle."""
import numpy as np
import pytest

from oracle import circuits as OC
from oracle import einsum_sim as OE
from qml_essentials_amd import operations as op
from qml_essentials_amd.model import Model
from qml_essentials_amd.script import Script

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
c128 = np.complex128


def test_rx_gradient_is_minus_sin():
    """tests/test_jaqsi.py:131-141 (d<Z>/dtheta = -sin theta) and :764-786 (batched)."""
    from qml_essentials_amd import operations as op
    from qml_essentials_amd.script import Script

    s = Script(lambda th: op.RX(th, wires=0))
    (g,) = s.gradient([op.PauliZ(0, record=False)], args=(np.float32(0.5),))
    assert abs(g[0] + np.sin(0.5)) < 1e-6
    th = np.linspace(0, np.pi, 9).astype(np.float32)
    (gb,) = s.gradient([op.PauliZ(0, record=False)], args=(th,), in_axes=(0,))
    assert gb.shape == (9, 1) and np.abs(gb[:, 0] + np.sin(th)).max() < 1e-6
    # product of two arguments feeding one angle + a controlled rotation (four-term rule)
    def circ(a, b):
        op.H(wires=0)
        op.RY(a * b, wires=1)
        op.CRX(a, wires=[0, 1])
        op.CRZ(b * 2.0, wires=[1, 0])
        op.ControlledPhaseShift(a + b, wires=[0, 1])

    s2 = Script(circ, n_qubits=2)
    obs = [op.PauliZ(0, record=False), op.PauliZ(1, record=False)]
    a0, b0 = 0.7, 1.3

    def f(a, b):
        tape = [("H", [0], ()), ("RY", [1], (a * b,)), ("CRX", [0, 1], (a,)),
                ("CRZ", [1, 0], (2.0 * b,)), ("CPhase", [0, 1], (a + b,))]
        return OE.simulate_and_measure(tape, 2, "expval", [("PauliZ", [0]), ("PauliZ", [1])], c128)

    ga, gb2 = s2.gradient(obs, args=(np.float64(a0), np.float64(b0)), argnums=(0, 1))
    eps = 1e-6
    fa = (f(a0 + eps, b0) - f(a0 - eps, b0)) / (2 * eps)
    fb = (f(a0, b0 + eps) - f(a0, b0 - eps)) / (2 * eps)
    assert np.abs(ga - fa).max() < 2e-5 and np.abs(gb2 - fb).max() < 2e-5


@pytest.mark.parametrize("ansatz", ["Hardware_Efficient", "Circuit_19", "Strongly_Entangling",
                                    "Circuit_13", "Circuit_9"])
def test_model_param_gradient_vs_finite_differences(ansatz):
    from qml_essentials_amd.model import Model

    n = 3
    m = Model(n, 1, ansatz)
    spec = OC.ModelSpec(n, 1, ansatz)
    rng = np.random.default_rng(5)
    p = rng.uniform(0, 2 * np.pi, spec.params_shape)
    x = 0.4
    jac = m.gradient(params=p.astype(np.float32), inputs=np.array([x], dtype=np.float32))
    assert jac.shape == (n, *spec.params_shape)

    def f(pp):
        t = OC.model_tape(spec, pp, [x])
        return OE.simulate_and_measure(t, n, "expval", [("PauliZ", [q]) for q in range(n)], c128)

    eps = 1e-6
    for (l, j) in [(0, 0), (spec.params_shape[0] - 1, spec.params_shape[1] - 1), (0, spec.params_shape[1] // 2)]:
        d = np.zeros_like(p)
        d[l, j] = eps
        fd = (f(p + d) - f(p - d)) / (2 * eps)
        assert np.abs(jac[:, l, j] - fd).max() < 3e-5, (ansatz, l, j)


def test_model_input_and_enc_param_gradients_batched():
    from qml_essentials_amd.model import Model

    n = 2
    m = Model(n, 2, "Circuit_19")
    spec = OC.ModelSpec(n, 2, "Circuit_19")
    p = np.asarray(m.params[0], dtype=np.float64)
    X = np.array([[0.3], [1.1], [2.0]], dtype=np.float32)
    ji = m.gradient(inputs=X, wrt="inputs")
    assert ji.shape == (3, n, 1)

    def f(x, ep=None):
        t = OC.model_tape(spec, p, [x], enc_params=ep, zero_inputs_batch1=False)
        return OE.simulate_and_measure(t, n, "expval", [("PauliZ", [q]) for q in range(n)], c128)

    eps = 1e-6
    for b, x in enumerate(X[:, 0].astype(np.float64)):
        fd = (f(x + eps) - f(x - eps)) / (2 * eps)
        assert np.abs(ji[b, :, 0] - fd).max() < 3e-5
    je = m.gradient(inputs=X[:1], wrt="enc_params")
    assert je.shape == (n, 2, n, 1)
    ep = np.ones((2, n, 1))
    d = np.zeros_like(ep)
    d[1, 0, 0] = eps
    fd = (f(float(X[0, 0]), ep + d) - f(float(X[0, 0]), ep - d)) / (2 * eps)
    assert np.abs(je[:, 1, 0, 0] - fd).max() < 3e-5
    # a training step decreases an MSE cost (smoke, test_model.py:1082-1095)
    y = np.array([0.2, -0.1, 0.4])
    params = np.asarray(m.params[0], dtype=np.float32)
    def cost(pp):
        return float(np.mean((m(params=pp, inputs=X, force_mean=True) - y) ** 2))
    c0 = cost(params)
    out = m(params=params, inputs=X, force_mean=True)
    jp = m.gradient(params=params, inputs=X, force_mean=True)  # (3, L, P)
    grad = np.mean(2 * (out - y)[:, None, None] * jp, axis=0)
    assert cost(params - 0.1 * grad.astype(np.float32)) < c0


def _everything_circuit(th, x):
    op.H(wires=0); op.RX(th[0], wires=0); op.RY(th[1], wires=1); op.CX(wires=[0, 1])
    op.CRX(th[2], wires=[1, 2]); op.Rot(th[3], th[4], th[5], wires=2); op.CRZ(th[6] * x, wires=[2, 0])
    op.ControlledPhaseShift(th[7], wires=[0, 1]); op.RXX(th[8], wires=[1, 2]); op.RZZ(th[9], wires=[0, 2])
    op.RYY(th[10], wires=[0, 1]); op.RZX(th[11], wires=[2, 1]); op.CRY(th[12], wires=[0, 2])
    op.S(wires=1); op.RZ(th[13], wires=1); op.CCX(wires=[0, 1, 3]); op.SWAP(wires=[2, 3])
    op.Operation(wires=[1, 3], matrix=np.linalg.qr(np.arange(16).reshape(4, 4) % 5 + 1j * np.eye(4))[0])
    op.RY(th[14] + x, wires=3)


def test_adjoint_vjp_equals_parameter_shift_jacobian():
    """One backward sweep (qmle_adjoint_gradient) == cotangent . parameter-shift Jacobian for
    every differentiable gate kind, Z and Z-parity observables, batched and un-batched, also
    through products / sums of the arguments (tolerance 2e-6: complex64 engine)."""
    from qml_essentials_amd import jaqsi as js

    s = Script(_everything_circuit, n_qubits=4)
    rng = np.random.default_rng(0)
    obs = [op.PauliZ(wires=0, record=False), op.PauliZ(wires=3, record=False),
           js.build_parity_observable([0, 1, 2])]
    th, x = rng.uniform(0, 6.28, 15), np.array(0.7)
    jac, jx = s.gradient(obs, args=(th, x), argnums=(0, 1))
    w = np.array([0.3, -1.1, 0.8])
    g, gx = s.vjp(obs, w, args=(th, x), argnums=(0, 1))
    assert np.allclose(g, w @ jac, atol=2e-6) and abs(gx - w @ jx) < 2e-6
    TH, X = rng.uniform(0, 6.28, (5, 15)), rng.uniform(0, 1, 5)
    jac, jx = s.gradient(obs, args=(TH, X), in_axes=(0, 0), argnums=(0, 1))
    W = rng.normal(size=(5, 3))
    g, gx = s.vjp(obs, W, args=(TH, X), in_axes=(0, 0), argnums=(0, 1))
    assert np.allclose(g, np.einsum("bk,bkp->bp", W, jac), atol=4e-6)
    assert np.allclose(gx, np.einsum("bk,bk->b", W, jx), atol=4e-6)
    # several cotangents at once, (K, B, n_obs): one trace, one sweep over K * B states (the Jacobian route
    # of Model.gradient(method="adjoint")) -- the same numbers as K calls
    WK = rng.normal(size=(4, 5, 3))
    gk, gxk = s.vjp(obs, WK, args=(TH, X), in_axes=(0, 0), argnums=(0, 1))
    assert gk.shape == (4, 5, 15) and gxk.shape == (4, 5)
    for k in range(4):
        g1, gx1 = s.vjp(obs, WK[k], args=(TH, X), in_axes=(0, 0), argnums=(0, 1))
        assert np.allclose(gk[k], g1, atol=1e-6) and np.allclose(gxk[k], gx1, atol=1e-6)
    g1k, _ = s.vjp(obs, WK[:, :1], args=(th, x), argnums=(0, 1))     # un-batched arguments, K cotangents
    assert g1k.shape == (4, 15) and np.allclose(g1k[2], s.vjp(obs, WK[2, 0], args=(th, x), argnums=(0, 1))[0], atol=1e-6)
    with pytest.raises(NotImplementedError):
        s.vjp([op.PauliX(wires=0, record=False)], np.ones(1), args=(th, x))


@pytest.mark.parametrize("ansatz,n", [("Hardware_Efficient", 5), ("Circuit_19", 4),
                                      ("Strongly_Entangling", 4), ("Circuit_9", 6)])
def test_model_adjoint_gradient_matches_parameter_shift(ansatz, n):
    model = Model(n_qubits=n, n_layers=2, circuit_type=ansatz)
    rng = np.random.default_rng(n)
    x = rng.uniform(0, 2 * np.pi, (3, 1))
    ps = model.gradient(inputs=x)
    ad = model.gradient(inputs=x, method="adjoint")
    assert ps.shape == ad.shape and np.allclose(ps, ad, atol=3e-6)
    ps_m = model.gradient(inputs=x, force_mean=True)
    ad_m = model.gradient(inputs=x, force_mean=True, method="adjoint")      # ONE sweep
    assert ps_m.shape == ad_m.shape and np.allclose(ps_m, ad_m, atol=3e-6)
    for wrt in ("inputs", "enc_params"):
        a = model.gradient(inputs=x, wrt=wrt, force_mean=True)
        b = model.gradient(inputs=x, wrt=wrt, force_mean=True, method="adjoint")
        assert a.shape == b.shape and np.allclose(a, b, atol=3e-6), wrt
    # explicit cotangent: gradient of a cost with dC/d<Z_q>_b = cot[b, q]
    cot = rng.normal(size=(3, n))
    vjp = model.gradient(inputs=x, method="adjoint", cotangent=cot)
    assert np.allclose(vjp, np.einsum("bk,bk...->b...", cot, ps), atol=5e-6)


def test_golomb_encoding_adjoint_input_gradient():
    from qml_essentials_amd.unitary import UnitaryGates

    def circ(th, x):
        op.H(wires=0); op.H(wires=1); op.H(wires=2)
        op.RY(th[0], wires=1)
        UnitaryGates.GolombEncoding(x * 0.37, wires=[0, 1, 2])
        op.CX(wires=[0, 2]); op.RX(th[1], wires=2); op.H(wires=0)

    s = Script(circ, n_qubits=3)
    obs = [op.PauliZ(wires=q, record=False) for q in range(3)]
    th, x = np.array([0.4, 1.3]), np.array(0.9)
    w = np.array([1.0, -0.5, 0.25])
    g, gx = s.vjp(obs, w, args=(th, x), argnums=(0, 1))
    f = lambda t, xx: float(w @ np.asarray(s.execute("expval", obs, args=(t, np.array(xx)))))  # noqa: E731
    e = 1e-3
    fd_x = (f(th, 0.9 + e) - f(th, 0.9 - e)) / (2 * e)
    fd_0 = (f(th + [e, 0], 0.9) - f(th - [e, 0], 0.9)) / (2 * e)
    assert abs(gx - fd_x) < 2e-3 and abs(g[0] - fd_0) < 2e-3


def test_adjoint_tiny_registers_and_streaming_path(monkeypatch):
    """n = 1, 2 (LDS-resident sweep only) and the per-gate streaming sweep forced at n = 6
    (QMLE_ADJOINT_NO_LDS is read once per process, so the streaming path is exercised through
    a 15-qubit circuit instead)."""
    def one(th):
        op.RX(th[0], wires=0); op.RY(th[1], wires=0); op.RZ(th[2], wires=0); op.RX(th[3], wires=0)

    s = Script(one, n_qubits=1)
    obs = [op.PauliZ(wires=0, record=False)]
    th = np.array([0.3, 1.1, -0.7, 2.0])
    (jac,) = s.gradient(obs, args=(th,))
    (g,) = s.vjp(obs, np.ones(1), args=(th,))
    assert np.allclose(g, jac[0], atol=2e-6)

    def big(th):
        for q in range(15):
            op.RY(th[q], wires=q)
        for q in range(14):
            op.CRX(th[15 + q], wires=[q, q + 1])
        op.ControlledPhaseShift(th[29], wires=[14, 0]); op.RXX(th[30], wires=[3, 9])
        op.Rot(th[31], th[32], th[33], wires=7)

    s = Script(big, n_qubits=15)
    obs = [op.PauliZ(wires=q, record=False) for q in (0, 7, 14)]
    th = np.random.default_rng(3).uniform(0, 6.28, 34)
    (jac,) = s.gradient(obs, args=(th,))
    w = np.array([0.5, -1.0, 2.0])
    (g,) = s.vjp(obs, w, args=(th,))
    assert np.allclose(g, w @ jac, atol=4e-6)


@pytest.mark.parametrize("ansatz,n", [("Hardware_Efficient", 14), ("Strongly_Entangling", 15),
                                      ("Circuit_9", 14), ("Circuit_19", 16)])
def test_fused_adjoint_tile_passes_match_parameter_shift(ansatz, n):
    """n >= 14: the sweep runs as fused tile passes over [psi; lambda] (k_tile_adj) -- RX/RY/RZ,
    Rot (split), CX, CZ, CRX generators and inverses inside register groups; compare with the
    parameter-shift Jacobian for parameters, inputs and encoding weights."""
    model = Model(n_qubits=n, n_layers=1, circuit_type=ansatz)
    rng = np.random.default_rng(n)
    x = rng.uniform(0, 2 * np.pi, (2, 1))
    cot = rng.normal(size=(2, n))
    for wrt in ("params", "inputs", "enc_params"):
        jac = np.asarray(model.gradient(inputs=x, wrt=wrt))
        vjp = np.asarray(model.gradient(inputs=x, wrt=wrt, method="adjoint", cotangent=cot))
        assert np.allclose(vjp, np.einsum("bk,bk...->b...", cot, jac), atol=1e-5), wrt
