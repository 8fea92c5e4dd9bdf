"""Correctness at the sizes that are otherwise only *timed* (VERDICT r1, weak #1 / ADVICE r1):

* n = 28 (2 GiB state, byte offsets >= 2^31): one gate of every K1 kind through the streaming
  (non-temporal) ``k_direct_1q`` path on wires 0 / 1 / 13 / 26 / 27, against closed-form <Z> of a
  product state AND against the LDS-tile path bit pattern for bit pattern;
* Meyer-Wallach at n = 28: product state (0), GHZ (1), and a Hardware-Efficient layer whose
  single-qubit purities are rebuilt independently from <X>, <Y>, <Z>;
* K2 at >= 1 GiB per launch (n = 24, 8 states): default engine vs all-amplitudes-live plan vs
  the oracle's C port;
* BASELINE configs C3 (12 qubits x 1024 pairs) and C4 (10 qubits x 4096-point grid) once at full
  size, 64 sampled states / grid points against the complex128 oracle;
* the fast tile kernel (k_tile2: table-addressed groups, CX folded into the LDS layout) against
  the generic tile kernel on random CX-rich tapes.

Reference behaviour pinned: tests/test_jaqsi.py:372-427 (<Z> = cos theta, wire order),
qml_essentials/entanglement.py:69-103 (Meyer-Wallach), expressibility.py:14-66, coefficients.py:109-150.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
c128 = np.complex128


def _N():
    from qml_essentials_amd import _native as N
    return N


def _product_state(n, theta):
    """RY(theta_w) on every wire of |0..0>: <Z_w> = cos theta_w (tests/test_jaqsi.py:372-380)."""
    N = _N()
    plan = N.Plan([("RY", [q], [q], -1) for q in range(n)], n, n)
    a = torch.from_numpy(np.asarray(theta, dtype=np.float32)[None]).cuda()
    return plan.run(a, "state")


def _apply(st, n, gate, wires, angle=None, force_tile=False):
    N = _N()
    slots = [] if angle is None else [0]
    flags = N.plan_flags(force_tile=True, force_global=True) if force_tile else N.PLAN_NO_FUSION
    plan = N.Plan([(gate, list(wires), slots, -1)], n, 1, flags=flags)
    kinds = [s["kind"] for s in plan.describe()["stages"]]
    ang = torch.full((1, 1), 0.0 if angle is None else float(angle), device="cuda")
    N.apply_inplace(plan, ang, st)
    return kinds


def test_n28_single_gates_streaming_path_closed_form_and_tile_path():
    N = _N()
    n = 28
    if torch.cuda.mem_get_info()[0] < 12 * (1 << 30):
        pytest.skip("needs ~8 GiB of free HBM")
    rng = np.random.default_rng(28)
    theta = rng.uniform(0.3, 2.8, n)
    c = np.cos(theta)
    base = _product_state(n, theta)
    assert base.shape == (1, 1 << n)
    ez = N.expval_z(base, list(range(n))).cpu().numpy()[0]
    assert np.abs(ez - c).max() < 1e-6  # the product state itself, every wire
    phi, alpha = 0.9, 1.7
    for w in (0, 1, 13, 26, 27):
        c2 = (w + 1) % n  # control wire of the two-qubit kinds (control = target + 1)
        cases = {
            "RX": ([("RX", [w], phi)], {w: c[w] * np.cos(phi)}),
            # RZ is invisible in <Z>: follow it with RX(alpha) on the same wire
            "RZ": ([("RZ", [w], phi), ("RX", [w], alpha)],
                   {w: np.sin(theta[w]) * np.sin(phi) * np.sin(alpha) + c[w] * np.cos(alpha)}),
            "CX": ([("CX", [c2, w], None)], {w: c[c2] * c[w]}),
            "CRX": ([("CRX", [c2, w], phi)],
                    {w: c[w] * ((1 + c[c2]) / 2 + (1 - c[c2]) / 2 * np.cos(phi))}),
        }
        for kind, (gates, expect) in cases.items():
            st = base.clone()
            kinds = []
            for g, wires, ang in gates:
                kinds += _apply(st, n, g, wires, ang)
            # control on bit positions 1..3 goes through a single-gate tile pass by design
            # (qmle_plan.cpp direct_ok) unless the target sits on positions 1..6, which the
            # lane-exchange streaming mode 7 of k_direct_1q serves; everything else must stream
            low_ctrl = kind in ("CX", "CRX") and 1 <= n - 1 - c2 <= 3 and not 1 <= n - 1 - w <= 6
            assert all(k == ("tile" if low_ctrl else "direct") for k in kinds), (kind, w, kinds)
            got = N.expval_z(st, list(range(n))).cpu().numpy()[0]
            want = c.copy()
            for q, v in expect.items():
                want[q] = v
            assert np.abs(got - want).max() < 1e-6, (kind, w, np.abs(got - want).max())
            # the LDS-tile path on the same input: same amplitudes to rounding
            st2 = base.clone()
            for g, wires, ang in gates:
                _apply(st2, n, g, wires, ang, force_tile=True)
            diff = float((torch.view_as_real(st) - torch.view_as_real(st2)).abs().max())
            assert diff < 2e-7, (kind, w, diff)
            del st, st2
    del base
    torch.cuda.empty_cache()


def _xyz_of_wire(st, n, j):
    """<X_j>, <Y_j>, <Z_j> from <Z_j> after RY(-pi/2) / RX(pi/2) on a copy (independent of the
    Meyer-Wallach kernels: single-gate streaming kernel + the all-qubit <Z> reduction)."""
    N = _N()
    z = float(N.expval_z(st, [j])[0, 0])
    t = st.clone()
    _apply(t, n, "RY", [j], -np.pi / 2)
    x = float(N.expval_z(t, [j])[0, 0])
    t.copy_(st)
    _apply(t, n, "RX", [j], np.pi / 2)
    y = float(N.expval_z(t, [j])[0, 0])
    del t
    return x, y, z


def test_n28_meyer_wallach_product_ghz_and_he_layer():
    N = _N()
    n = 28
    if torch.cuda.mem_get_info()[0] < 12 * (1 << 30):
        pytest.skip("needs ~8 GiB of free HBM")
    st = _product_state(n, np.random.default_rng(5).uniform(0, 6.28, n))
    q, pur = N.meyer_wallach(st, return_purities=True)
    # (2e-6: the float32 product state's own norm is off by 2.5e-7 -- 28 rounded factors per amplitude --
    # and a purity is quadratic in it; the reductions themselves agree with float64 sums of the same
    # state to 4e-8, tools/accum_probe.py)
    assert abs(float(q[0])) < 2e-6 and float((pur - 1).abs().max()) < 2e-6
    del st
    ghz = N.Plan([("H", [0], [], -1)] + [("CX", [k, k + 1], [], -1) for k in range(n - 1)], n, 0)
    st = ghz.run(None, "state")
    q, pur = N.meyer_wallach(st, return_purities=True)
    assert abs(float(q[0]) - 1) < 1e-6 and float((pur - 0.5).abs().max()) < 1e-6
    del st
    from tests.test_abi_cpu import he_layer_ops
    ops, slots = he_layer_ops(n)
    ang = torch.from_numpy(np.random.default_rng(6).uniform(0, 6.28, (1, slots)).astype(np.float32)).cuda()
    st = N.Plan(ops, n, slots).run(ang, "state")
    assert abs(float((st.abs() ** 2).sum()) - 1) < 1e-6
    q, pur = N.meyer_wallach(st, return_purities=True)
    pur = pur.cpu().numpy()[0]
    for j in (0, 13, 27):
        x, y, z = _xyz_of_wire(st, n, j)
        assert abs(pur[j] - (1 + x * x + y * y + z * z) / 2) < 2e-6, (j, pur[j], x, y, z)  # (three float32 states' <Z>)
    assert abs(float(q[0]) - 2 * (1 - pur.mean())) < 1e-6
    del st
    torch.cuda.empty_cache()


def test_k2_one_gib_launch_default_vs_all_live_vs_c_port():
    """K2 (n = 24 Hardware-Efficient layer) at 8 states = 1 GiB per launch, so the non-temporal
    tile loads / stores and k_product_stream<NT> / k_reg_measure_mono<NT> variants run: default
    engine == all-amplitudes-live plan (no known zeros, nothing folded) == oracle C port."""
    N = _N()
    from oracle import c_port, circuits as OC
    from tests.test_abi_cpu import he_layer_ops

    n, B = 24, 8
    ops, slots = he_layer_ops(n)
    rng = np.random.default_rng(1000)
    ang_h = rng.uniform(0, 2 * np.pi, (B, slots)).astype(np.float32)
    ang = torch.from_numpy(ang_h).cuda()
    obs = list(range(n))
    e_def = N.Plan(ops, n, slots).run(ang, "expval", obs).cpu().numpy()
    dense = N.Plan(ops, n, slots, flags=N.PLAN_NO_SPARSE | N.PLAN_NO_ABSORB)
    assert all(s["fast"] for s in dense.describe()["stages"])
    e_live = dense.run(ang, "expval", obs).cpu().numpy()
    e_fold = N.Plan(ops, n, slots, flags=N.PLAN_NO_SPARSE).run(ang, "expval", obs).cpu().numpy()
    assert np.abs(e_def - e_live).max() < 1e-6 and np.abs(e_fold - e_live).max() < 1e-6
    # the state itself, all-live vs default engine
    s_live = dense.run(ang[:2], "state")
    s_def = N.Plan(ops, n, slots).run(ang[:2], "state")
    assert float((torch.view_as_real(s_live) - torch.view_as_real(s_def)).abs().max()) < 3e-7
    del s_live, s_def
    # two of the statevectors on the CPU port (he_layer_ops order: RY, RZ, RY per wire, then CX)
    for b in (0, B - 1):
        tape = []
        for name, wires, sl, _ in ops:
            tape.append((name, wires, tuple(float(ang_h[b, s]) for s in sl)))
        psi = c_port.simulate(tape, n)
        ez = c_port.expval_z(psi, n, obs)
        assert np.abs(ez - e_live[b]).max() < 1e-6, (b, np.abs(ez - e_live[b]).max())


def test_multi_tile_workgroups_state_and_probs_vs_c_port():
    """All-live storing passes hand a workgroup several consecutive tiles once the grid is large
    enough (n = 22 x 16 states: 2 tiles per workgroup, next tile prefetched into registers).  The
    TM_STORE and TM_PROBS epilogues of that variant against the oracle's C port, and against the
    default engine (known zeros tracked, one tile per workgroup)."""
    N = _N()
    from oracle import c_port
    from tests.test_abi_cpu import he_layer_ops

    n, B = 22, 16
    ops, slots = he_layer_ops(n)
    rng = np.random.default_rng(22016)
    ang_h = rng.uniform(0, 2 * np.pi, (B, slots)).astype(np.float32)
    ang = torch.from_numpy(ang_h).cuda()
    dense = N.Plan(ops, n, slots, flags=N.PLAN_NO_SPARSE | N.PLAN_NO_ABSORB)
    assert all(s["fast"] for s in dense.describe()["stages"])
    st = dense.run(ang, "state")
    pr = dense.run(ang, "probs")
    ref = N.Plan(ops, n, slots).run(ang, "state")
    assert float((torch.view_as_real(st) - torch.view_as_real(ref)).abs().max()) < 3e-7
    assert float((pr - (st.real ** 2 + st.imag ** 2)).abs().max()) < 1e-9   # same plan, same arithmetic
    assert float((pr - (ref.real ** 2 + ref.imag ** 2)).abs().max()) < 2e-8
    for b in (0, 7, B - 1):
        tape = [(name, wires, tuple(float(ang_h[b, s]) for s in sl)) for name, wires, sl, _ in ops]
        psi = c_port.simulate(tape, n)
        assert np.abs(st[b].cpu().numpy() - psi).max() < 3e-7
        assert np.abs(pr[b].cpu().numpy() - np.abs(psi) ** 2).max() < 2e-8


def test_c3_expressibility_full_size_sampled_oracle():
    """BASELINE config 3: 12 qubits, 1024 pairs (2048 parameter sets), HE 3 layers, no DRU."""
    from oracle import circuits as OC, einsum_sim as OE
    from qml_essentials_amd.expressibility import Expressibility
    from qml_essentials_amd.model import Model

    n, S = 12, 1024
    m = Model(n, 3, "Hardware_Efficient", data_reupload=False)
    fid = Expressibility._sample_state_fidelities(m, S, random_key=1000).cpu().numpy()
    assert fid.shape == (S,)
    params = np.asarray(m.params)
    assert params.shape[0] == 2 * S
    spec = OC.ModelSpec(n, 3, "Hardware_Efficient", data_reupload=False)
    idx = np.random.default_rng(0).choice(S, 64, replace=False)
    worst = 0.0
    for i in idx:
        a = OE.simulate_pure(OC.model_tape(spec, params[i], [0.0]), n, c128)
        b = OE.simulate_pure(OC.model_tape(spec, params[i + S], [0.0]), n, c128)
        worst = max(worst, abs(fid[i] - abs(np.vdot(a, b)) ** 2))
    assert worst < 1e-6, worst


def test_c4_fourier_grid_full_size_sampled_oracle():
    """BASELINE config 4: Model(10, 6, HE), 4096-point input grid, force_mean."""
    from oracle import circuits as OC, einsum_sim as OE
    from qml_essentials_amd.model import Model

    n, G = 10, 4096
    m = Model(n, 6, "Hardware_Efficient")
    p = np.asarray(m.params[0])
    grid = (np.arange(G, dtype=np.float64) * 2 * np.pi / G).astype(np.float32).reshape(G, 1)
    out = np.asarray(m(inputs=grid, force_mean=True))
    assert out.shape == (G,)
    spec = OC.ModelSpec(n, 6, "Hardware_Efficient")
    idx = np.random.default_rng(1).choice(G, 64, replace=False)
    obs = [("PauliZ", [q]) for q in range(n)]
    worst = 0.0
    for k in idx:
        want = OE.simulate_and_measure(OC.model_tape(spec, p, [float(grid[k, 0])]), n, "expval", obs, c128)
        worst = max(worst, abs(out[k] - float(np.mean(want))))
    assert worst < 1e-6, worst


def test_fast_tile_kernel_equals_generic_kernel_on_cx_rich_tapes():
    """k_tile2 (host-built address tables, X / CX folded into the LDS layout, asm gate blocks)
    against the generic k_tile (QMLE_NO_FAST_TILE is read once per process, so the generic side
    is a plan the fast path refuses: NO_REGTILE) and against the oracle, on random tapes of
    1-qubit / controlled gates dense in X / CX."""
    N = _N()
    from oracle import einsum_sim as OE
    from tests.helpers import tape_to_native

    rng = np.random.default_rng(77)
    one = ["RX", "RY", "RZ", "H", "PauliX", "S", "Rot"]
    two = ["CX", "CX", "CX", "CZ", "CRX", "CRY", "CRZ", "CPhase", "CY"]
    npar = {"RX": 1, "RY": 1, "RZ": 1, "Rot": 3, "CRX": 1, "CRY": 1, "CRZ": 1, "CPhase": 1}
    for trial in range(12):
        n = int(rng.integers(15, 18))
        tape = []
        for _ in range(int(rng.integers(20, 70))):
            if rng.random() < 0.55:
                name = two[rng.integers(len(two))]
                wires = [int(x) for x in rng.choice(n, 2, replace=False)]
            else:
                name = one[rng.integers(len(one))]
                wires = [int(rng.integers(n))]
            tape.append((name, wires, tuple(float(x) for x in rng.uniform(0, 6.28, npar.get(name, 0)))))
        ops, angles, consts = tape_to_native(tape, n)
        ang = torch.from_numpy(angles[None]).cuda()
        T = int(rng.choice([10, 11, 12, 13]))
        L = int(rng.integers(3, 7))
        flags = N.PLAN_NO_SPARSE | N.PLAN_NO_ABSORB | N.plan_flags(tile_bits=T, low_bits=L)
        fast = N.Plan(ops, n, max(1, len(angles)), consts, flags)
        assert any(s["fast"] for s in fast.describe()["stages"])
        slow = N.Plan(ops, n, max(1, len(angles)), consts, flags | N.PLAN_NO_REGTILE)
        assert not any(s["fast"] for s in slow.describe()["stages"])
        want = OE.simulate_pure(tape, n, c128)
        for meas in ("state", "probs", "expval"):
            a = fast.run(ang, meas, list(range(n)) if meas == "expval" else ())
            b = slow.run(ang, meas, list(range(n)) if meas == "expval" else ())
            assert float((torch.view_as_real(a) if a.is_complex() else a).sub(
                torch.view_as_real(b) if b.is_complex() else b).abs().max()) < 1e-6, (trial, meas)
        got = fast.run(ang, "state").cpu().numpy()[0]
        assert np.abs(got - want).max() < 1e-6, (trial, np.abs(got - want).max())
        # the same kernel with known-zero tracking on (partial loads, zero tiles, compacted grids,
        # idle work items): default engine against the oracle, state and <Z>
        geo = N.plan_flags(tile_bits=T, low_bits=L)
        for extra in (0, N.PLAN_NO_ABSORB):
            sp = N.Plan(ops, n, max(1, len(angles)), consts, geo | extra)
            got = sp.run(ang, "state").cpu().numpy()[0]
            assert np.abs(got - want).max() < 1e-6, (trial, extra, np.abs(got - want).max())
            ez = sp.run(ang, "expval", list(range(n))).cpu().numpy()[0]
            p = np.abs(want) ** 2
            idx = np.arange(1 << n)
            wz = np.array([np.sum(p * (1 - 2 * ((idx >> (n - 1 - q)) & 1))) for q in range(n)])
            assert np.abs(ez - wz).max() < 1e-6, (trial, extra, np.abs(ez - wz).max())


def test_n29_falls_back_to_64_bit_addressing():
    """n = 29 (4 GiB per state): the fast tile kernel addresses a tile with 32-bit byte
    offsets, valid up to n = 28; beyond that the generic kernels must take over (the Meyer-Wallach
    read kernels use 64-bit offsets and take a fourth read: chunks {12-15, 25-28}, {16-19, 24-27},
    {8-11, 20-23}).  Product
    state through the fused tile passes: <Z_w> = cos(theta_w) on every wire, Meyer-Wallach 0."""
    N = _N()
    n = 29
    if torch.cuda.mem_get_info()[0] < 16 * (1 << 30):
        pytest.skip("needs ~10 GiB of free HBM")
    theta = np.random.default_rng(29).uniform(0.2, 2.9, n)
    st = _product_state(n, theta)
    ez = N.expval_z(st, list(range(n))).cpu().numpy()[0]
    assert np.abs(ez - np.cos(theta)).max() < 1e-6
    q = N.meyer_wallach(st)
    assert abs(float(q[0])) < 2e-6
    # one gate on the top and on the bottom wire through the streaming kernel
    _apply(st, n, "RX", [0], 0.7)
    _apply(st, n, "RX", [n - 1], 0.4)
    ez = N.expval_z(st, [0, n - 1]).cpu().numpy()[0]
    assert abs(ez[0] - np.cos(theta[0]) * np.cos(0.7)) < 1e-6
    assert abs(ez[1] - np.cos(theta[n - 1]) * np.cos(0.4)) < 1e-6
    del st
    torch.cuda.empty_cache()


def test_known_zero_fuzz_short():
    """A short run of tools/fuzz_sparse_kernels.py inside the suite: random shallow tapes on random
    wire subsets and tile geometries, state / probs / <Z> / parities with known-zero tracking
    against the all-live plan (both now run on k_tile2 wherever the stage qualifies)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FUZZ_N="24", FUZZ_SEED="7")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_sparse_kernels.py")],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "mismatches: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("n", [29, 30, 32])
def test_registers_beyond_28_qubits(n):
    """n = 29 / 30 / 32 = QMLE_MAX_QUBITS (4 / 8 / 32 GiB per state; the fast tile kernel addresses a state with 32-bit byte
    offsets up to n = 28, larger registers take the generic kernels): RY on every wire, CX on the
    first and the last pair, a diagonal and a controlled rotation -- closed forms <Z_w> = cos t_w,
    <Z_1> = cos t_0 cos t_1, <Z_{n-1}> = cos t_{n-2} cos t_{n-1} (`operations.py:1029-1031,1074`);
    default engine, all-live engine, stored state + stand-alone <Z>, Meyer-Wallach purities."""
    from qml_essentials_amd import _native as N

    rng = np.random.default_rng(n)
    # 1.5e-6: the float32 STATE, not the sums -- every amplitude is a product of n rounded factors (measured
    # 1.0e-6 at n = 29, 1.25e-6 at n = 32); the reductions agree with float64 sums of the same float32
    # amplitudes to 7e-8 at n = 28 (tools/accum_probe.py): block partials in fp32 over <= 512 terms, fp64 above
    tol = 1.5e-6
    th = rng.uniform(0, np.pi, n).astype(np.float32)
    th[[0, n - 2]], th[[1, n - 1]] = np.pi / 2, np.pi / 3
    ops = [("RY", [q], [q], -1) for q in range(n)] + [("CX", [0, 1], [], -1), ("CX", [n - 2, n - 1], [], -1),
                                                       ("RZ", [3], [n], -1), ("CRX", [5, 6], [n + 1], -1)]
    ang = torch.from_numpy(np.concatenate([th, [0.3, 0.0]]).astype(np.float32)[None]).cuda()
    want = np.cos(th.astype(np.float64))
    want[1] = np.cos(th[0]) * np.cos(th[1])
    want[n - 1] = np.cos(th[n - 2]) * np.cos(th[n - 1])
    for flags in (0, N.PLAN_NO_SPARSE | N.PLAN_NO_ABSORB):
        z = N.Plan(ops, n, n + 2, flags=flags).run(ang, "expval", list(range(n))).cpu().numpy()[0]
        assert np.abs(z - want).max() < tol, flags
    st = N.Plan(ops, n, n + 2).run(ang, "state")
    assert abs(float((st.abs() ** 2).sum()) - 1.0) < 2e-6  # (the norm of a float32 state after n + 4 rounded gates: 1.0e-6 .. 1.3e-6)
    assert np.abs(N.expval_z(st, list(range(n))).cpu().numpy()[0] - want).max() < tol
    q, pur = N.meyer_wallach(st, return_purities=True)
    pur = pur.cpu().numpy()[0].astype(np.float64)
    # wires outside the two CX pairs are in product states (purity 1); a pair RY(a), RY(b), CX has
    # Tr rho^2 = 1 - sin^2(a) cos^2(b) / 2 on both of its wires: 0.875 for a = pi/2, b = pi/3
    free = [w for w in range(n) if w not in (0, 1, n - 2, n - 1)]
    assert np.abs(pur[free] - 1.0).max() < 2 * tol
    assert np.abs(pur[[0, 1, n - 2, n - 1]] - 0.875).max() < 2 * tol
    assert abs(float(q) - 2 * (1 - pur.mean())) < 2 * tol
    del st
    torch.cuda.empty_cache()


def test_one_gate_passes_at_24_qubits_controls_and_targets_all_over_the_register():
    """k_direct_1q as the library picks its modes at n >= 24 (round 4: workgroups spread over the state for
    controls on positions 7 / 8, 4-rows-per-stream bursts of the controlled gate by the measured position
    rules, bursts of the plain gate from position 21): a state of random single-qubit rotations, then CX /
    CRX / CRY on control-target pairs that hit every rule -- neighbours above and below, distant pairs, controls
    on positions 0..3, 4, 7, 8 -- one gate per launch (QMLE_PLAN_NO_FUSION | FORCE_GLOBAL), against the oracle's
    C port and against the fused default engine."""
    N = _N()
    from oracle import c_port
    from tests.helpers import tape_to_native

    n = 24
    rng = np.random.default_rng(2404)
    pos = lambda p: n - 1 - p  # wire of bit position p
    tape = [("RY", [q], (float(rng.uniform(0, 6.28)),)) for q in range(n)]
    tape += [("RZ", [q], (float(rng.uniform(0, 6.28)),)) for q in range(n)]
    pairs = [(22, 23), (23, 22), (20, 21), (21, 17), (16, 17), (18, 17), (10, 11), (12, 11), (9, 10), (11, 10),
             (7, 8), (8, 9), (8, 20), (7, 23), (4, 5), (4, 22), (3, 21), (2, 23), (0, 12), (1, 6), (15, 14), (13, 14),
             (19, 12), (12, 19), (5, 16), (23, 9)]
    kinds = ["CX", "CRX", "CRY"]
    for i, (pc, pt) in enumerate(pairs):
        g = kinds[i % 3]
        tape.append((g, [pos(pc), pos(pt)], () if g == "CX" else (float(rng.uniform(0.3, 2.8)),)))
        tape.append(("RX", [pos(pt)], (float(rng.uniform(0, 6.28)),)))
    ops, angles, consts = tape_to_native(tape, n)
    ang = torch.from_numpy(np.ascontiguousarray(angles[None, :], dtype=np.float32)).cuda()
    one = N.Plan(ops, n, len(angles), consts, N.plan_flags(no_fusion=True, force_global=True))
    kinds_run = [s["kind"] for s in one.describe()["stages"]]
    assert kinds_run.count("direct") >= len(kinds_run) - 8, kinds_run  # (a few low-control pairs are tile passes)
    got = one.run(ang, "state")[0]
    fused = N.Plan(ops, n, len(angles), consts).run(ang, "state")[0]
    assert float((torch.view_as_real(got) - torch.view_as_real(fused)).abs().max()) < 1e-6
    psi = c_port.simulate(tape, n)
    assert np.abs(got.cpu().numpy() - psi).max() < 1e-6


def test_diagonal_controlled_passes_at_24_qubits_all_over_the_register():
    """Round 5: CZ / CPhase run as the |11>-quarter pass (k_direct_1q mode 9) and CRZ as the diagonal controlled
    pass; at n >= 24 both take the spread workgroup order when the control or the target sits on position 7 / 8
    (qmle_direct.hip, blk_mul 4097) -- an order the small-register sweeps of test_gpu_kernels never reach.  One
    gate per launch over pairs on every rule (positions 7 / 8 as control and as target, neighbours, distant
    pairs, low controls that fall back to the tile pass), against the oracle's C port and the fused engine."""
    N = _N()
    from oracle import c_port
    from tests.helpers import tape_to_native

    n = 24
    rng = np.random.default_rng(2405)
    pos = lambda p: n - 1 - p
    tape = [("RY", [q], (float(rng.uniform(0, 6.28)),)) for q in range(n)]
    tape += [("RX", [q], (float(rng.uniform(0, 6.28)),)) for q in range(n)]
    pairs = [(7, 8), (8, 7), (7, 20), (20, 7), (8, 23), (23, 8), (7, 4), (4, 8), (8, 12), (12, 7), (22, 23),
             (16, 17), (5, 6), (6, 5), (9, 10), (21, 4), (4, 21), (2, 8), (7, 1), (0, 23), (3, 2), (13, 19),
             (17, 18), (18, 16), (16, 18)]  # (pairs inside 16..18: workgroups 17 blocks apart)
    kinds = ["CZ", "CPhase", "CRZ"]
    for i, (pc, pt) in enumerate(pairs):
        for j in range(3 if pc in (7, 8) or pt in (7, 8) or (16 <= pc <= 18 and 16 <= pt <= 18) else 1):
            g = kinds[(i + j) % 3]
            tape.append((g, [pos(pc), pos(pt)], () if g == "CZ" else (float(rng.uniform(0.3, 2.8)),)))
        tape.append(("RY", [pos(pt)], (float(rng.uniform(0, 6.28)),)))
        tape.append(("RX", [pos(pc)], (float(rng.uniform(0, 6.28)),)))
    ops, angles, consts = tape_to_native(tape, n)
    ang = torch.from_numpy(np.ascontiguousarray(angles[None, :], dtype=np.float32)).cuda()
    one = N.Plan(ops, n, len(angles), consts, N.plan_flags(no_fusion=True, force_global=True))
    kinds_run = [s["kind"] for s in one.describe()["stages"]]
    assert kinds_run.count("direct") >= len(kinds_run) - 8, kinds_run
    got = one.run(ang, "state")[0]
    fused = N.Plan(ops, n, len(angles), consts).run(ang, "state")[0]
    assert float((torch.view_as_real(got) - torch.view_as_real(fused)).abs().max()) < 1e-6
    psi = c_port.simulate(tape, n)
    assert np.abs(got.cpu().numpy() - psi).max() < 1e-6
