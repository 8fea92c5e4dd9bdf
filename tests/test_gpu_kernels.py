"""GPU parity of the C-ABI kernels against the NumPy oracle (bit-exact indexing,
<= 1e-6 on values; tolerance stated per test)."""
import numpy as np
import pytest

from oracle import analysis as OA
from oracle import einsum_sim as OE
from tests.helpers import oracle_tape, random_tape, tape_to_native

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _run(tape, n, meas, obs=(), flags=0, batch_angles=None):
    from qml_essentials_amd import _native as N

    ops, angles, consts = tape_to_native(tape, n)
    plan = N.Plan(ops, n, len(angles), consts, flags)
    a = angles[None, :] if batch_angles is None else batch_angles
    ang = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    if ang.shape[1] == 0:
        ang = torch.zeros((ang.shape[0], 1), dtype=torch.float32, device="cuda")[:, :0]
    out = plan.run(ang, meas, obs)
    torch.cuda.synchronize()
    return out.cpu().numpy(), plan


def _modes(n):
    from qml_essentials_amd import _native as N

    modes = {"lds": 0}
    if n >= 3:
        modes["direct"] = N.plan_flags(no_fusion=True, force_global=True)
    if n >= 6:
        modes["tile_unfused"] = N.plan_flags(no_fusion=True, force_global=True, force_tile=True,
                                             tile_bits=5, low_bits=2)
        modes["tile_fused"] = N.plan_flags(force_global=True, tile_bits=min(6, n - 1), low_bits=3)
        modes["tile_fused_L1"] = N.plan_flags(force_global=True, tile_bits=5, low_bits=1)
    return modes


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 11])
def test_random_circuits_state_parity_all_modes(n):
    rng = np.random.default_rng(100 + n)
    tape = random_tape(n, 40 if n > 1 else 10, rng)
    want = OE.simulate_pure(tape, n, np.complex128)
    for mode, flags in _modes(n).items():
        got, _ = _run(tape, n, "state", flags=flags)
        err = np.abs(got[0] - want).max()
        assert err < 1e-6, (mode, n, err)  # fp32 state vs fp64 oracle, 40 gates (measured: <= 7e-7)


@pytest.mark.parametrize("n,layers", [(15, 1), (16, 2), (18, 2), (21, 1)])
def test_from_zero_variant_equals_the_plans_own_schedule_and_the_oracle(n, layers):
    """Round 3: qmle_run_batch executes the from-|0..0> variant of an all-live plan (first stage
    on a 2^14-amplitude tile, bit position 6 carried), qmle_apply_inplace the plan's own stages.
    Same tape, same angles: both must give the oracle's state (n <= 18) and each other's."""
    from qml_essentials_amd import _native as N
    from tests.test_abi_cpu import he_layer_ops

    ops1, slots1 = he_layer_ops(n)
    ops = []
    for l in range(layers):
        ops += [(g, w, [s + l * slots1 for s in sl], off) for g, w, sl, off in ops1]
    ops += [("CRX", [0, n - 1], [0], -1), ("RZ", [n // 2], [1], -1), ("H", [1], [], -1)]
    slots = layers * slots1
    ang = np.random.default_rng(n).uniform(0, 2 * np.pi, (2, slots)).astype(np.float32)
    angd = torch.from_numpy(ang).cuda()
    plan = N.Plan(ops, n, slots, flags=N.PLAN_NO_SPARSE | N.PLAN_NO_ABSORB)
    assert plan.expval_child() is None          # (nothing folded under NO_ABSORB)
    var = plan.executed("state")                # the from-|0..0> variant when the model preferred one
    assert var is plan or var.describe()["zero_run"]
    if n >= 18:   # (small registers may keep the plain schedule: the pass-cost model decides)
        assert var is not plan and var.describe()["stages"][0]["T"] == 14
        with pytest.raises(NotImplementedError):  # a from-zero schedule is not one for live states
            N.apply_inplace(var, angd, torch.zeros((2, 1 << n), dtype=torch.complex64, device="cuda"))
    got = plan.run(angd, "state")                      # from |0..0>: the variant
    st = torch.zeros((2, 1 << n), dtype=torch.complex64, device="cuda")
    st[:, 0] = 1
    N.apply_inplace(plan, angd, st)                    # live state: the plan's own schedule
    assert float((torch.view_as_real(got) - torch.view_as_real(st)).abs().max()) < 1e-6
    ez = plan.run(angd, "expval", list(range(n))).cpu().numpy()
    p = (st.abs() ** 2).double().cpu().numpy()
    idx = np.arange(1 << n)
    want_z = np.stack([[np.sum(pb * (1 - 2 * ((idx >> (n - 1 - q)) & 1))) for q in range(n)] for pb in p])
    assert np.abs(ez - want_z).max() < 1e-6
    if n <= 18:
        from tests.helpers import oracle_tape
        for b in range(2):
            tape = [(g, w, tuple(float(ang[b, s]) for s in sl)) for g, w, sl, _ in ops]
            want = OE.simulate_pure(tape, n, np.complex128)
            assert np.abs(got[b].cpu().numpy() - want).max() < 1e-6


@pytest.mark.parametrize("n", [14, 16])
def test_low_control_streaming_mode_every_control_target_pair(n):
    """k_direct_1q mode 7: non-diagonal controlled gates with the control on bit
    position 0..3 and the target on position 1..6, BOTH sides of the control (per-lane control
    branch with the target above and below it), one gate per launch, against the oracle."""
    from qml_essentials_amd import _native as N

    rng = np.random.default_rng(70 + n)
    prefix = [("RY", [q], (float(rng.uniform(0, 6.28)),)) for q in range(n)]
    prefix += [("CX", [q, (q + 1) % n], ()) for q in range(n - 1)]
    prefix += [("RX", [q], (float(rng.uniform(0, 6.28)),)) for q in range(n)]
    flags = N.plan_flags(no_fusion=True, force_global=True)
    for pc in range(0, 4):
        for pt in range(1, 7):
            if pt == pc:
                continue
            c, t = n - 1 - pc, n - 1 - pt
            for gate in (("CX", [c, t], ()), ("CRX", [c, t], (0.9,)), ("CRY", [c, t], (1.3,)),
                         ("CY", [c, t], ())):
                tape = prefix + [gate]
                got, plan = _run(tape, n, "state", flags=flags)
                kinds = [st["kind"] for st in plan.describe()["stages"]]
                assert kinds[-1] == "direct", (gate, pc, pt, kinds[-1])  # streamed, not a tile pass
                want = OE.simulate_pure(tape, n, np.complex128)
                err = np.abs(got[0] - want).max()
                assert err < 1e-6, (gate, pc, pt, err)


@pytest.mark.parametrize("burst", ["0", "9"])
def test_controlled_one_gate_passes_every_control_target_pair(burst, monkeypatch):
    """k_direct_1q modes 2 / 3 / 4 / 7 / 8 (round 4): every (control, target) position pair of a 16-qubit
    register for CRX and CX, one gate per launch, against the oracle -- with the 4-rows-per-stream burst
    form (mode 8: control and target on positions >= 9) switched off and switched on from position 9 (at
    n = 28 the library picks it by measured rules on positions this register does not have; the switch
    is read per launch)."""
    from qml_essentials_amd import _native as N

    monkeypatch.setenv("QMLE_K1_CTRL_BURST", burst)
    n = 16
    rng = np.random.default_rng(161)
    prefix = [("RY", [q], (float(rng.uniform(0, 6.28)),)) for q in range(n)]
    prefix += [("CX", [q, (q + 1) % n], ()) for q in range(n - 1)]
    prefix += [("RX", [q], (float(rng.uniform(0, 6.28)),)) for q in range(n)]
    flags = N.plan_flags(no_fusion=True, force_global=True)
    worst = 0.0
    for pc in range(n):
        for pt in range(n):
            if pt == pc or not (pc <= 3 or pc >= 9 or (pc + pt) % 3 == 0):
                continue  # (every low / high control; a third of the middle ones)
            c, t = n - 1 - pc, n - 1 - pt
            gate = ("CRX", [c, t], (0.9,)) if (pc + pt) % 2 else ("CX", [c, t], ())
            tape = prefix + [gate]
            got, plan = _run(tape, n, "state", flags=flags)
            want = OE.simulate_pure(tape, n, np.complex128)
            err = np.abs(got[0] - want).max()
            assert err < 1e-6, (gate, pc, pt, err)
            worst = max(worst, err)
    assert worst > 0.0


@pytest.mark.parametrize("n", [3, 5, 14, 16])
def test_controlled_phase_one_gate_passes_every_control_target_pair(n):
    """k_direct_1q mode 9 (round 5): CZ / ControlledPhaseShift touch the |11> quarter only when control
    and target both sit on chunk bits (positions >= 1); CRZ keeps the control = 1 half (mode 2).  Every
    (control, target) position pair, one gate per launch, against the oracle."""
    from qml_essentials_amd import _native as N

    rng = np.random.default_rng(900 + n)
    prefix = [("RY", [q], (float(rng.uniform(0, 6.28)),)) for q in range(n)]
    prefix += [("CX", [q, (q + 1) % n], ()) for q in range(n - 1)]
    prefix += [("RX", [q], (float(rng.uniform(0, 6.28)),)) for q in range(n)]
    flags = N.plan_flags(no_fusion=True, force_global=True)
    worst = 0.0
    for pc in range(n):
        for pt in range(n):
            if pt == pc or (n == 16 and not (pc <= 1 or pt <= 1 or (pc + pt) % 4 == 0)):
                continue
            c, t = n - 1 - pc, n - 1 - pt
            gate = (("CZ", [c, t], ()), ("CPhase", [c, t], (0.7,)), ("CRZ", [c, t], (1.1,)))[(pc + 2 * pt) % 3]
            tape = prefix + [gate]
            got, plan = _run(tape, n, "state", flags=flags)
            # (registers that fit one LDS tile run as a tile pass whatever the flags; so do diagonal gates whose
            # control sits on positions 1..3, inside every 128-byte line: qmle_plan.cpp direct_ok)
            if n >= 14 and (pc == 0 or pc >= 4):
                assert plan.describe()["stages"][-1]["kind"] == "direct"
            want = OE.simulate_pure(tape, n, np.complex128)
            err = np.abs(got[0] - want).max()
            assert err < 1e-6, (gate, pc, pt, err)
            worst = max(worst, err)
    assert worst > 0.0


@pytest.mark.parametrize("n", [4, 9, 13, 14])
def test_every_wire_every_single_gate_kind(n):
    """One gate per circuit on a random state prefix: exercises every target /
    control position (incl. the in-chunk bit 0 = last wire) of every kernel mode."""
    rng = np.random.default_rng(n)
    prefix = [("RY", [q], (float(rng.uniform(0, 6.28)),)) for q in range(n)]
    prefix += [("CX", [q, (q + 1) % n], ()) for q in range(n - 1)]
    prefix += [("RX", [q], (float(rng.uniform(0, 6.28)),)) for q in range(n)]
    cases = []
    for q in range(n):
        cases += [("RX", [q], (1.234,)), ("RZ", [q], (0.77,)), ("H", [q], ())]
    pairs = [(0, n - 1), (n - 1, 0), (1, 2), (2, 1), (n - 2, n - 1), (n - 1, n - 2), (0, 1)]
    for c, t in pairs:
        cases += [("CX", [c, t], ()), ("CRX", [c, t], (0.9,)), ("CRZ", [c, t], (2.1,)),
                  ("CZ", [c, t], ()), ("CPhase", [c, t], (0.4,)), ("RZX", [c, t], (1.1,)),
                  ("SWAP", [c, t], ())]
    cases += [("CCX", [0, n - 1, 1], ()), ("CSWAP", [n - 1, 0, 2], ())]
    modes = _modes(n)
    for gate in cases:
        tape = prefix + [gate]
        want = OE.simulate_pure(tape, n, np.complex128)
        for mode, flags in modes.items():
            if mode == "lds" and n > 14:
                continue
            got, _ = _run(tape, n, "state", flags=flags)
            err = np.abs(got[0] - want).max()
            assert err < 1e-6, (mode, gate, err)


def test_batched_angles_share_one_plan():
    """jax.vmap semantics (script.py:302-315): row b of the angle table drives sample b."""
    n, B = 6, 37
    rng = np.random.default_rng(5)
    tape = random_tape(n, 30, rng)
    ops, angles, consts = tape_to_native(tape, n)
    table = rng.uniform(0, 2 * np.pi, (B, len(angles))).astype(np.float32)
    for flags in _modes(n).values():
        got, _ = _run(tape, n, "state", flags=flags, batch_angles=table)
        for b in (0, 7, 36):
            it = iter(table[b])
            tb = [(nm, w, tuple(float(next(it)) for _ in p)) for nm, w, p in tape]
            want = OE.simulate_pure(tb, n, np.complex128)
            assert np.abs(got[b] - want).max() < 1e-6


@pytest.mark.parametrize("n", [3, 7, 12])
def test_measurements_probs_expval_density(n):
    rng = np.random.default_rng(n)
    tape = random_tape(n, 25, rng)
    psi = OE.simulate_pure(tape, n, np.complex128)
    obs = list(range(n))[::-1] + [0]
    want_e = OE.measure_state(psi, n, "expval", [("PauliZ", [w]) for w in obs])
    for mode, flags in _modes(n).items():
        p, _ = _run(tape, n, "probs", flags=flags)
        assert np.abs(p[0] - np.abs(psi) ** 2).max() < 1e-6, mode
        assert abs(p[0].sum() - 1) < 1e-5  # test_jaqsi.py:836-859
        e, _ = _run(tape, n, "expval", obs=obs, flags=flags)
        assert np.abs(e[0] - want_e).max() < 1e-6, mode
    if n <= 7:
        rho, _ = _run(tape, n, "density")
        assert np.abs(rho[0] - np.outer(psi, psi.conj())).max() < 1e-6


def test_constant_matrix_and_golomb_ops():
    n = 5
    rng = np.random.default_rng(2)
    u1 = np.linalg.qr(rng.normal(size=(2, 2)) + 1j * rng.normal(size=(2, 2)))[0]
    u2 = np.linalg.qr(rng.normal(size=(4, 4)) + 1j * rng.normal(size=(4, 4)))[0]
    tape = [("H", [q], ()) for q in range(n)]
    tape += [("Matrix", [3], (u1,)), ("Matrix", [4, 1], (u2,)), ("Golomb", [], (0.37,)),
             ("RY", [2], (0.3,)), ("Matrix", [0, 4], (u2,))]
    want = OE.simulate_pure(oracle_tape(tape, n), n, np.complex128)
    from qml_essentials_amd import _native as N

    for mode, flags in {"lds": 0, "global": N.plan_flags(force_global=True, tile_bits=4,
                                                           low_bits=1)}.items():
        got, _ = _run(tape, n, "state", flags=flags)
        # Golomb phases are evaluated in fp32 like the reference (unitary.py:694)
        assert np.abs(got[0] - want).max() < 5e-6, mode


def test_standalone_analysis_kernels():
    from qml_essentials_amd import _native as N

    n, B = 9, 10
    rng = np.random.default_rng(11)
    st = rng.normal(size=(B, 2**n)) + 1j * rng.normal(size=(B, 2**n))
    st /= np.linalg.norm(st, axis=1, keepdims=True)
    dev = torch.from_numpy(st.astype(np.complex64)).cuda()
    assert np.abs(N.probs(dev).cpu().numpy() - np.abs(st) ** 2).max() < 1e-7
    wires = [0, 4, 8, 3]
    want = np.stack([OE.measure_state(s, n, "expval", [("PauliZ", [w]) for w in wires]) for s in st])
    assert np.abs(N.expval_z(dev, wires).cpu().numpy() - want).max() < 1e-6
    f = N.pair_fidelity(dev).cpu().numpy()
    assert np.abs(f - OA.fidelities_pure(st, B // 2)).max() < 1e-6
    q, pur = N.meyer_wallach(dev, return_purities=True)
    want_q = np.array([OA.meyer_wallach_pure(s, n) for s in st])
    assert np.abs(q.cpu().numpy() - want_q).max() < 1e-6
    want_p = np.stack([OA.qubit_purities_pure(s, n) for s in st])
    assert np.abs(pur.cpu().numpy() - want_p).max() < 1e-6
    for keep in ([0], [8, 2], [5, 1, 7]):
        m = N.marginal_probs(dev, keep).cpu().numpy()
        assert np.abs(m - OA.marginalize_probs(np.abs(st) ** 2, n, keep)).max() < 1e-6
    rho = N.density(dev[:2, : 2**5].contiguous() * 1.0).cpu().numpy()
    assert rho.shape == (2, 32, 32)


def test_histogram_matches_numpy():
    from qml_essentials_amd import _native as N

    rng = np.random.default_rng(0)
    v = rng.random(5000).astype(np.float32)
    v[:5] = [0.0, 1.0, 0.5, 1.0 / 75, 74.0 / 75]
    for nb in (4, 75, 300):
        want, _ = np.histogram(v, bins=np.linspace(0, 1, nb + 1))
        got = N.histogram(torch.from_numpy(v).cuda(), nb).cpu().numpy()
        assert np.array_equal(got, want), nb


def test_large_state_direct_and_tiled_agree_n22():
    """Size-independent property at a size the oracle cannot reach quickly:
    fused-tile path == unfused direct path; norm preserved; <Z> consistent."""
    from qml_essentials_amd import _native as N
    from tests.test_abi_cpu import he_layer_ops

    n = 22
    ops, slots = he_layer_ops(n)
    rng = np.random.default_rng(1000)
    ang = torch.from_numpy(rng.uniform(0, 2 * np.pi, (1, slots)).astype(np.float32)).cuda()
    a = N.Plan(ops, n, slots).run(ang, "state")
    b = N.Plan(ops, n, slots, flags=N.plan_flags(no_fusion=True)).run(ang, "state")
    assert float((a - b).abs().max()) < 1e-6
    assert abs(float((a.abs() ** 2).sum()) - 1) < 1e-4
    e1 = N.Plan(ops, n, slots).run(ang, "expval", list(range(n)))
    e2 = N.expval_z(b, list(range(n)))
    assert float((e1 - e2).abs().max()) < 1e-5


@pytest.mark.parametrize("n", list(range(12, 25)))
def test_meyer_wallach_lds_tile_path(n):
    """n >= 12 takes the LDS-staged read kernels (1 + ceil((n-12)/8) reads): every wire's purity
    against the oracle for EVERY size from 12 to 24 -- one read (12), one later read with an
    already reported lower chunk (13..16), overlapping chunks (17..19, 21..23), the outermost /
    innermost pairing and the odd chunk (21..24)."""
    from qml_essentials_amd import _native as N

    rng = np.random.default_rng(n)
    B = 3 if n <= 18 else 1
    assert N.mw_reads(n) == 1 + (n - 12 + 7) // 8
    st = rng.normal(size=(B, 2**n)) + 1j * rng.normal(size=(B, 2**n))
    st /= np.linalg.norm(st, axis=1, keepdims=True)
    # make it less uniform: entangle-ish structure via a phase ramp and amplitude decay
    st *= np.exp(-np.arange(2**n) / 2**n)[None, :]
    # ... and wire-dependent: a few controlled sign flips / swaps of halves
    idx = np.arange(2**n)
    for k in range(0, n, 3):
        st[:, (idx >> k) & 1 == 1] *= (1.0 + 0.15 * k)
    st /= np.linalg.norm(st, axis=1, keepdims=True)
    dev = torch.from_numpy(st.astype(np.complex64)).cuda()
    q, pur = N.meyer_wallach(dev, return_purities=True)
    want_p = np.stack([OA.qubit_purities_pure(s, n) for s in st])
    assert np.abs(pur.cpu().numpy() - want_p).max() < 1e-6
    want_q = np.array([OA.meyer_wallach_pure(s, n) for s in st])
    assert np.abs(q.cpu().numpy() - want_q).max() < 1e-6


@pytest.mark.parametrize("n", [3, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24])
def test_meyer_wallach_out_of_the_producing_pass(n, monkeypatch):
    """QMLE_MEAS_MEYER_WALLACH: the plan's last tile pass reports its own tile's sums from LDS
    (tile_mw_row), the remaining positions come from later reads -- every wire's purity against the
    oracle's purities of the state the SAME plan stores (the state path has its own oracle tests), for
    Hardware-Efficient layers (fast tile kernel) and random tapes with controlled rotations / two-qubit
    Pauli rotations (generic tile kernel, streaming last stages that cannot be fused), with and
    without known-zero tracking, whole-state regime (n <= 14: no state is stored) and tiled."""
    from qml_essentials_amd import _native as N
    from tests.helpers import random_tape, tape_to_native
    from tests.test_abi_cpu import he_layer_ops

    # (round 5: tiled states fuse by default too -- lean epilogue + streaming stores of the producing pass;
    # QMLE_MW_FUSE_TILED=0 selects the stand-alone reads of the stored state)
    monkeypatch.delenv("QMLE_MW_FUSE_TILED", raising=False)
    rng = np.random.default_rng(100 + n)
    B = 3 if n <= 16 else 1
    cases = []
    for layers in (1, 2):
        ops, slots = [], 0
        for _ in range(layers):
            o, sl = he_layer_ops(n) if n >= 3 else ([], 0)
            ops += [(g, w, [x + slots for x in k], m) for g, w, k, m in o]
            slots += sl
        cases.append(("he%d" % layers, ops, slots, None))
    tape = random_tape(n, 3 * n, rng, three_q=n <= 12)
    r_ops, r_values, r_consts = tape_to_native(tape, n)
    cases.append(("random", r_ops, len(r_values), (r_consts, r_values)))
    for name, ops, slots, extra in cases:
        for flags in (0, N.PLAN_NO_SPARSE):
            consts = extra[0] if extra else None
            plan = N.Plan(ops, n, slots, consts=consts, flags=flags)
            if extra:
                ang = torch.from_numpy(np.tile(np.asarray(extra[1], dtype=np.float32), (B, 1))).cuda()
                ang = ang + torch.from_numpy(rng.uniform(0, 0.5, ang.shape).astype(np.float32)).cuda()
            else:
                ang = torch.from_numpy(rng.uniform(0, 2 * np.pi, (B, slots)).astype(np.float32)).cuda()
            got = plan.run(ang, "mw").cpu().numpy()
            assert got.shape == (B, n + 1)
            st = plan.run(ang, "state").cpu().numpy().astype(np.complex128)
            want_p = np.stack([OA.qubit_purities_pure(v, n) for v in st])
            want_q = 2 * (1 - want_p.mean(axis=1))
            assert np.abs(got[:, 1:] - want_p).max() < 1e-6, (name, flags, np.abs(got[:, 1:] - want_p).max())
            assert np.abs(got[:, 0] - want_q).max() < 1e-6, (name, flags)
            if n > 14:  # the stand-alone reads of the stored state: the same numbers
                monkeypatch.setenv("QMLE_MW_FUSE_TILED", "0")
                dflt = plan.run(ang, "mw").cpu().numpy()
                monkeypatch.delenv("QMLE_MW_FUSE_TILED")
                assert np.abs(dflt - got).max() < 1e-6, (name, flags)
                # round 5: by default the producing pass leaves positions 0..3 to the first later read
                # (TileArgs::mw_lean, k_mw_read_later_low); QMLE_MW_NO_LEAN=1 is the round-4 split
                monkeypatch.setenv("QMLE_MW_NO_LEAN", "1")
                full = plan.run(ang, "mw").cpu().numpy()
                monkeypatch.delenv("QMLE_MW_NO_LEAN")
                assert np.abs(full - got).max() < 1e-6, (name, flags)


@pytest.mark.parametrize("n,layers,flags", [(16, 2, 0), (18, 2, 128 | 32), (20, 1, 128 | 32)])
def test_plan_autotuner_keeps_results_and_reports_its_choice(n, layers, flags):
    """qmle_plan_autotune (opt-in): the model's best schedules timed on the device, the fastest kept.
    Whatever it picks, <Z> and the state stay what they were (the candidates are the same tape), the
    plan says so (`autotuned`), a second plan of the same tape adopts the remembered choice without
    timing, and plans with a single schedule come back untouched."""
    from qml_essentials_amd import _native as N
    from tests.test_abi_cpu import he_layer_ops

    ops, slots = [], 0
    for _ in range(layers):
        o, sl = he_layer_ops(n)
        ops += [(g, w, [x + slots for x in k], m) for g, w, k, m in o]
        slots += sl
    ang = torch.from_numpy(np.random.default_rng(n).uniform(0, 2 * np.pi, (4, slots)).astype(np.float32)).cuda()
    plan = N.Plan(ops, n, slots, flags=flags)
    z0 = plan.run(ang, "expval", list(range(n)))
    s0 = plan.run(ang, "state")
    rep = plan.autotune("expval", n, batch=4, top_k=4, reps=2)
    assert rep["candidate"] >= 0 and rep["ms_after"] <= rep["ms_before"] * 1.0 + 1e-9
    exe = plan.executed("expval")
    assert exe.describe()["autotuned"] is True and exe.describe()["candidate"] == rep["candidate"]
    z1 = plan.run(ang, "expval", list(range(n)))
    assert float((z0 - z1).abs().max()) < 1e-6
    rep_s = plan.autotune("state", 0, batch=4, top_k=4, reps=2)
    s1 = plan.run(ang, "state")
    assert float((torch.view_as_real(s0) - torch.view_as_real(s1)).abs().max()) < 1e-6
    again = N.Plan(ops, n, slots, flags=flags)
    rep2 = again.autotune("expval", n, batch=4, top_k=4, reps=2)
    assert rep2["candidate"] == rep["candidate"] and rep2["padding"] == rep["padding"] and rep2["ms_before"] == 0.0
    assert float((again.run(ang, "expval", list(range(n))) - z0).abs().max()) < 1e-6
    ops12, slots12 = he_layer_ops(12)
    small = N.Plan(ops12, 12, slots12)
    assert small.autotune("expval", 12, batch=4)["candidate"] == -1   # whole state in LDS: one schedule
    assert rep_s["candidate"] >= -1


def test_dense_4wire_operator_and_density_measurements():
    """QMLE_OP_MAT4 (16x16 on 4 wires, any wire order) against the dense oracle, in the
    whole-state and the tiled regime; diag / <Z> of a vectorised density matrix."""
    from oracle.dense import lift
    from qml_essentials_amd import _native as N

    rng = np.random.default_rng(21)
    n = 7
    M = rng.normal(size=(16, 16)) + 1j * rng.normal(size=(16, 16))
    M /= np.linalg.norm(M, 2)
    pre = [("RY", [q], (float(rng.uniform(0, 6.28)),)) for q in range(n)]
    pre += [("CX", [q, q + 1], ()) for q in range(n - 1)]
    psi = OE.simulate_pure(pre, n, np.complex128)
    blob = np.stack([M.real, M.imag], axis=-1).astype(np.float32).reshape(-1)
    for wires in ([0, 1, 2, 3], [6, 2, 0, 4], [3, 6, 5, 1]):
        want = lift(M, wires, n) @ psi
        ops, angles, consts = tape_to_native(pre, n)
        off = len(consts)
        ops = ops + [("MAT4", wires, [], off)]
        consts = np.concatenate([consts, blob])
        for flags in (0, N.plan_flags(force_global=True, tile_bits=5, low_bits=1),
                      N.plan_flags(force_global=True, no_fusion=True, tile_bits=6, low_bits=2)):
            plan = N.Plan(ops, n, len(angles), consts, flags)
            got = plan.run(torch.from_numpy(angles[None, :]).cuda(), "state").cpu().numpy()[0]
            assert np.abs(got - want).max() < 1e-6, (wires, flags)
    # density measurements on a random (non-physical is fine) vectorised matrix
    nq, B = 4, 3
    rho = (rng.normal(size=(B, 16, 16)) + 1j * rng.normal(size=(B, 16, 16))).astype(np.complex64)
    dev = torch.from_numpy(rho.reshape(B, -1)).cuda()
    pr = N.density_probs(dev, nq).cpu().numpy()
    assert np.abs(pr - np.real(np.einsum("bii->bi", rho))).max() < 1e-6
    ez = N.density_expval_z(dev, nq, [0, 3, 1]).cpu().numpy()
    idx = np.arange(16)
    for k, w in enumerate([0, 3, 1]):
        sign = 1 - 2 * ((idx >> (nq - 1 - w)) & 1)
        assert np.abs(ez[:, k] - (pr * sign).sum(axis=1)).max() < 1e-5


@pytest.mark.parametrize("n,B,tile_bits,low_bits", [(18, 40, 12, 4), (18, 33, 12, 7), (19, 17, 13, 5),
                                                     (20, 9, 12, 1)])
def test_prefetching_tile_kernel_matches_plain_tile_kernel(n, B, tile_bits, low_bits):
    """k_tile_pf (opt-in double-buffered LDS-DMA variant, one run of tiles per workgroup) against k_tile on the
    same plan geometry, bit for bit (same arithmetic in the same order), for every epilogue:
    state store, fused all-qubit <Z> partials, probabilities; batch sizes that leave ragged
    last chunks; and against the oracle for one sample."""
    from qml_essentials_amd import _native as N

    rng = np.random.default_rng(n * 7 + B)
    tape = random_tape(n, 60, rng)
    ops, angles, consts = tape_to_native(tape, n)
    table = rng.uniform(0, 2 * np.pi, size=(B, len(angles))).astype(np.float32)
    table[0] = angles
    ang = torch.from_numpy(table).cuda()
    res = {}
    for name, pf in (("pf", True), ("plain", False)):
        flags = N.plan_flags(force_global=True, force_tile=True, tile_bits=tile_bits,
                             low_bits=low_bits, prefetch=pf)
        plan = N.Plan(ops, n, len(angles), consts, flags)
        res[name] = (plan.run(ang, "state").cpu().numpy(),
                     plan.run(ang, "expval", list(range(n))).cpu().numpy(),
                     plan.run(ang, "probs").cpu().numpy())
        if name == "pf":
            kinds = [s["kind"] for s in plan.describe()["stages"]]
            assert kinds.count("tile") >= 2     # at least one pass loads its tiles
    for got, want in zip(res["pf"], res["plain"]):
        assert np.array_equal(got, want)
    psi = OE.simulate_pure(oracle_tape(tape, n), n, dtype=np.complex128)
    assert np.allclose(res["pf"][0][0], psi, atol=2e-6)
    assert np.allclose(res["pf"][2][0], np.abs(psi) ** 2, atol=1e-6)


def _tail_tape(n, rng, n_head, n_tail):
    """Random circuit followed by a tail of CX / SWAP / diagonal gates on random wires."""
    tape = random_tape(n, n_head, rng)
    for _ in range(n_tail):
        kind = rng.choice(["CX", "CX", "CX", "SWAP", "RZ", "CZ", "CRZ", "CPhase", "RZZ", "PauliZ", "S"])
        if kind in ("RZ", "PauliZ", "S"):
            w = [int(rng.integers(n))]
        else:
            w = [int(x) for x in rng.choice(n, size=2, replace=False)]
        tape.append((kind, w, (float(rng.uniform(0, 6.28)),) if kind in ("RZ", "CRZ", "CPhase", "RZZ") else ()))
    return tape


@pytest.mark.parametrize("n,flags_kw", [(5, {}), (10, {}), (14, {}), (16, {}), (18, {}),
                                        (18, dict(force_global=True, tile_bits=13, low_bits=5)),
                                        (9, dict(force_global=True, tile_bits=6, low_bits=3)),
                                        (20, {})])
def test_trailing_permutation_gates_fold_into_z_observables(n, flags_kw):
    """<Z> with the trailing CX / SWAP / diagonal gates folded into parity observables
    (Z_t -> Z_c Z_t) == the same plan simulating them (QMLE_PLAN_NO_ABSORB) == the oracle.
    Covers the whole-state kernel (n <= 14), the Walsh-Hadamard tile epilogue and the
    stand-alone parity kernels (tail-only circuits)."""
    from qml_essentials_amd import _native as N

    rng = np.random.default_rng(n)
    for n_head, n_tail in ((40, 12), (25, 40), (0, 15), (30, 0)):
        tape = _tail_tape(n, rng, n_head, n_tail)
        ops, angles, consts = tape_to_native(tape, n)
        B = 3
        table = rng.uniform(0, 2 * np.pi, size=(B, max(1, len(angles)))).astype(np.float32)
        if len(angles):
            table[0, :len(angles)] = angles
        ang = torch.from_numpy(table[:, :len(angles)] if len(angles) else table[:, :0]).cuda()
        obs = [int(w) for w in rng.permutation(n)[: max(1, n - 2)]]
        plan = N.Plan(ops, n, len(angles), consts, N.plan_flags(**flags_kw))
        ref = N.Plan(ops, n, len(angles), consts, N.plan_flags(no_absorb=True, **flags_kw))
        got = plan.run(ang, "expval", obs).cpu().numpy()
        want = ref.run(ang, "expval", obs).cpu().numpy()
        assert np.allclose(got, want, atol=2e-6), (n_head, n_tail, np.abs(got - want).max())
        d = plan.describe()
        if n_tail >= 12:
            assert d.get("absorbed_ops", 0) >= 1 and plan.expval_child() is not None
            assert plan.expval_child().stats()["n_ops"] + d["absorbed_ops"] == len(ops)
        if n_tail == 0 and n_head:
            pass  # (random heads may end in an absorbable gate by chance)
        if n <= 14:  # oracle for one sample
            want0 = OE.simulate_and_measure(oracle_tape(tape, n), n, "expval",
                                            [("PauliZ", [w]) for w in obs], dtype=np.complex128)
            assert np.allclose(got[0], want0, atol=2e-6)
        # other measurement types are untouched by the folding
        if n <= 16:
            assert np.array_equal(plan.run(ang, "probs").cpu().numpy(),
                                  ref.run(ang, "probs").cpu().numpy())


@pytest.mark.parametrize("n", [17, 20])
def test_parities_with_one_position_in_the_last_tile_take_the_33_sums_epilogue(n):
    """Z-parities that meet the last tile pass in at most one bit position (all other factors sit
    on outer positions = bits of the tile index) are measured by the single-bit epilogue with
    per-row signs (run_batch_masks: `semi_single`) instead of the general-mask one: same numbers
    as the general path (QMLE_NO_SEMI_SINGLE is read once per process, so the reference here is
    the state + stand-alone parity kernel)."""
    from qml_essentials_amd import _native as N
    from tests.test_abi_cpu import he_layer_ops

    ops, slots = he_layer_ops(n)
    rng = np.random.default_rng(170 + n)
    ang = torch.from_numpy(rng.uniform(0, 2 * np.pi, (3, slots)).astype(np.float32)).cuda()
    plan = N.Plan(ops, n, slots, flags=N.PLAN_NO_SPARSE | N.PLAN_NO_ABSORB)
    last = plan.describe()["stages"][-1]
    assert last["kind"] == "tile" and last["T"] < n
    tile_wires = sorted(n - 1 - b for b in last["bits"])
    outer_wires = sorted(set(range(n)) - set(tile_wires))
    assert len(outer_wires) >= 3
    groups = [[tile_wires[0], outer_wires[0]], [outer_wires[0], outer_wires[-1]],
              [tile_wires[-1], outer_wires[1], outer_wires[2]], [tile_wires[3]], outer_wires[:3]]
    got = plan.run_parity(ang, groups).cpu().numpy()
    states = plan.run(ang, "state")
    want = N.expval_parity(states, groups).cpu().numpy()
    assert np.allclose(got, want, atol=2e-6), np.abs(got - want).max()
    # a parity with two positions inside the last tile: the general-mask epilogue, same answer
    groups2 = groups + [[tile_wires[0], tile_wires[1]]]
    got2 = plan.run_parity(ang, groups2).cpu().numpy()
    want2 = N.expval_parity(states, groups2).cpu().numpy()
    assert np.allclose(got2, want2, atol=2e-6), np.abs(got2 - want2).max()


@pytest.mark.parametrize("n", [6, 13, 17, 19])
def test_run_batch_parity_matches_oracle_and_standalone_kernel(n):
    """qmle_run_batch_parity: Z-parity observables measured out of the last pass (with the
    trailing CX layer folded in) == state + stand-alone parity kernel == oracle (n <= 13)."""
    from qml_essentials_amd import _native as N

    rng = np.random.default_rng(100 + n)
    tape = _tail_tape(n, rng, 40, 10)
    ops, angles, consts = tape_to_native(tape, n)
    ang = torch.from_numpy(np.stack([angles, angles * 0.5]).astype(np.float32)).cuda()
    groups = [[0, n - 1], [1], list(range(n)), [2, 3, n - 2], [n // 2, 0]]
    plan = N.Plan(ops, n, len(angles), consts)
    got = plan.run_parity(ang, groups).cpu().numpy()
    states = N.Plan(ops, n, len(angles), consts, N.plan_flags(no_absorb=True)).run(ang, "state")
    want = N.expval_parity(states, groups).cpu().numpy()
    assert np.allclose(got, want, atol=2e-6), np.abs(got - want).max()
    if n <= 13:
        psi = OE.simulate_pure(oracle_tape(tape, n), n, dtype=np.complex128)
        idx = np.arange(2**n)
        for k, g in enumerate(groups):
            par = np.zeros_like(idx)
            for w in g:
                par ^= (idx >> (n - 1 - w)) & 1
            assert abs(got[0, k] - np.sum(np.abs(psi) ** 2 * (1 - 2 * par))) < 1e-6
    with pytest.raises(ValueError):
        plan.run_parity(ang, [[n]])


def _shallow_tapes(n, rng):
    """Circuits whose first passes meet amplitudes that are still exactly zero: every wire
    touched at most a few times, some never, controls on untouched wires, diagonal-only wires."""
    ang = lambda: (float(rng.uniform(0, 6.28)),)
    he = [(g, [q], ang()) for g in ("RY", "RZ", "RY") for q in range(n)]
    he += [("CX", [q, q + 1], ()) for q in range(0, n - 1, 2)]
    he += [("CX", [q, (q + 1) % n], ()) for q in range(1, n, 2)]
    half = sorted(int(w) for w in rng.choice(n, size=n // 2, replace=False))
    rest = [w for w in range(n) if w not in half]
    part = [("RX", [w], ang()) for w in half]
    part += [("CX", [rest[0], half[0]], ()), ("CRY", [half[1], rest[1]], ang()),
             ("RZ", [rest[2]], ang()), ("CZ", [rest[0], rest[2]], ()), ("PauliX", [rest[3]], ())]
    part += [("RY", [w], ang()) for w in half[: len(half) // 2]]
    diag = [("RZ", [q], ang()) for q in range(n)] + [("H", [n - 1], ()), ("CX", [n - 1, 0], ())]
    two = he + [(g, [q], ang()) for g in ("RX",) for q in range(n)]
    # bits 0..9 first, then 4 fresh bits, the top bits never: a product pass (k_tile_product) that
    # is the LAST stage has to store zeros for tiles its predecessor never wrote
    low_mid = [("RX", [n - 1 - b], ang()) for b in range(min(10, n - 5))]
    low_mid += [("RY", [n - 1 - b], ang()) for b in range(min(10, n - 5), min(14, n - 1))]
    return {"he": he, "partial": part, "diag_then_h": diag, "he_plus_layer": two, "empty": [],
            "low_then_mid": low_mid}


@pytest.mark.parametrize("n,tile_bits,low_bits", [(14, 8, 2), (15, 7, 3), (16, 12, 4), (16, 9, 1), (18, 12, 4),
                                                  (18, 10, 2), (17, 11, 4)])
def test_known_zero_amplitudes_are_skipped(n, tile_bits, low_bits):
    """Runs from |0..0> do not read, compute or store amplitudes that are provably zero
    (Stage::zero_in, compact grids, folded gates): results equal those of the dense run
    (QMLE_PLAN_NO_SPARSE) to float32 rounding for state / probs / <Z> / parities, with the same
    exact zeros, and match the oracle."""
    from qml_essentials_amd import _native as N

    rng = np.random.default_rng(n * 100 + tile_bits)
    junk = torch.full((3 << n,), 7.0, dtype=torch.complex64, device="cuda")  # recycled as state buffers
    del junk
    for name, tape in _shallow_tapes(n, rng).items():
        ops, angles, consts = tape_to_native(tape, n)
        B = 3
        table = rng.uniform(0, 2 * np.pi, size=(B, max(1, len(angles)))).astype(np.float32)
        table[0, : len(angles)] = angles
        ang = torch.from_numpy(table[:, : len(angles)].copy()).cuda()
        res = {}
        for mode in ("sparse", "dense"):
            flags = N.plan_flags(force_global=True, force_tile=True, tile_bits=tile_bits,
                                 low_bits=low_bits, no_sparse=(mode == "dense"))
            plan = N.Plan(ops, n, len(angles), consts, flags)
            masks = [[0], [1, n - 1], [0, n // 2, n - 1]]
            res[mode] = (plan.run(ang, "state").cpu().numpy(),
                         plan.run(ang, "probs").cpu().numpy(),
                         plan.run(ang, "expval", list(range(n))).cpu().numpy(),
                         plan.run_parity(ang, masks).cpu().numpy())
            if mode == "sparse" and name == "he":
                st = plan.describe()["stages"]
                assert len(st) >= 2 and st[0]["zero_in"] == (1 << n) - 1 and st[1]["zero_in"] != 0
        # gates acting on known zeros are folded (k_tile_product: first columns of the gate
        # groups; k_reg_measure<FOLD> / _mono in the measuring pass): same numbers up to float32
        # rounding -- and exactly the same zeros
        for got, want in zip(res["sparse"], res["dense"]):
            assert np.abs(got - want).max() < 1e-6, name
        assert np.array_equal(res["sparse"][0] == 0, res["dense"][0] == 0), name
        psi = OE.simulate_pure(oracle_tape(tape, n), n, dtype=np.complex128)
        assert np.abs(res["sparse"][0][0] - psi).max() < 1e-6, name
        # one state buffer recycled for every sample: regions a pass skipped hold the previous
        # sample's amplitudes and must never be read
        one = plan.run(ang, "expval", list(range(n)), states_in_flight=1).cpu().numpy()
        assert np.abs(one - res["dense"][2]).max() < 1e-6, name
        one = plan.run(ang, "probs", states_in_flight=1).cpu().numpy()
        assert np.abs(one - res["dense"][1]).max() < 1e-6, name


@pytest.mark.parametrize("n,tile_bits,low_bits,B", [(16, 12, 4, 5), (17, 11, 3, 3), (18, 13, 5, 2), (20, 12, 4, 70)])
def test_register_measuring_last_pass(n, tile_bits, low_bits, B):
    """k_reg_measure: a last pass whose gates share one register-tile group runs without LDS
    staging and accumulates the observables across tiles per work item.  Dense input (a
    full-register diagonal separates a deep head from a 4-wire tail) and known-zero input
    (shallow circuit), single-bit <Z> and parity observables, against the stored state (k_tile's
    store epilogue, itself oracle-checked above) and, for the shallow circuit, the fp64 oracle."""
    from qml_essentials_amd import _native as N

    rng = np.random.default_rng(n * 31 + B)
    ang = lambda: (float(rng.uniform(0, 6.28)),)
    last4 = [n - 1 - p for p in (tile_bits - 1, tile_bits - 2, tile_bits - 3, tile_bits - 4)]  # wires of 4 tile bits
    head = random_tape(n, 50, rng) + [("RY", [q], ang()) for q in range(n)]
    tail = [("RX", [w], ang()) for w in last4] + [("CRY", [last4[0], last4[1]], ang()),
                                                   ("CX", [last4[2], last4[3]], ()), ("RZ", [last4[1]], ang())]
    marks = rng.integers(0, 50, size=1 << n).astype(np.float32)
    x = 0.37
    ops1, ang1, c1 = tape_to_native(head, n)
    ops3, ang3, _ = tape_to_native(tail, n)
    base = len(ang1) + 1
    dense_ops = (ops1 + [("DIAG_ALL", [], [len(ang1)], len(c1))] +
                 [(nm, w, [s_ + base for s_ in sl], off) for nm, w, sl, off in ops3])
    dense_ang = np.concatenate([ang1, [x], ang3]).astype(np.float32)
    dense_consts = np.concatenate([c1, marks]).astype(np.float32)
    dense_oracle = None  # (the oracle applies a full-register diagonal as a 2^n x 2^n matrix)
    shallow = [(g_, [q], ang()) for g_ in ("RY", "RZ") for q in range(n)]
    sh_ops, sh_ang, sh_consts = tape_to_native(shallow, n)
    cases = {"dense": (dense_ops, dense_ang, dense_consts, dense_oracle),
             "shallow": (sh_ops, sh_ang, sh_consts, shallow)}
    if n >= 20:
        del cases["dense"]  # 2^20 marks: keep the oracle run short
    for name, (ops, angles, consts, otape) in cases.items():
        table = rng.uniform(0, 2 * np.pi, size=(B, len(angles))).astype(np.float32)
        table[0] = angles
        a_dev = torch.from_numpy(table).cuda()
        flags = N.plan_flags(force_global=True, force_tile=True, tile_bits=tile_bits, low_bits=low_bits,
                             no_absorb=True)
        plan = N.Plan(ops, n, len(angles), consts, flags)
        st = plan.describe()["stages"]
        if name == "dense":  # (the shallow plan ends in a single-group pass for n - T <= 4)
            assert st[-1]["kind"] == "tile" and st[-1]["lds_round_trips"] == 1 and len(st) >= 2, st[-1]
        masks = [[0], [n - 1], [1, n - 2], last4[:2], last4, [0, last4[3], n - 1], list(range(n))]
        ez = plan.run(a_dev, "expval", list(range(n))).cpu().numpy()
        par = plan.run_parity(a_dev, masks).cpu().numpy()
        psi = plan.run(a_dev, "state").cpu().numpy().astype(np.complex128)
        pr = np.abs(psi) ** 2
        idx = np.arange(1 << n)
        for w in range(n):
            sign = 1.0 - 2.0 * ((idx >> (n - 1 - w)) & 1)
            assert np.abs(ez[:, w] - pr @ sign).max() < 1e-6, (name, w)
        for k, mk in enumerate(masks):
            par_bits = np.zeros(1 << n, dtype=np.int64)
            for w in mk:
                par_bits ^= (idx >> (n - 1 - w)) & 1
            assert np.abs(par[:, k] - pr @ (1.0 - 2.0 * par_bits)).max() < 1e-6, (name, mk)
        if otape is not None:
            want = OE.simulate_and_measure(otape, n, "expval", [("PauliZ", [w]) for w in range(n)], np.complex128)
            assert np.abs(ez[0] - want).max() < 3e-6, name


def test_workspace_falls_back_to_fewer_states_in_flight_when_memory_is_short():
    """The engine asks for state buffers for up to 32 GiB of states per launch; with less free
    HBM the Python wrapper hands it a smaller workspace and the engine runs more chunks."""
    from qml_essentials_amd import _native as N

    n, B = 22, 1536                      # 32 MiB per state: the default wants 32 GiB
    ops, slots = _he_ops(n)
    plan = N.Plan(ops, n, slots)
    rng = np.random.default_rng(5)
    ang = torch.from_numpy(rng.uniform(0, 6.28, (B, slots)).astype(np.float32)).cuda()
    want = plan.run(ang[:8], "expval", list(range(n))).cpu().numpy()
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    hog = torch.empty(max(0, free - (6 << 30)), dtype=torch.uint8, device="cuda")  # leave ~6 GiB
    try:
        assert plan.workspace_bytes(B, "expval", n) > (6 << 30)
        got = plan.run(ang, "expval", list(range(n))).cpu().numpy()
    finally:
        del hog
        torch.cuda.empty_cache()
    assert got.shape == (B, n) and np.abs(got[:8] - want).max() < 1e-6
    assert np.isfinite(got).all() and np.abs(got).max() <= 1.0 + 1e-5


def _he_ops(n):
    from tests.test_abi_cpu import he_layer_ops

    return he_layer_ops(n)


@pytest.mark.parametrize("n", [10, 11, 12, 13])
def test_whole_state_expval_epilogue_masks(n):
    """k_tile2's whole-state <Z> epilogue (one LDS tile per sample, n = 10..13): single-wire and
    parity observables whose positions fall on the work item's slots (positions 0, n-3..n-1), on
    lane bits (1..6) and on the wave index (7..n-4), more than eight of them (two rounds of wave
    sums), against the fp64 oracle -- `simulation.measure_state` (`simulation.py:241-261`)."""
    from qml_essentials_amd import _native as N
    from tests.test_abi_cpu import he_layer_ops

    rng = np.random.default_rng(4100 + n)
    ops, slots = [], 0
    for _ in range(2):
        o, s = he_layer_ops(n)
        ops += [(g, w, [x + slots for x in sl], m) for g, w, sl, m in o]
        slots += s
    B = 5
    ang = rng.uniform(0, 2 * np.pi, (B, slots)).astype(np.float32)
    plan = N.Plan(ops, n, slots, flags=N.plan_flags(no_absorb=True))
    st0 = plan.describe()["stages"][0]
    assert plan.stats()["whole_state_lds"] == 1 and st0["fast"]
    # wire w <-> position n - 1 - w
    groups = [[w] for w in range(n)] + [[0, n - 1], [1, 2, n - 2], list(range(n)), [n - 1, n - 2, n - 3, n - 4],
                                        [0, 1, 2], [n - 8, n - 9], [3, n - 1], [n // 2]]
    if n >= 12:
        groups += [[n - 8], [n - 9, 0], [n - 8, n - 2, 1]]
    got = plan.run_parity(torch.from_numpy(ang).cuda(), groups).cpu().numpy()
    z = plan.run(torch.from_numpy(ang).cuda(), "expval", list(range(n))).cpu().numpy()
    idx = np.arange(2**n)
    for b in range(B):
        tape = [(g, w, [float(ang[b, s]) for s in sl]) for g, w, sl, _ in ops]
        psi = OE.simulate_pure(oracle_tape(tape, n), n, dtype=np.complex128)
        pr = np.abs(psi) ** 2
        for k, g in enumerate(groups):
            par = np.zeros_like(idx)
            for w in g:
                par ^= (idx >> (n - 1 - w)) & 1
            assert abs(got[b, k] - np.sum(pr * (1 - 2 * par))) < 1e-6, (b, k, g)
        assert np.allclose(z[b], got[b, :n], atol=1e-7)


def test_last_stage_padding_switch_changes_the_tile_not_the_result(monkeypatch):
    """QMLE_PAD_HIGH (tuning switch): the last stage's spare tile positions come from the
    top -- another tile, the same <Z> as the default schedule and as the C-side oracle path."""
    from qml_essentials_amd import _native as N
    from tests.test_abi_cpu import he_layer_ops

    n = 20
    ops, slots = he_layer_ops(n)
    ang = torch.from_numpy(np.random.default_rng(77).uniform(0, 2 * np.pi, (3, slots)).astype(np.float32)).cuda()
    flags = N.PLAN_NO_SPARSE | N.PLAN_NO_ABSORB
    base = N.Plan(ops, n, slots, flags=flags)
    want = base.run(ang, "expval", list(range(n))).cpu().numpy()
    tiles = {}
    for v in ("1", "8"):
        monkeypatch.setenv("QMLE_PAD_HIGH", v)
        p = N.Plan(ops, n, slots, flags=flags)
        tiles[v] = p.executed("expval").describe()["stages"][-1]["bits"]
        got = p.run(ang, "expval", list(range(n))).cpu().numpy()
        assert np.allclose(got, want, atol=2e-6), (v, np.abs(got - want).max())
    monkeypatch.delenv("QMLE_PAD_HIGH")
    default_tile = base.executed("expval").describe()["stages"][-1]["bits"]
    assert tiles["1"] != default_tile or tiles["8"] != default_tile
    tape = [(g, w, [float(ang[0, s]) for s in sl]) for g, w, sl, _ in ops]
    psi = OE.simulate_pure(oracle_tape(tape, n), n, dtype=np.complex128)
    pr = (np.abs(psi) ** 2).reshape((2,) * n)
    z = [float(pr.take(0, axis=w).sum() - pr.take(1, axis=w).sum()) for w in range(n)]
    assert np.allclose(want[0], z, atol=2e-6)


@pytest.mark.parametrize("layers", [2, 3])
def test_multi_tile_walk_over_known_zeros_inside_the_tile(layers, monkeypatch):
    """Default engine (known-zero tracking), n = 22: stages whose tile still holds known-zero
    positions while every tile is live run as multi-tile walks whose loads skip the zeros
    (round 3).  Same state, bit for bit, as with one tile per workgroup (QMLE_NO_MULTI_ZIN=1, the
    round-2 launch), <Z> within float32 summation order, the all-live plan's state at float32 level."""
    from qml_essentials_amd import _native as N
    from tests.test_abi_cpu import he_layer_ops

    n, B = 22, 32  # (>= 5120 workgroups after halving the grid: launch_tile's condition for a walk)
    ops, slots = [], 0
    for _ in range(layers):
        o, s_ = he_layer_ops(n)
        ops += [(g, w, [x + slots for x in sl], m) for g, w, sl, m in o]
        slots += s_
    ang = torch.from_numpy(np.random.default_rng(31 + layers).uniform(0, 2 * np.pi, (B, slots)).astype(np.float32)).cuda()
    plan = N.Plan(ops, n, slots)
    zin_stage = [s for s in plan.describe()["stages"][1:] if s["kind"] == "tile" and s["zero_in"]
                 and all((s["zero_in"] >> b) & 1 == 0 for b in range(n) if b not in s["bits"])]
    assert zin_stage, "no stage with known zeros inside the tile only: pick another circuit"
    walk = plan.run(ang, "state")
    z_walk = plan.run(ang, "expval", list(range(n)))
    monkeypatch.setenv("QMLE_NO_MULTI_ZIN", "1")
    single = plan.run(ang, "state")
    z_single = plan.run(ang, "expval", list(range(n)))
    monkeypatch.delenv("QMLE_NO_MULTI_ZIN")
    assert torch.equal(walk, single)
    # (<Z>: the walk sums a workgroup's tiles in registers before the row reduction -- another order)
    assert (z_walk - z_single).abs().max().item() < 5e-7
    dense = N.Plan(ops, n, slots, flags=N.PLAN_NO_SPARSE).run(ang[:4], "state")
    assert (walk[:4] - dense).abs().max().item() < 1e-6


def test_every_schedule_candidate_gives_the_same_result(monkeypatch):
    """All 60 schedule candidates of the plan compiler (QMLE_FORCE_CAND; tile geometry x lazy CX x
    wide first tile / carried position 6 / first tile on the top positions), with and without the last-stage padding switch, on one
    3-layer circuit at n = 18: state and <Z> equal the model-chosen schedule's at float32 level and
    the fp64 oracle's <Z> -- the rarely chosen geometries (L = 5..7 rows, T = 13 with a carried
    position) run the same kernels as the common ones."""
    from qml_essentials_amd import _native as N
    from tests.test_abi_cpu import he_layer_ops

    n, B = 18, 3
    ops, slots = [], 0
    for _ in range(3):
        o, s_ = he_layer_ops(n)
        ops += [(g, w, [x + slots for x in sl], m) for g, w, sl, m in o]
        slots += s_
    ang = torch.from_numpy(np.random.default_rng(91).uniform(0, 2 * np.pi, (B, slots)).astype(np.float32)).cuda()
    flags = N.PLAN_NO_SPARSE | N.PLAN_NO_ABSORB
    ref = N.Plan(ops, n, slots, flags=flags)
    want_s = ref.run(ang, "state")
    want_z = ref.run(ang, "expval", list(range(n)))
    tape = [(g, w, [float(ang[0, s]) for s in sl]) for g, w, sl, _ in ops]
    psi = OE.simulate_pure(oracle_tape(tape, n), n, dtype=np.complex128)
    pr = (np.abs(psi) ** 2).reshape((2,) * n)
    z = np.array([pr.take(0, axis=w).sum() - pr.take(1, axis=w).sum() for w in range(n)])
    assert np.abs(want_z[0].cpu().numpy() - z).max() < 1e-6
    shapes, top_first = set(), set()
    for pad in (None, "1"):
        if pad:
            monkeypatch.setenv("QMLE_PAD_HIGH", pad)
        for k in range(60):  # (48..59, round 5: first tile on the TOP 14 positions, stored amplitude by amplitude)
            monkeypatch.setenv("QMLE_FORCE_CAND", str(k))
            p = N.Plan(ops, n, slots, flags=flags)
            d = p.executed("expval").describe()
            shapes.add(tuple(tuple(st["bits"]) for st in d["stages"]))
            if k >= 48:
                assert d["candidate"] == k and d["stages"][0]["shift"] == n - 14, (k, d["candidate"])
                assert d["stages"][0]["bits"] == list(range(n - 14, n))
                top_first.add(len(d["stages"]))
            assert (p.run(ang, "state") - want_s).abs().max().item() < 1e-6, (k, pad)
            assert (p.run(ang, "expval", list(range(n))) - want_z).abs().max().item() < 1e-6, (k, pad)
    assert len(shapes) >= 8 and top_first


@pytest.mark.parametrize("layers", [1, 2])
def test_default_engine_under_every_tile_geometry(layers, monkeypatch):
    """The default engine (known-zero tracking, observable folding) under each of its 12 schedule
    candidates (QMLE_FORCE_CAND 0..11: six tile geometries x lazy CX) at n = 18: state, <Z> and a
    parity observable equal the all-live plan's -- its special first-pass kernels are chosen per
    stage shape, and every shape must reach the same answer."""
    from qml_essentials_amd import _native as N
    from tests.test_abi_cpu import he_layer_ops

    n, B = 18, 4
    ops, slots = [], 0
    for _ in range(layers):
        o, s_ = he_layer_ops(n)
        ops += [(g, w, [x + slots for x in sl], m) for g, w, sl, m in o]
        slots += s_
    ang = torch.from_numpy(np.random.default_rng(17 + layers).uniform(0, 2 * np.pi, (B, slots)).astype(np.float32)).cuda()
    dense = N.Plan(ops, n, slots, flags=N.PLAN_NO_SPARSE | N.PLAN_NO_ABSORB)
    want_s = dense.run(ang, "state")
    want_z = dense.run(ang, "expval", list(range(n)))
    groups = [[0, n - 1], [3, 4, 5], list(range(n))]
    want_p = dense.run_parity(ang, groups)
    for k in range(12):
        monkeypatch.setenv("QMLE_FORCE_CAND", str(k))
        p = N.Plan(ops, n, slots)
        assert (p.run(ang, "state") - want_s).abs().max().item() < 1e-6, k
        assert (p.run(ang, "expval", list(range(n))) - want_z).abs().max().item() < 1e-6, k
        assert (p.run_parity(ang, groups) - want_p).abs().max().item() < 1e-6, k


def test_batches_longer_than_one_launch_row_limit():
    """More than 65 535 samples per call (the grid's y extent; reference loops have no such bound:
    `script.py:399-553` only chunks for memory): every engine entry point and stand-alone analysis
    wrapper cuts the batch itself -- rows around the seam equal the same rows computed alone."""
    from qml_essentials_amd import _native as N
    from tests.test_abi_cpu import he_layer_ops

    n, B = 4, 65540
    ops, slots = he_layer_ops(n)
    rng = np.random.default_rng(5)
    ang = torch.from_numpy(rng.uniform(0, 6.28, (B, slots)).astype(np.float32)).cuda()
    rows = [0, 1, 65534, 65535, 65536, 65539]
    plan = N.Plan(ops, n, slots)
    sub_ang = ang[rows].contiguous()
    st, st_sub = plan.run(ang, "state"), plan.run(sub_ang, "state")
    assert torch.equal(st[rows], st_sub)
    assert torch.equal(plan.run(ang, "expval", list(range(n)))[rows], plan.run(sub_ang, "expval", list(range(n))))
    assert torch.equal(plan.run(ang, "probs")[rows], plan.run(sub_ang, "probs"))
    assert torch.equal(plan.run_parity(ang, [[0, 3], [1]])[rows], plan.run_parity(sub_ang, [[0, 3], [1]]))
    q, pur = N.meyer_wallach(st, return_purities=True)
    q_sub, pur_sub = N.meyer_wallach(st_sub, return_purities=True)
    assert q.shape == (B,) and torch.equal(q[rows], q_sub) and torch.equal(pur[rows], pur_sub)
    for fn in (lambda s: N.expval_z(s, [0, 2, 3]), lambda s: N.marginal_probs(s, [1, 3]), N.probs, N.density,
               lambda s: N.expval_parity(s, [[0, 1], [2]])):
        assert torch.equal(fn(st)[rows], fn(st_sub))
    assert torch.equal(N.overlap(st, st.flip(0))[rows], N.overlap(st_sub, st.flip(0)[rows].contiguous()))
    # pairs (i, i + S) with S > 65535
    big = torch.cat([st, st[:B]])[: 2 * 65538]
    S = big.shape[0] // 2
    fid = N.pair_fidelity(big)
    pick = [0, 65534, 65535, 65536, S - 1]
    want = torch.stack([(big[i].conj() * big[i + S]).sum().abs() ** 2 for i in pick])
    assert fid.shape == (S,) and (fid[pick] - want).abs().max().item() < 1e-6
    # complex128 engine
    o64 = plan.run64(ang.double(), "expval", [[q_] for q_ in range(n)])
    assert o64.shape == (B, n) and (o64[rows].float() - plan.run(sub_ang, "expval", list(range(n)))).abs().max().item() < 1e-6


def test_device_parameter_sampler_is_numpys_stream_bit_for_bit():
    """qmle_philox_uniform_f32_device == numpy.random.Generator(Philox(key)).uniform(...).astype(float32)
    == the host-side qmle_philox_uniform_f32, for lengths around the block boundaries, spawned keys and
    other ranges; `utils.uniform` (the sampler of `Model.initialize_params`, `model.py:687-693`) draws
    the same parameters through either."""
    import os

    from qml_essentials_amd import _native as N
    from qml_essentials_amd import utils

    for seed in (0, 1000, 2**40 + 3):
        for n in (1, 2, 3, 4, 5, 1023, 1024, 1025, 73728, 262147, (1 << 20) + 1):
            state = np.random.SeedSequence(seed).generate_state(2, np.uint64)
            ref = np.random.Generator(np.random.Philox(np.random.SeedSequence(seed))).uniform(0, 2 * np.pi, n).astype(np.float32)
            got = N.philox_uniform_device(state, n, 0.0, 2 * np.pi).cpu().numpy()
            assert np.array_equal(ref, got), (seed, n)
            assert np.array_equal(got, N.philox_uniform(state, n, 0.0, 2 * np.pi))
    child = np.random.SeedSequence(5).spawn(3)[2]
    ref = np.random.Generator(np.random.Philox(child)).uniform(-1.5, 3.25, 40001).astype(np.float32)
    again = np.random.SeedSequence(entropy=child.entropy, spawn_key=child.spawn_key)
    assert np.array_equal(ref, N.philox_uniform_device(again.generate_state(2, np.uint64), 40001, -1.5, 3.25).cpu().numpy())
    k = utils.key(7).split(3)[1]
    a = utils.uniform(k, (2048, 3, 12), 0.0, 2 * np.pi)          # device route (>= 16384 values)
    os.environ["QMLE_HOST_SAMPLER"] = "1"
    try:
        b = utils.uniform(k, (2048, 3, 12), 0.0, 2 * np.pi)      # host route
    finally:
        del os.environ["QMLE_HOST_SAMPLER"]
    assert a.dtype == np.float32 and a.shape == (2048, 3, 12) and np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("offset", [0, 5, (1 << 33) + 3])
def test_angle_table_and_batch_behind_one_call_equal_the_two_calls(offset):
    """qmle_run_batch_map == qmle_build_angles + qmle_run_batch bit for bit (model.py:804-816: the affine
    angle map; model.py:1449-1481: rows of the cartesian inputs x params batch), with a batch offset in
    the 32-bit and in the 64-bit range of the row arithmetic; the table against the host formula."""
    import torch

    from qml_essentials_amd import _native as N
    from qml_essentials_amd.model import Model

    m = Model(6, 2, "Hardware_Efficient")
    rng = np.random.default_rng(11)
    # 90 samples: from 64 on qmle_run_batch_map forms the angles inside the matrix builder (no table)
    x = torch.from_numpy(rng.uniform(0, 6.28, (30, 1)).astype(np.float32)).cuda()
    p = torch.from_numpy(rng.uniform(0, 6.28, (3, *m.params.shape[-2:])).astype(np.float32)).cuda()
    cc, leaves, divs, mods, B = m._forward_device(p, x, None, "expval", False, _want_call=True)
    assert B == 90 and cc.n_slots > 0
    strides = [t[0].numel() for t in leaves]
    how, wires = cc._measure()
    assert how == "z"
    table = N.build_angles(leaves, strides, divs, mods, cc.d_ptr, cc.d_arg, cc.d_idx, cc.d_coef, cc.d_const,
                           cc.n_slots, B, offset, d_period=cc.d_period)
    two = cc.plan.run(table, "expval", wires)
    one = cc.run(leaves, divs, mods, B, offset)
    assert torch.equal(one, two)
    few = cc.run(leaves, divs, mods, 21, offset)  # < 64 samples: table + batch inside the one call
    assert torch.equal(few, two[:21])
    # the table itself: const + sum coef * leaf[row(b)][idx], rows from the flattened sample index
    ptr, arg, idx, coef = cc._map
    host = [t.cpu().numpy().reshape(t.shape[0], -1).astype(np.float64) for t in leaves]
    cst = cc.d_const.cpu().numpy().astype(np.float64)
    per = cc.d_period.cpu().numpy()
    want = np.zeros((B, cc.n_slots))
    for b in range(B):
        gb = b + offset
        for s in range(cc.n_slots):
            acc = cst[s]
            for t in range(ptr[s], ptr[s + 1]):
                k = arg[t]
                acc += float(coef[t]) * host[k][(gb // divs[k]) % mods[k], idx[t]]
            if per[s] > 0 and abs(acc) > per[s]:
                acc -= per[s] * np.rint(acc / per[s])
            want[b, s] = acc
    np.testing.assert_allclose(table.cpu().numpy(), want, rtol=0, atol=2e-6)
    if offset == 0:
        np.testing.assert_allclose(m(params=p, inputs=x).cpu().numpy().reshape(one.shape), one.cpu().numpy(), rtol=0, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("n,flags_kw", [(6, {}), (12, {}), (15, {}), (6, {"no_fusion": True, "force_global": True})])
def test_a_row_does_not_depend_on_its_batch(n, flags_kw):
    """Rows of a batch are independent circuits (script.py:302-327: vmap over the batch axis): the row of
    sample b must be the same bits whether it runs alone, at the end of 63 samples, or inside 64 / 65 / 130 --
    the matrix builder takes whole waves per gate group from 64 samples on and samples-then-groups items
    below (qmle_matrices.h), with a ragged last wave at 65 and 130; mixed 2x2 / 4x4 groups (RZX, SWAP)."""
    from qml_essentials_amd import _native as N

    rng = np.random.default_rng(640 + n)
    tape = [("RY", [q], (0.0,)) for q in range(n)] + [("CX", [q, (q + 1) % n], ()) for q in range(n)]
    tape += [("RZX", [0, n - 1], (0.0,)), ("SWAP", [1, 2], ()), ("Rot", [3], (0.0, 0.0, 0.0)), ("CRX", [2, 0], (0.0,))]
    tape += [("RZ", [q], (0.0,)) for q in range(n)]
    ops, angles, consts = tape_to_native(tape, n)
    plan = N.Plan(ops, n, len(angles), consts, N.plan_flags(**flags_kw) if flags_kw else 0)
    table = rng.uniform(-6.28, 6.28, (130, len(angles))).astype(np.float32)
    dev = torch.from_numpy(table).cuda()
    whole = plan.run(dev, "state").cpu().numpy()
    zs = plan.run(dev, "expval", list(range(n))).cpu().numpy()
    for B in (1, 63, 64, 65):
        part = plan.run(dev[:B].contiguous(), "state").cpu().numpy()
        assert np.array_equal(part, whole[:B]), B
        assert np.array_equal(plan.run(dev[:B].contiguous(), "expval", list(range(n))).cpu().numpy(), zs[:B]), B
    alone = plan.run(dev[129:130].contiguous(), "state").cpu().numpy()
    assert np.array_equal(alone[0], whole[129])
    row, k, filled = table[129], 0, []  # (tape_to_native: every float parameter takes the next slot)
    for g, w, prm in tape:
        filled.append((g, w, tuple(float(row[k + i]) for i in range(len(prm)))))
        k += len(prm)
    want = OE.simulate_pure(filled, n, np.complex128)
    assert np.abs(whole[129] - want).max() < 1e-6


@pytest.mark.parametrize("n", [2, 5, 9, 13, 16, 19])
def test_dense_4x4_operators_absorb_their_neighbours_and_run_in_register_tiles(n):
    """Round 5: an uncontrolled 4x4 (two-qubit Pauli rotations, SWAP, explicit matrices -- the Kraus
    superoperators of the density path) takes the 1-qubit gates before and behind it on its wires and the next
    4x4 on the same ordered pair into ONE per-sample matrix (plan compiler), and runs inside a register-tile
    group (GK_REG4X) instead of an LDS sweep of its own.  Tapes made of such runs, every engine mode, against
    the double-precision oracle; the batch rows differ in every angle."""
    from qml_essentials_amd import _native as N

    rng = np.random.default_rng(900 + n)
    two = ["RXX", "RYY", "RZZ", "RZX", "SWAP"]
    one = ["RX", "RY", "RZ", "H", "S", "Rot"]
    n_par = {"RX": 1, "RY": 1, "RZ": 1, "Rot": 3, "RXX": 1, "RYY": 1, "RZZ": 1, "RZX": 1}
    tape = []
    for _ in range(60):
        r = rng.random()
        if r < 0.45:
            g = two[rng.integers(len(two))]
            w = [int(x) for x in rng.choice(n, 2, replace=False)]
            if rng.random() < 0.5 and tape and len(tape[-1][1]) == 2:
                w = list(tape[-1][1]) if rng.random() < 0.7 else list(tape[-1][1])[::-1]  # same pair (either order)
        elif r < 0.55 and n >= 2:
            g, w = "CX", [int(x) for x in rng.choice(n, 2, replace=False)]
        else:
            g = one[rng.integers(len(one))]
            w = [int(rng.integers(n))]
        tape.append((g, w, tuple(float(x) for x in rng.uniform(0, 2 * np.pi, n_par.get(g, 0)))))
    want = OE.simulate_pure(tape, n, np.complex128)
    merged = None
    for mode, flags in _modes(n).items():
        got, plan = _run(tape, n, "state", flags=flags)
        err = np.abs(got[0] - want).max()
        assert err < 2e-6, (mode, n, err)
        if mode == "lds":
            merged = plan.describe()
    assert merged["n_lowered"] < 0.6 * merged["n_ops"], (merged["n_lowered"], merged["n_ops"])
    # a batch whose rows differ: per-sample 4x4 products
    ops, angles, consts = tape_to_native(tape, n)
    A = np.stack([angles, angles[::-1].copy(), rng.uniform(0, 6.28, angles.shape)]).astype(np.float32)
    plan = N.Plan(ops, n, len(angles), consts)
    z = plan.run(torch.from_numpy(A).cuda(), "expval", list(range(n))).cpu().numpy()
    for b in range(3):
        k, t2 = 0, []
        for g, w, p in tape:
            t2.append((g, w, tuple(float(x) for x in A[b, k:k + len(p)])))
            k += len(p)
        psi = OE.simulate_pure(t2, n, np.complex128)
        pr = np.abs(psi) ** 2
        idx = np.arange(1 << n)
        ref = [float((pr * (1 - 2 * ((idx >> (n - 1 - q)) & 1))).sum()) for q in range(n)]
        assert np.abs(z[b] - np.array(ref)).max() < 2e-6, (b, n)


@pytest.mark.parametrize("n", [16, 17, 19, 22, 25])
def test_first_tile_on_the_top_positions_every_register_size(n, monkeypatch):
    """Round 5: candidates 48..59 put the one tile a run from |0..0> computes per state on the TOP
    14 positions (amplitude j of the tile at address j << shift, one 8-byte store each, behind the zero fill) and
    schedule the rest on the low positions.  From the smallest register that admits it (shift = 2) upwards, two
    geometries each: the state and <Z> equal the round-4 schedule's (QMLE_NO_TOP_FIRST=1) bit for float32 rounding,
    the state equals the oracle's C port, and a batch of distinct parameter sets stays distinct."""
    from oracle import c_port
    from qml_essentials_amd import _native as N
    from tests.test_abi_cpu import he_layer_ops

    layers = 2 if n <= 22 else 1
    ops, slots = [], 0
    for _ in range(layers):
        o, s_ = he_layer_ops(n)
        ops += [(g, w, [x + slots for x in sl], m) for g, w, sl, m in o]
        slots += s_
    ops += [("CRX", [0, n - 1], [0], -1), ("RZ", [n // 2], [1], -1), ("H", [1], [], -1), ("CX", [n - 1, 2], [], -1)]
    B = 3 if n <= 22 else 2
    ang = np.random.default_rng(n).uniform(0, 2 * np.pi, (B, slots)).astype(np.float32)
    angd = torch.from_numpy(ang).cuda()
    flags = N.PLAN_NO_SPARSE | N.PLAN_NO_ABSORB
    monkeypatch.setenv("QMLE_NO_TOP_FIRST", "1")
    ref = N.Plan(ops, n, slots, flags=flags)
    assert all(st["shift"] == 0 for st in ref.executed("state").describe()["stages"])
    want_s, want_z = ref.run(angd, "state"), ref.run(angd, "expval", list(range(n)))
    monkeypatch.delenv("QMLE_NO_TOP_FIRST")
    assert (want_s[0] - want_s[1]).abs().max().item() > 1e-3
    if n <= 22:
        tape = [(g, w, [float(ang[1, s]) for s in sl]) for g, w, sl, _ in ops]
        psi = c_port.simulate(oracle_tape(tape, n), n)
        assert np.abs(want_s[1].cpu().numpy() - psi).max() < 1e-6
    for k in (50, 57):
        monkeypatch.setenv("QMLE_FORCE_CAND", str(k))
        p = N.Plan(ops, n, slots, flags=flags)
        d = p.executed("state").describe()
        assert d["candidate"] == k and d["stages"][0]["shift"] == n - 14 and len(d["stages"]) >= 2, (k, d["candidate"])
        assert (p.run(angd, "state") - want_s).abs().max().item() < 1e-6, (n, k)
        assert (p.run(angd, "expval", list(range(n))) - want_z).abs().max().item() < 1e-6, (n, k)
        monkeypatch.delenv("QMLE_FORCE_CAND")


@pytest.mark.parametrize("n,flags", [(16, 0), (18, 128 | 32), (20, 128)])
def test_chunks_of_a_batch_on_two_streams_give_the_one_stream_results(n, flags, monkeypatch):
    """Round 5: a batch that needs several chunks runs them alternately on two internal streams, one stage apart
    (own state / partial-sum buffers per stream; the caller's stream forks and joins).  Seven chunks of three states
    (states_in_flight = 3) for every measurement: equal to the one-stream loop (QMLE_NO_CHUNK_OVERLAP=1) and
    to the single-chunk run to the last bit or two, with work queued on the caller's stream before and after the call."""
    from qml_essentials_amd import _native as N
    from tests.test_abi_cpu import he_layer_ops

    ops, slots = [], 0
    for _ in range(2):
        o, s_ = he_layer_ops(n)
        ops += [(g, w, [x + slots for x in sl], m) for g, w, sl, m in o]
        slots += s_
    B = 20
    ang = torch.from_numpy(np.random.default_rng(5 * n).uniform(0, 2 * np.pi, (B, slots)).astype(np.float32)).cuda()
    plan = N.Plan(ops, n, slots, flags=flags)
    for meas, obs in (("expval", list(range(n))), ("state", ()), ("probs", ()), ("mw", ())):
        whole = plan.run(ang, meas, obs).clone()
        scratch = torch.zeros(1 << 20, device="cuda")
        scratch += 1.0                                   # (queued on the caller's stream before the call)
        piped = plan.run(ang, meas, obs, states_in_flight=3).clone()
        scratch *= 2.0
        monkeypatch.setenv("QMLE_NO_CHUNK_OVERLAP", "1")
        serial = plan.run(ang, meas, obs, states_in_flight=3).clone()
        monkeypatch.delenv("QMLE_NO_CHUNK_OVERLAP")
        # (equal up to the order in which a measurement's partial sums arrive -- the final reductions of the larger
        # registers add in arrival order, and a launch that shares the card is scheduled differently: 1 ulp)
        assert (piped - serial).abs().max().item() < 2e-7, (meas, (piped - serial).abs().max().item())
        assert (piped - whole).abs().max().item() < 1e-6, meas
        assert float(scratch[0]) == 2.0 and float(scratch[-1]) == 2.0
        assert (piped[0] - piped[1]).abs().max().item() > 1e-4   # rows are distinct parameter sets
