"""World-size-2 gloo test of the batch-sharding path (CPU): shard bounds, the single
all-gather, and Script-level sharding with the oracle standing in for the engine."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_everything():
    from qml_essentials_amd.distributed import all_shard_bounds

    for n in (1, 2, 7, 8, 1024, 1025, 4096):
        for size in (1, 2, 3, 4, 8):
            b = all_shard_bounds(n, size)
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(size - 1))
            assert max(hi - lo for lo, hi in b) == -(-n // size)
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1
            if n >= size:  # no rank may be left without work: it would skip the engine call
                assert all(hi > lo for lo, hi in b), (n, size, b)


def test_no_empty_shards_for_any_split():
    """ceil(n/size) blocks left trailing ranks empty for (9, 8), (17, 8), (10, 8)."""
    from qml_essentials_amd.distributed import all_shard_bounds

    for size in range(1, 9):
        for n in range(size, 4 * size + 3):
            assert all(hi > lo for lo, hi in all_shard_bounds(n, size)), (n, size)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, size, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(size), LOCAL_RANK=str(rank))
    from oracle import einsum_sim as OE
    from qml_essentials_amd import distributed, memory, simulation
    from qml_essentials_amd import operations as op
    from qml_essentials_amd.script import Script

    distributed.init_from_env("gloo")
    assert distributed.world() == (rank, size) and distributed.enabled()
    # 1. uneven all-gather, real + complex
    n = 7
    lo, hi = distributed.shard_bounds(n)
    full = np.arange(n * 3, dtype=np.float32).reshape(n, 3)
    got = distributed.all_gather_rows(full[lo:hi], n)
    assert np.array_equal(got, full)
    cfull = (full[:, 0] + 1j * full[:, 1]).astype(np.complex64).reshape(n, 1)
    assert np.array_equal(distributed.all_gather_rows(cfull[lo:hi], n), cfull)
    with distributed.local_only():
        assert not distributed.enabled()

    # 2. Script-level sharding; the oracle stands in for the HIP engine on CPU
    calls = []

    offsets = []

    def fake_engine(tape, n_qubits, type, obs, use_density, shots=None, key=None, batch=None,
                    as_tensor=False, row_offset=0):
        calls.append(batch)
        offsets.append(row_offset)
        out = []
        for b in range(batch):
            t = [(o.name, o.wires, tuple(float(p[b]) if np.ndim(p) else float(p) for p in o.parameters))
                 for o in tape if o.name != "Barrier"]
            out.append(OE.simulate_and_measure(t, n_qubits, type, [("PauliZ", o.wires) for o in obs]))
        return np.stack(out)

    simulation.simulate_and_measure = fake_engine
    memory.available_memory_bytes = lambda: 1 << 40

    def circuit(theta, phi):
        op.RX(theta, wires=0)
        op.CRX(phi, wires=[0, 1])

    th = np.linspace(0, 3, 9).astype(np.float32)
    res = Script(circuit, n_qubits=2).execute(
        type="expval", obs=[op.PauliZ(0, record=False), op.PauliZ(1, record=False)],
        args=(th, np.float32(0.4)), in_axes=(0, None))
    want = np.stack([OE.simulate_and_measure([("RX", [0], (float(t),)), ("CRX", [0, 1], (0.4,))], 2,
                                             "expval", [("PauliZ", [0]), ("PauliZ", [1])]) for t in th])
    ok = res.shape == (9, 2) and np.allclose(res, want, atol=1e-6) and calls == [hi2 - lo2 for lo2, hi2 in [distributed.shard_bounds(9)]]
    # shot-sampling streams are keyed by the GLOBAL row: the shard passes its first row
    ok = ok and offsets == [distributed.shard_bounds(9)[0]]
    q.put((rank, bool(ok), calls))
    distributed.barrier()
    torch.distributed.destroy_process_group()


def test_world_size_2_gloo_sharded_execution():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, calls in results:
        assert ok, (rank, calls)
    assert sorted(c[0] for _, _, c in results) == [4, 5]  # 9 samples -> 5 + 4


# ---------------------------------------------------------------------------------------------
# the three sampling loops under world size 2 and a world-size-4 split
# ---------------------------------------------------------------------------------------------
def _install_fake_engine(calls):
    """The oracle stands in for libqmle_sv on CPU ranks: statevectors / expvals per batch row,
    pair fidelities, Meyer-Wallach and the histogram from ``oracle/analysis.py``."""
    from oracle import analysis as OA, einsum_sim as OE
    from qml_essentials_amd import _native as N, memory, simulation

    def fake_engine(tape, n_qubits, type, obs, use_density, shots=None, key=None, batch=None,
                    as_tensor=False, row_offset=0):
        calls.append((type, batch, row_offset))
        out = []
        for b in range(batch):
            t = [(o.name, o.wires, tuple(float(np.asarray(p).reshape(-1)[b if np.size(p) > 1 else 0])
                                         for p in o.parameters))
                 for o in tape if o.name != "Barrier"]
            out.append(OE.simulate_and_measure(t, n_qubits, type, [("PauliZ", o.wires) for o in obs]))
        res = np.stack(out)
        res = res.astype(np.complex64 if np.iscomplexobj(res) else np.float32)
        return torch.from_numpy(res) if as_tensor else res

    simulation.simulate_and_measure = fake_engine
    memory.available_memory_bytes = lambda: 1 << 40
    from qml_essentials_amd.model import Model
    Model.host_arrays_via_device = False  # no GPU here: every call takes the Script path (same my_block() split)
    N.require_gpu = lambda: torch
    N.pair_fidelity = lambda st: torch.from_numpy(
        OA.fidelities_pure(st.numpy(), st.shape[0] // 2).astype(np.float32))
    N.meyer_wallach = lambda st: torch.from_numpy(
        np.array([OA.meyer_wallach_pure(v, int(np.log2(v.shape[0]))) for v in st.numpy()],
                 dtype=np.float32))
    N.histogram = lambda x, n_bins, lo, hi: torch.from_numpy(
        np.histogram(x.numpy(), bins=np.linspace(lo, hi, n_bins + 1))[0].astype(np.int32))


def _loops_worker(rank, size, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(size), LOCAL_RANK=str(rank))
    from qml_essentials_amd import distributed
    from qml_essentials_amd.coefficients import Coefficients
    from qml_essentials_amd.entanglement import Entanglement
    from qml_essentials_amd.expressibility import Expressibility
    from qml_essentials_amd.model import Model

    distributed.init_from_env("gloo")
    calls = []
    _install_fake_engine(calls)
    report = {}

    # Expressibility: pairs (i, i + S) stay on one rank; the S fidelities come back in global order
    S = 7
    m = Model(3, 1, "Hardware_Efficient", data_reupload=False)
    fid = Expressibility._sample_state_fidelities(m, S, random_key=5).numpy()
    lo, hi = distributed.shard_bounds(S)
    report["expr_calls"] = list(calls)
    report["expr_shard"] = (lo, hi)
    report["expr_fid"] = fid
    params_all = np.asarray(m.params)                     # (2S, ...): restored after the call
    report["expr_params_shape"] = params_all.shape
    with distributed.local_only():                          # unsharded reference on this rank
        calls.clear()
        want = Expressibility._sample_state_fidelities(
            Model(3, 1, "Hardware_Efficient", data_reupload=False), S, random_key=5).numpy()
    report["expr_want"] = want

    # Meyer-Wallach: samples sharded, mean over all of them
    calls.clear()
    m2 = Model(3, 1, "Hardware_Efficient", data_reupload=False)
    mw = Entanglement.meyer_wallach(m2, n_samples=5, random_key=9)
    report["mw_calls"] = list(calls)
    with distributed.local_only():
        mw_want = Entanglement.meyer_wallach(Model(3, 1, "Hardware_Efficient", data_reupload=False),
                                             n_samples=5, random_key=9)
    report["mw"] = (mw, mw_want)

    # Coefficients: the input grid is the batch axis
    calls.clear()
    m3 = Model(2, 1, "Hardware_Efficient")
    m3.host_arrays_via_device = False  # no GPU here: the Script path (same my_block() split)
    coeffs, freqs = Coefficients.get_spectrum(m3, mfs=2, shift=True)
    report["coef_calls"] = list(calls)
    with distributed.local_only():
        c_want, f_want = Coefficients.get_spectrum(m3, mfs=2, shift=True)
    report["coef"] = (coeffs, c_want, np.asarray(freqs), np.asarray(f_want))
    q.put((rank, report))
    distributed.barrier()
    torch.distributed.destroy_process_group()


def _spawn(target, size, timeout=240):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, size, port, q)) for r in range(size)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=timeout) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return results


def test_world_size_2_gloo_sampling_loops():
    from qml_essentials_amd.distributed import all_shard_bounds

    res = _spawn(_loops_worker, 2)
    S = 7
    bounds = all_shard_bounds(S, 2)
    for rank, rep in res.items():
        lo, hi = bounds[rank]
        # ONE engine call per rank, for its own pairs: 2 * (hi - lo) states (i and i + S local)
        assert rep["expr_calls"] == [("state", 2 * (hi - lo), 0)], rep["expr_calls"]
        assert rep["expr_shard"] == (lo, hi)
        assert rep["expr_params_shape"][0] == 2 * S
        # gathered fidelities: all S of them, global order, equal to the unsharded run
        assert rep["expr_fid"].shape == (S,)
        np.testing.assert_allclose(rep["expr_fid"], rep["expr_want"], atol=1e-6)
        mlo, mhi = all_shard_bounds(5, 2)[rank]
        assert rep["mw_calls"] == [("state", mhi - mlo, 0)], rep["mw_calls"]
        assert abs(rep["mw"][0] - rep["mw"][1]) < 1e-6
        c, cw, f, fw = rep["coef"]
        n_grid = c.shape[0]
        glo, ghi = all_shard_bounds(n_grid, 2)[rank]
        assert [(t, b) for t, b, _ in rep["coef_calls"]] == [("expval", ghi - glo)], rep["coef_calls"]
        np.testing.assert_allclose(c, cw, atol=1e-6)
        np.testing.assert_array_equal(f, fw)
    # both ranks hold the same gathered result
    np.testing.assert_array_equal(res[0]["expr_fid"], res[1]["expr_fid"])


def _uneven_worker(rank, size, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(size), LOCAL_RANK=str(rank))
    from qml_essentials_amd import distributed
    from qml_essentials_amd.model import Model

    distributed.init_from_env("gloo")
    calls = []
    _install_fake_engine(calls)
    m = Model(2, 1, "Hardware_Efficient")
    m.host_arrays_via_device = False
    x = np.linspace(0, 1, 5, dtype=np.float32).reshape(5, 1)   # n = 5 on 4 ranks: 1 + 1 + 1 + 2
    out = np.asarray(m(inputs=x))
    with distributed.local_only():
        want = np.asarray(m(inputs=x))
    # opt-out: a DDP-style caller with its own minibatch per rank gets exactly its own rows
    distributed.enable(False)
    mine = np.asarray(m(inputs=x[rank:rank + 2]))
    distributed.enable(True)
    q.put((rank, dict(calls=calls[:1], out=out, want=want, mine=mine, mine_want=want[rank:rank + 2])))
    distributed.barrier()
    torch.distributed.destroy_process_group()


def test_world_size_4_gloo_uneven_split_and_opt_out():
    res = _spawn(_uneven_worker, 4)
    sizes = sorted(r["calls"][0][1] for r in res.values())
    assert sizes == [1, 1, 1, 2]
    for rank, r in res.items():
        assert r["out"].shape == r["want"].shape
        np.testing.assert_allclose(r["out"], r["want"], atol=1e-6)
        np.testing.assert_allclose(r["mine"], r["mine_want"], atol=1e-6)


def test_process_group_alone_does_not_switch_sharding_on():
    """sharding is opt-in (init_from_env / enable / QMLE_SHARD), so a foreign
    torch.distributed process group (DDP with per-rank minibatches) leaves calls local."""
    import qml_essentials_amd.distributed as d

    saved = d._opt_in
    try:
        d._opt_in = False
        d.world = lambda: (1, 4)
        assert not d.enabled() and d.my_block(10) == (0, 10, False)
        d.enable()
        assert d.enabled() and d.my_block(10) == (2, 5, True)
        assert d.my_block(3) == (0, 3, False)  # fewer rows than ranks: everybody computes all
        with d.local_only():
            assert not d.enabled() and d.my_block(10) == (0, 10, False)
    finally:
        import importlib
        importlib.reload(d)
        d._opt_in = saved


# ---------------------------------------------------------------------------------------------
# QMLE_SHARD_CHECK=1: the identical-arguments contract is verified at every my_block() call site
# (check_same_arguments existed but nothing called it)
# ---------------------------------------------------------------------------------------------
def _check_worker(rank, size, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(size), LOCAL_RANK=str(rank), QMLE_SHARD_CHECK="1")
    from qml_essentials_amd import distributed
    from qml_essentials_amd.entanglement import Entanglement
    from qml_essentials_amd.model import Model

    distributed.init_from_env("gloo")
    calls = []
    _install_fake_engine(calls)
    m = Model(2, 1, "Hardware_Efficient")
    m.host_arrays_via_device = False
    x = np.linspace(0, 1, 6, dtype=np.float32).reshape(6, 1)
    same = np.asarray(m(inputs=x))                      # identical arguments: passes
    report = {"same_shape": same.shape}
    try:                                                # per-rank minibatch with sharding on: refused
        m(inputs=x + np.float32(rank))
        report["script"] = "no error"
    except RuntimeError as e:
        report["script"] = str(e)
    m2 = Model(3, 1, "Hardware_Efficient", data_reupload=False)
    try:                                                # different sampling key -> different params
        Entanglement.meyer_wallach(m2, n_samples=4, random_key=100 + rank)
        report["mw"] = "no error"
    except RuntimeError as e:
        report["mw"] = str(e)
    q.put((rank, report))
    distributed.barrier()
    torch.distributed.destroy_process_group()


def test_shard_check_refuses_different_arguments_across_ranks():
    res = _spawn(_check_worker, 2)
    for rank, rep in res.items():
        assert rep["same_shape"] == (6, 2)
        assert "different arguments" in rep["script"], rep
        assert "different arguments" in rep["mw"], rep
