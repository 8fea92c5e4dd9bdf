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
