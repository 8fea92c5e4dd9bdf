"""The real engine under N > 1 ranks: two FRESH processes, one gloo group, both on
cuda:0 (what a 1-GPU box can host -- RCCL needs a GPU per rank), running the sharded call sites --
Script, Model (device and host arguments), Expressibility pairs, Meyer-Wallach samples, the Fourier
grid -- on libqmle_sv itself, no oracle stand-in.  The gathered rows must equal the single-process rows
BIT FOR BIT (a row's value does not depend on which rank computed it, or in which batch), with ONE
collective per call.  The authors' replacement point: qml_essentials/script.py:443-453."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "gpu_rank_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("size", [2, 3])
def test_sharded_call_sites_on_the_real_engine(size, tmp_path):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    port = _free_port()
    procs = []
    for rank in range(size):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(size), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), QMLE_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("QMLE_SHARD", None)
        procs.append(subprocess.Popen([sys.executable, WORKER, str(tmp_path / f"rank{rank}.npz")], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(out)
    for rank, p in enumerate(procs):
        assert p.returncode == 0, f"rank {rank}:\n{logs[rank][-3000:]}"
    # the single-process reference, in THIS process (no process group: nothing is sharded)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_rank_worker import workload

    want, _ = workload()
    for rank in range(size):
        got = np.load(tmp_path / f"rank{rank}.npz", allow_pickle=False)
        assert int(got["rank"]) == rank and int(got["size"]) == size
        assert any("libqmle_sv.so" in str(s) for s in got["lib"]), got["lib"]   # the HIP engine was mapped in the rank
        for name, ref in want.items():
            val = got["out_" + name]
            assert val.shape == ref.shape, (rank, name, val.shape, ref.shape)
            assert np.array_equal(val, ref), (rank, name, float(np.abs(val - ref).max()))
            # one collective per call (kl = one call of state_fidelities; host-argument calls included)
            assert int(got["cnt_" + name]) == 1, (rank, name, int(got["cnt_" + name]))
