"""GPU: RCCL itself.  The gloo tests cover the sharding logic with the
oracle standing in for the engine; here a CUDA tensor goes through ``all_gather_into_tensor`` on
the ``nccl`` (= RCCL) backend -- a one-rank group, which is what a 1-GPU box can host -- and the
real engine runs under an initialised process group.  Reference seam: script.py:443-453."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture()
def nccl_group():
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        yield dist
    finally:
        dist.destroy_process_group()


def test_rccl_all_gather_into_tensor_on_cuda_tensors(nccl_group):
    import torch

    dist = nccl_group
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    x = torch.arange(24 * 1024, dtype=torch.float32, device="cuda").reshape(1024, 24)
    out = torch.empty_like(x)
    dist.all_gather_into_tensor(out, x)   # the ONE collective of the data path
    torch.cuda.synchronize()
    assert torch.equal(out, x)
    t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)   # bench.py's max-over-ranks
    assert float(t.item()) == 1.5


def test_all_gather_rows_device_path_under_nccl(nccl_group):
    """distributed.all_gather_rows with CUDA tensors / host arrays / complex rows on RCCL."""
    import torch

    from qml_essentials_amd import distributed

    assert distributed.is_initialized() and distributed.world() == (0, 1)
    rows = torch.randn(7, 5, device="cuda")
    got = distributed.all_gather_rows(rows, 7)
    assert got.is_cuda and torch.equal(got, rows)
    c = torch.view_as_complex(torch.randn(6, 4, 2, device="cuda"))
    assert torch.equal(distributed.all_gather_rows(c, 6), c)
    h = np.arange(12, dtype=np.float32).reshape(4, 3)   # host rows travel through the GPU under nccl
    np.testing.assert_array_equal(distributed.all_gather_rows(h, 4), h)
    with pytest.raises(ValueError):
        distributed.all_gather_rows(rows[:3], 7)        # a shard of the wrong size is refused


def test_engine_results_gathered_by_rccl_equal_the_oracle(nccl_group):
    """The compiled device path of Model under an initialised RCCL group: rows computed by
    libqmle_sv, gathered on the GPU, equal to the oracle."""
    import torch

    from oracle import circuits as OC, einsum_sim as OE
    from qml_essentials_amd import distributed
    from qml_essentials_amd.model import Model

    rng = np.random.default_rng(3)
    m = Model(5, 1, "Hardware_Efficient")
    P = rng.uniform(0, 2 * np.pi, (6, *m.params.shape[1:])).astype(np.float32)
    x = np.array([[0.3]], dtype=np.float32)
    out = m(params=torch.from_numpy(P).cuda(), inputs=torch.from_numpy(x).cuda())
    gathered = distributed.all_gather_rows(out, 6)       # what a sharded call does with its block
    spec = OC.ModelSpec(5, 1, "Hardware_Efficient")
    want = np.stack([OE.simulate_and_measure(OC.model_tape(spec, P[b], [0.3]), 5, "expval",
                                             [("PauliZ", [q]) for q in range(5)], np.complex128)
                     for b in range(6)])
    assert gathered.is_cuda
    np.testing.assert_allclose(gathered.cpu().numpy(), want, atol=1e-6)
