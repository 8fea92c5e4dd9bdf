"""One rank of tests/test_gpu_multirank.py: a fresh process (started by the test with subprocess, never a
re-exec of a process that has touched the GPU) that joins a gloo group of WORLD_SIZE ranks sharing
cuda:0 and runs the sharded call sites with the REAL engine.  Usage: gpu_rank_worker.py <out.npz>
(RANK / WORLD_SIZE / MASTER_* from the environment, QMLE_DIST_BACKEND=gloo)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def workload(counter=None):
    """The calls both the ranks and the single-process reference make; `counter()` -> collectives so far."""
    import torch

    from qml_essentials_amd import operations as op
    from qml_essentials_amd.coefficients import Coefficients
    from qml_essentials_amd.entanglement import Entanglement
    from qml_essentials_amd.expressibility import Expressibility
    from qml_essentials_amd.model import Model
    from qml_essentials_amd.script import Script

    count = counter or (lambda: 0)
    out, per_call = {}, {}

    def track(name, fn):
        c0 = count()
        out[name] = np.asarray(fn())
        per_call[name] = count() - c0

    # 1. Script.execute over a batch of 9 angle pairs (uneven split 4 + 5)
    def circuit(theta, phi):
        op.RX(theta, wires=0)
        op.RY(phi, wires=1)
        op.CRX(phi, wires=[0, 1])
        op.CX(wires=[1, 2])

    th = np.linspace(0.1, 3.0, 9).astype(np.float32)
    ph = np.linspace(2.0, 0.2, 9).astype(np.float32)
    obs = [op.PauliZ(q, record=False) for q in range(3)]
    track("script", lambda: Script(circuit, n_qubits=3).execute(type="expval", obs=obs, args=(th, ph), in_axes=(0, 0)))
    # 2. Model: CUDA inputs (the compiled device path), 37 grid points x 1 parameter set, 6 qubits
    m = Model(6, 2, "Hardware_Efficient")
    x = torch.linspace(0, 6.0, 37, device="cuda").reshape(-1, 1)
    track("model_device", lambda: m(inputs=x).cpu().numpy())
    track("model_host", lambda: m(inputs=np.linspace(0, 6.0, 11, dtype=np.float32)))
    # 3. Expressibility: pairs (i, i + S) stay on one rank (device-resident parameters: 12 qubits, 700 pairs)
    me = Model(12, 2, "Hardware_Efficient", data_reupload=False)
    track("fidelities", lambda: Expressibility._sample_state_fidelities(me, 700, random_key=11).cpu().numpy())
    track("kl", lambda: Expressibility.kl_divergence_to_haar(me, n_samples=700, n_bins=40, random_key=11))
    # (small draw: host-resident parameters, 5 qubits, 33 pairs)
    ms = Model(5, 1, "Circuit_19", data_reupload=False)
    track("fidelities_small", lambda: Expressibility._sample_state_fidelities(ms, 33, random_key=5).cpu().numpy())
    # 4. Meyer-Wallach: samples sharded (11 qubits: the fused measurement; 16 qubits: tiled states)
    mw = Model(11, 2, "Hardware_Efficient", data_reupload=False)
    track("mw", lambda: Entanglement.meyer_wallach(mw, n_samples=1500, random_key=3))
    mt = Model(16, 1, "Hardware_Efficient", data_reupload=False)
    track("mw_tiled", lambda: Entanglement.meyer_wallach(mt, n_samples=5, random_key=4))
    # 5. Fourier spectrum: the input grid is the batch axis
    mc = Model(4, 2, "Circuit_19")
    track("spectrum", lambda: Coefficients.get_spectrum(mc, mfs=3, shift=True)[0])
    return out, per_call


def main():
    import torch
    import torch.distributed as dist

    from qml_essentials_amd import distributed

    torch.cuda.set_device(0)
    rank, size = distributed.init_from_env("gloo")
    assert distributed.enabled() and size == int(os.environ["WORLD_SIZE"])
    calls = {"n": 0}
    real = dist.all_gather_into_tensor

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    dist.all_gather_into_tensor = counting
    from qml_essentials_amd import _native as N

    out, per_call = workload(lambda: calls["n"])
    loaded = [ln.split()[-1] for ln in open("/proc/self/maps") if "libqmle_sv" in ln]
    np.savez(sys.argv[1], rank=rank, size=size, lib=np.array(sorted(set(loaded))), version=N.lib().qmle_sv_version(),
             **{"out_" + k: v for k, v in out.items()}, **{"cnt_" + k: v for k, v in per_call.items()})
    distributed.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
