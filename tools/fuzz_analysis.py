#!/usr/bin/env python3
"""Differential check of the device-resident analysis loops (round 4) against plain numpy on the SAME
sampled parameter sets: Expressibility's pair fidelities (sampler on the GPU or the host, pairs sliced on
the device, k_pair_fidelity_small) and Entanglement.meyer_wallach (QMLE_MEAS_MEYER_WALLACH out of the
producing pass) -- every ansatz, 2..12 qubits, sample counts on both sides of the 64-sample and the
16384-value (device sampler) thresholds.  The states for the numpy side come through the host-array route."""
import os, sys, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd.ansaetze import Ansaetze
from qml_essentials_amd.entanglement import Entanglement
from qml_essentials_amd.expressibility import Expressibility
from qml_essentials_amd.model import Model

warnings.simplefilter("ignore")
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "11")))
names = [a.__name__ for a in Ansaetze.get_available()]
bad = ran = 0


def purities(psi, n):
    out = []
    for w in range(n):
        m = np.moveaxis(psi.reshape((2,) * n), w, 0).reshape(2, -1)
        rho = m @ m.conj().T
        out.append(float(np.real(np.trace(rho @ rho))))
    return np.array(out)


for trial in range(int(os.environ.get("FUZZ_N", "60"))):
    n = int(rng.integers(2, 13))
    kw = dict(n_qubits=n, n_layers=int(rng.integers(1, 4)), circuit_type=str(rng.choice(names)),
              data_reupload=False)
    try:
        m = Model(**kw)
    except Exception:
        continue
    if 0 in m.params.shape:  # nothing to sample (the reference's loops have no meaning there either)
        continue
    S = int(rng.choice([3, 31, 32, 33, 64, 100, 400])) if n <= 10 else int(rng.choice([3, 33, 100]))
    key = int(rng.integers(1, 10**6))
    # --- Expressibility: fidelity of sample i with sample i + S
    fid = Expressibility._sample_state_fidelities(m, S, random_key=key)
    fid = (fid.cpu().numpy() if hasattr(fid, "is_cuda") else np.asarray(fid)).reshape(-1)
    P = np.asarray(m.params, dtype=np.float32)
    if P.shape[0] != 2 * S and 0 not in P.shape:
        print("SHAPE", kw, S, P.shape); bad += 1; continue
    st = np.asarray(m(params=P, execution_type="state")).reshape(-1, 1 << n) if 0 not in P.shape else None
    if st is not None:
        want = np.abs(np.einsum("bi,bi->b", st[:S].conj(), st[S:])) ** 2
        err = float(np.abs(fid - want).max())
        if not (fid.shape == want.shape and err < 2e-6):
            print("FIDELITY MISMATCH", kw, S, key, fid.shape, err); bad += 1
    # --- Meyer-Wallach: mean Q over S sampled sets
    q = Entanglement.meyer_wallach(m, S, random_key=key + 1)
    P = np.asarray(m.params, dtype=np.float32)
    if 0 not in P.shape and P.shape[0] == S:
        st = np.asarray(m(params=P, execution_type="state")).reshape(-1, 1 << n)
        qs = [2.0 * (1.0 - purities(psi.astype(np.complex128), n).mean()) for psi in st[: min(S, 40)]]
        if S <= 40:
            err = abs(q - float(np.mean(qs)))
            if not err < 2e-6:
                print("MEYER-WALLACH MISMATCH", kw, S, key, q, float(np.mean(qs))); bad += 1
        # the stand-alone kernels on the stored states, all S of them
        from qml_essentials_amd import _native as N
        q2 = float(N.meyer_wallach(torch.from_numpy(st.astype(np.complex64)).cuda()).double().mean())
        if not abs(q - q2) < 2e-6:
            print("MEYER-WALLACH vs stand-alone", kw, S, key, q, q2); bad += 1
    ran += 1
print(f"{ran} models run, mismatches: {bad}")
