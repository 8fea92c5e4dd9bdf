#!/usr/bin/env python3
"""How much of the 1e-6 budget do the REDUCTIONS use (as opposed to the float32 state)?  <Z> of every
wire and the purities of a few wires of a float32 n-qubit state from the library against float64 sums
over the SAME float32 amplitudes (torch, on the GPU)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import _native as N
from tests.test_abi_cpu import he_layer_ops

for n in (int(a) for a in (sys.argv[1:] or ["24", "28"])):
    ops, slots = he_layer_ops(n)
    ang = torch.from_numpy(np.random.default_rng(6).uniform(0, 6.28, (1, slots)).astype(np.float32)).cuda()
    plan = N.Plan(ops, n, slots)
    st = plan.run(ang, "state")
    p = (st[0].real.double() ** 2 + st[0].imag.double() ** 2)
    norm = float(p.sum())
    ez_ref = []
    for w in range(n):
        pos = n - 1 - w
        v = p.reshape(1 << (n - 1 - pos), 2, 1 << pos)
        ez_ref.append(float((v[:, 0, :].sum() - v[:, 1, :].sum())))
    del p
    ez_ref = np.array(ez_ref)
    ez = N.expval_z(st, list(range(n))).cpu().numpy()[0].astype(np.float64)
    ez_fused = plan.run(ang, "expval", list(range(n))).cpu().numpy()[0].astype(np.float64)
    print(f"n={n}: norm {norm:.9f}; <Z> stand-alone vs fp64 sums of the same state: max |d| {np.abs(ez - ez_ref).max():.2e}; "
          f"fused last pass (own state): {np.abs(ez_fused - ez_ref).max():.2e}")
    q, pur = N.meyer_wallach(st, return_purities=True)
    pur = pur.cpu().numpy()[0].astype(np.float64)
    worst = 0.0
    for w in (0, 1, n // 2, n - 2, n - 1):
        pos = n - 1 - w
        v = st[0].reshape(1 << (n - 1 - pos), 2, 1 << pos)
        a0, a1 = v[:, 0, :].to(torch.complex128), v[:, 1, :].to(torch.complex128)
        pa = float((a0.abs() ** 2).sum()); pd = float((a1.abs() ** 2).sum())
        c = complex((a0 * a1.conj()).sum())
        ref = pa * pa + pd * pd + 2 * abs(c) ** 2
        worst = max(worst, abs(pur[w] - ref))
        del a0, a1
    mwf = plan.run(ang, "mw").cpu().numpy()[0].astype(np.float64)
    print(f"n={n}: purity (wires 0, 1, n/2, n-2, n-1) vs fp64 sums of the same state: max |d| {worst:.2e}; "
          f"QMLE_MEAS_MEYER_WALLACH vs stand-alone: {np.abs(mwf[1:] - pur).max():.2e}")
    del st
    torch.cuda.empty_cache()
