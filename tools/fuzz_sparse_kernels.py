#!/usr/bin/env python3
"""Plan-level fuzz of known-zero tracking: random shallow tapes on random wire subsets, random
tile geometries; state / probs / <Z> / parities with and without QMLE_PLAN_NO_SPARSE."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qml_essentials_amd import _native as N
from tests.helpers import random_tape, tape_to_native

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
bad = 0
torch.full((1 << 22,), 5.0, dtype=torch.complex64, device="cuda")  # junk for recycled buffers
for trial in range(int(os.environ.get("FUZZ_N", "200"))):
    n = int(rng.integers(15, 21))
    k = int(rng.integers(2, n + 1))
    wires = sorted(int(w) for w in rng.choice(n, size=k, replace=False))
    sub = random_tape(k, int(rng.integers(3, 45)), rng, three_q=bool(rng.integers(2))) if k >= 3 else \
        [("RX", [0], (1.0,)), ("RY", [k - 1], (0.5,))]
    tape = [(nm, [wires[w] for w in ws], pr) for nm, ws, pr in sub]
    layered = trial % 2 == 1
    if layered:  # one layer of 1-qubit rotations (+ a foldable entangling tail): product / mono passes
        tape = []
        for w in wires:
            for g_ in rng.choice(["RX", "RY", "RZ", "H"], size=int(rng.integers(1, 4))):
                tape.append((str(g_), [w], (float(rng.uniform(0, 6.28)),) if g_ != "H" else ()))
        for _ in range(int(rng.integers(0, 6))):
            a_, b_ = (int(x) for x in rng.choice(n, size=2, replace=False))
            tape.append((str(rng.choice(["CX", "CZ"])), [a_, b_], ()))
    ops, angles, consts = tape_to_native(tape, n)
    T = int(rng.integers(10, 14)) if layered else int(rng.integers(6, 14)); T = min(T, n - 1)
    L = int(rng.integers(1, min(T, 8)))
    if rng.random() < 0.25:  # the plan compiler's own choice of tile geometry
        T = L = 0
    B = int(rng.integers(1, 4))
    if layered and rng.random() < 0.3:  # enough states per launch for the streaming product kernel
        B = int(rng.integers(24, 72))
    table = rng.uniform(0, 6.28, size=(B, max(1, len(angles)))).astype(np.float32)[:, : len(angles)]
    ang = torch.from_numpy(np.ascontiguousarray(table)).cuda()
    if ang.shape[1] == 0:
        ang = torch.zeros((B, 1), dtype=torch.float32, device="cuda")[:, :0]
    kw = dict(force_global=True, tile_bits=T, low_bits=L, force_tile=bool(rng.integers(2)),
              no_absorb=bool(rng.integers(2)), tape_order=bool(rng.integers(2)))
    masks = [[0], [n - 1], sorted(int(w) for w in rng.choice(n, size=3, replace=False))]
    res = {}
    for mode in ("sparse", "dense"):
        plan = N.Plan(ops, n, len(angles), consts, N.plan_flags(no_sparse=(mode == "dense"), **kw))
        res[mode] = [plan.run(ang, "state").cpu().numpy(), plan.run(ang, "probs").cpu().numpy(),
                     plan.run(ang, "expval", list(range(n))).cpu().numpy(),
                     plan.run_parity(ang, masks).cpu().numpy()]
    errs = [float(np.abs(a - b).max()) for a, b in zip(res["sparse"], res["dense"])]
    ok = max(errs) < 2e-6
    bad += not ok
    print(trial, "layered" if layered else "random", n, k, len(tape), T, L, kw["force_tile"], kw["no_absorb"], kw["tape_order"], errs,
          "" if ok else "<<< MISMATCH", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
