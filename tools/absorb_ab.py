#!/usr/bin/env python3
"""Observable folding as a choice: Hardware-Efficient circuits of 1-4 layers, <Z> on
all wires, microseconds per state with the engine's own choice between folding the trailing CX
layer into the observables and applying it, against QMLE_PLAN_NO_ABSORB (always applied).  Run
with QMLE_ALWAYS_FOLD=1 for the folded side of every row.

    python tools/absorb_ab.py
"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from qml_essentials_amd import _native as N
from dense_profile import he_ops


def timed(n, B, layers, flags):
    ops, slots = he_ops(n, layers)
    ang = torch.from_numpy(np.random.default_rng(1).uniform(0, 6.28, (B, slots)).astype(np.float32)).cuda()
    top = N.Plan(ops, n, slots, flags=flags)
    ws = torch.empty(top.workspace_bytes(B, "expval", n), dtype=torch.uint8, device="cuda")
    obs = list(range(n))
    out = top.run(ang, "expval", obs, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        out = top.run(ang, "expval", obs, workspace=ws)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3 / B * 1e3, out, "folded" if "expval_plan" in top.describe() else "applied"


if __name__ == "__main__":
    for n, layers in ((24, 1), (24, 2), (24, 3), (24, 4), (20, 4), (22, 3), (16, 2), (18, 3)):
        a, oa, choice = timed(n, 64, layers, 0)
        b, ob, _ = timed(n, 64, layers, N.PLAN_NO_ABSORB)
        print(f"n={n} layers={layers}: engine choice ({choice}) {a:.1f} us/state, NO_ABSORB {b:.1f} us/state, "
              f"max diff {float((oa - ob).abs().max()):.2e}", flush=True)
