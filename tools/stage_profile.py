#!/usr/bin/env python3
"""Per-pass HIP-event timing of the K2 plan (one state at a time)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import _native as N
from tests.test_abi_cpu import he_layer_ops

def run(n, B, flags, label, reps=3):
    ops, slots = he_layer_ops(n)
    rng = np.random.default_rng(1000)
    ang = torch.from_numpy(rng.uniform(0, 2 * np.pi, (B, slots)).astype(np.float32)).cuda()
    top = N.Plan(ops, n, slots, flags=flags)
    plan = top.executed("expval")     # what "expval" executes (trailing CX folded away)
    d = plan.describe()
    ws = torch.empty(top.workspace_bytes(B, "expval", n), dtype=torch.uint8, device="cuda")
    obs = list(range(n))
    top.run(ang, "expval", obs, workspace=ws)
    torch.cuda.synchronize()
    plan.profile_begin(len(d["stages"]) * B * reps + 8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        top.run(ang, "expval", obs, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    ms, cnt, _ = plan.profile_end()
    tot = e0.elapsed_time(e1) / reps / B
    print(f"{label}: n={n} B={B} total {tot*1e3:.1f} us/state; passes:",
          [(s["kind"], s["n_lowered"], s.get("lds_round_trips"), round(m / max(c, 1) * 1e3, 1))
           for s, m, c in zip(d["stages"], ms, cnt)], flush=True)

if __name__ == "__main__":
    F = N.plan_flags
    run(24, 32, 0, "auto n24")
    run(24, 32, F(no_absorb=True), "auto n24, no folding")
    for T, L in ((12, 4), (13, 4), (13, 5), (14, 4), (14, 5), (14, 6)):
        run(24, 32, F(tile_bits=T, low_bits=L), f"T{T} L{L}")
