#!/usr/bin/env python3
"""HE circuits of 1..4 layers at n = 24 with and without known-zero tracking: us per state and
per-pass times (profiles/r01_layers_sweep.md).  Run on the GPU box."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import _native as N
from tests.test_abi_cpu import he_layer_ops


def run(n, B, flags, label, layers, reps=3):
    ops, slots = he_layer_ops(n)
    allops = []
    for l in range(layers):
        allops += [(g, w, [s + l * slots for s in sl], off) for g, w, sl, off in ops]
    rng = np.random.default_rng(1000)
    ang = torch.from_numpy(rng.uniform(0, 2 * np.pi, (B, layers * slots)).astype(np.float32)).cuda()
    top = N.Plan(allops, n, layers * slots, flags=flags)
    plan = top.executed("expval")
    d = plan.describe()
    ws = torch.empty(top.workspace_bytes(B, "expval", n), dtype=torch.uint8, device="cuda")
    obs = list(range(n))
    top.run(ang, "expval", obs, workspace=ws)
    torch.cuda.synchronize()
    plan.profile_begin(len(d["stages"]) * reps * 4 + 8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        top.run(ang, "expval", obs, workspace=ws)
    e1.record()
    torch.cuda.synchronize()
    ms, cnt, _ = plan.profile_end()
    print(f"{label}: {e0.elapsed_time(e1) / reps / B * 1e3:.1f} us/state",
          [(s["T"], s["L"], s["n_lowered"], s["lds_round_trips"], round(m / max(c, 1) * 1e3))
           for s, m, c in zip(d["stages"], ms, cnt)], flush=True)


if __name__ == "__main__":
    n, B = int(os.environ.get("SWEEP_N", "24")), int(os.environ.get("SWEEP_B", "32"))
    for layers in (1, 2, 3, 4):
        run(n, B, N.plan_flags(no_sparse=True), f"{layers} layers dense", layers)
        run(n, B, 0, f"{layers} layers tracked", layers)
