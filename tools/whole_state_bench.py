#!/usr/bin/env python3
"""Whole-state regime (n <= 14: one LDS tile per sample): states per second of the circuit +
<Z> launch, HIP-event timed.  QMLE_NO_FAST_WHOLE=1 selects the generic k_tile for A/B."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import _native as N
from tests.test_abi_cpu import he_layer_ops


def he_ops(n, layers):
    ops, slots = [], 0
    for _ in range(layers):
        o, s = he_layer_ops(n)
        ops += [(g, w, [x + slots for x in sl], m) for g, w, sl, m in o]
        slots += s
    return ops, slots


SHAPES = ((10, 6, 65536), (12, 3, 32768), (13, 3, 16384), (10, 6, 4096), (12, 3, 2048))
if os.environ.get("WS_SHAPES"):  # e.g. WS_SHAPES=10:6:65536,12:3:32768
    SHAPES = tuple(tuple(int(v) for v in t.split(":")) for t in os.environ["WS_SHAPES"].split(","))
REPS = int(os.environ.get("WS_REPS", "20"))
for n, layers, B in SHAPES:
    ops, slots = he_ops(n, layers)
    ang = torch.from_numpy(np.random.default_rng(0).uniform(0, 6.28, (B, slots)).astype(np.float32)).cuda()
    plan = N.Plan(ops, n, slots)
    ws = torch.empty(plan.workspace_bytes(B, "expval", n), dtype=torch.uint8, device="cuda")
    obs = list(range(n))
    for _ in range(3):
        plan.run(ang, "expval", obs, workspace=ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = REPS
    e0.record()
    for _ in range(reps):
        plan.run(ang, "expval", obs, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    d = plan.executed("expval").describe()["stages"][0]
    print(f"n={n} layers={layers} B={B}: {ms*1e3:.1f} us per launch = {B/ms/1e3:.2f} M states/s, "
          f"{len(ops)} gates, fast={d['fast']} groups={len(d['fast_groups']) if d['fast'] else len(d['groups'])} "
          f"(generic grouping: {len(d['groups'])})", flush=True)
