#!/usr/bin/env python3
"""qmle_plan_autotune on the shapes whose schedules the round-3 searches found to be off the best:
per shape the per-state time of the cost model's schedule, the tuner's report, the per-state time
afterwards and the largest |difference| of the <Z> values between the two schedules."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import _native as N, simulation
from qml_essentials_amd.model import Model

DENSE = N.PLAN_NO_SPARSE | N.PLAN_NO_ABSORB
SHAPES = [  # (label, n, layers, dru, flags, batch)
    ("k2 all-live", 24, 1, False, DENSE, 64),
    ("k2_deep all-live", 24, 4, True, DENSE, 32),
    ("k2_deep default", 24, 4, True, 0, 64),
    ("n=22 3 layers all-live", 22, 3, False, DENSE, 128),
    ("n=26 HE layer all-live", 26, 1, False, DENSE, 16),
    ("n=26 2 layers default", 26, 2, False, 0, 16),
    ("n=28 HE layer all-live", 28, 1, False, DENSE, 4),
]
if os.environ.get("AT_SHAPES"):
    SHAPES = [s for i, s in enumerate(SHAPES) if str(i) in os.environ["AT_SHAPES"].split(",")]
TOP_K = int(os.environ.get("AT_TOP_K", "6"))


def per_state_us(plan, ang, obs, ws, reps=5):
    for _ in range(2):
        out = plan.run(ang, "expval", obs, workspace=ws)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = plan.run(ang, "expval", obs, workspace=ws)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / ang.shape[0] * 1e3, out


for label, n, layers, dru, flags, B in SHAPES:
    saved, simulation.PLAN_FLAGS = simulation.PLAN_FLAGS, flags
    try:
        m = Model(n, layers, "Hardware_Efficient", data_reupload=dru)
        rng = np.random.default_rng(1000)
        params = rng.uniform(0, 2 * np.pi, (2, *m.params.shape[1:])).astype(np.float32)
        x = np.full((1, 1), 0.5, dtype=np.float32) if dru else None
        tape, _ = m.record_tape(params=params, inputs=x)
        low = simulation.LoweredTape(tape, n)
        plan = N.Plan(low.ops, n, low.n_slots, consts=low.consts if len(low.consts) else None, flags=flags)
    finally:
        simulation.PLAN_FLAGS = saved
    ang = torch.from_numpy(rng.uniform(0, 2 * np.pi, (B, low.n_slots)).astype(np.float32)).cuda()
    obs = list(range(n))
    ws = torch.empty(plan.workspace_bytes(B, "expval", n), dtype=torch.uint8, device="cuda")
    before, z0 = per_state_us(plan, ang, obs, ws)
    t0 = time.perf_counter()
    rep = plan.autotune("expval", n, batch=B, top_k=TOP_K, reps=3)
    tune_s = time.perf_counter() - t0
    ws = torch.empty(plan.workspace_bytes(B, "expval", n), dtype=torch.uint8, device="cuda")
    after, z1 = per_state_us(plan, ang, obs, ws)
    d = plan.executed("expval").describe()
    print(f"{label}: {before:.1f} -> {after:.1f} us per state ({(after / before - 1) * 100:+.1f} %), tuner {rep}, "
          f"{tune_s:.2f} s, max |d<Z>| {float((z0 - z1).abs().max()):.1e}, stages now "
          f"{[(s['T'], s['bits'][:2] + ['..'] + s['bits'][-2:]) for s in d['stages']]}", flush=True)
    del ws, ang
    torch.cuda.empty_cache()
