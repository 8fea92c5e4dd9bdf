#!/usr/bin/env python3
"""Every schedule candidate of the plan compiler (QMLE_FORCE_CAND = 0..47) on one all-live HE circuit:
model cost vs HIP-event time per state, to check the pass-cost model away from the sizes it was fitted
at.  CAND_N / CAND_LAYERS / CAND_B choose the circuit."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import _native as N
from tests.test_abi_cpu import he_layer_ops

n = int(os.environ.get("CAND_N", "28"))
layers = int(os.environ.get("CAND_LAYERS", "1"))
B = int(os.environ.get("CAND_B", str(max(1, 64 >> max(0, n - 24)))))
ops, slots = [], 0
for _ in range(layers):
    o, s_ = he_layer_ops(n)
    ops += [(g, w, [x + slots for x in sl], m) for g, w, sl, m in o]
    slots += s_
ang = torch.from_numpy(np.random.default_rng(1000).uniform(0, 2 * np.pi, (B, slots)).astype(np.float32)).cuda()
flags = N.PLAN_NO_SPARSE | N.PLAN_NO_ABSORB
obs = list(range(n))


def run(tag):
    top = N.Plan(ops, n, slots, flags=flags)
    plan = top.executed("expval")
    d = plan.describe()
    ws = torch.empty(top.workspace_bytes(B, "expval", n), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        out = top.run(ang, "expval", obs, workspace=ws)
    torch.cuda.synchronize()
    reps = 4
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = top.run(ang, "expval", obs, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps / B * 1e3
    shapes = " | ".join(f"T{s['T']}L{s['L']}{s['bits'][s['L']:] if s['L'] < s['T'] else ''}g{len(s.get('fast_groups') or s.get('groups') or [])}" for s in d["stages"])
    print(f"{tag:>8}: cand {d['candidate']:2d} model {d['model_cost'] * 2.0 ** (n - 24):9.1f} measured {us:9.1f} us/state  {shapes}", flush=True)
    del ws
    return d["candidate"], us


os.environ.pop("QMLE_FORCE_CAND", None)
chosen, us0 = run("model")
seen = {}
for k in range(48):
    os.environ["QMLE_FORCE_CAND"] = str(k)
    c, us = run(f"force {k}")
print(f"n={n} layers={layers} B={B}: model's choice {chosen} at {us0:.1f} us/state")
