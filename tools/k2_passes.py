#!/usr/bin/env python3
"""Per-pass HIP-event times of the all-live K2 plan (bench.py's headline) -- for A/B runs under the
plan compiler's tuning switches (QMLE_FORCE_CAND, QMLE_PAD_BIT, ...)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import _native as N
from tests.test_abi_cpu import he_layer_ops

n = int(os.environ.get("K2_N", "24"))
B = int(os.environ.get("K2_B", str(max(2, 128 >> max(0, n - 24)) if n >= 24 else 128 << min(4, 24 - n))))
layers = int(os.environ.get("K2_LAYERS", "1"))
ops, slots = [], 0
for _ in range(layers):
    o, s_ = he_layer_ops(n)
    ops += [(g, w, [x + slots for x in sl], m) for g, w, sl, m in o]
    slots += s_
ang = torch.from_numpy(np.random.default_rng(1000).uniform(0, 2 * np.pi, (B, slots)).astype(np.float32)).cuda()
flags = N.PLAN_NO_SPARSE | N.PLAN_NO_ABSORB
top = N.Plan(ops, n, slots, flags=flags)
plan = top.executed("expval")
d = plan.describe()
ws = torch.empty(top.workspace_bytes(B, "expval", n), dtype=torch.uint8, device="cuda")
obs = list(range(n))
for _ in range(3):
    out = top.run(ang, "expval", obs, workspace=ws)
torch.cuda.synchronize()
reps = 5
plan.profile_begin(len(d["stages"]) * reps * 8 + 16)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    out = top.run(ang, "expval", obs, workspace=ws)
e1.record(); torch.cuda.synchronize()
ms, cnt, _ = plan.profile_end()
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("QMLE_"))
print(f"[{tag}] {e0.elapsed_time(e1) / reps / B * 1e3:.1f} us/state; checksum {float(out.double().sum()):.6f}")
for s, m in zip(d["stages"], ms):
    print(f"    T={s['T']} L={s['L']} bits={s['bits']} groups={len(s.get('fast_groups') or s.get('groups') or [])}: {m / reps / B * 1e3:.1f} us", flush=True)
