#!/usr/bin/env python3
"""Cost anatomy of one k_tile pass: groups x ops per group (n = 24, 32 states/launch)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import _native as N

n, B = 24, 32
flags = N.plan_flags(force_tile=True, force_global=True, tile_bits=int(os.environ.get("T", 12)),
                     low_bits=int(os.environ.get("L", 4)))
st = torch.randn((B, 1 << n, 2), device="cuda", dtype=torch.float32)
st = torch.view_as_complex(st / 4096.0).contiguous()


def rx(ws, g="RX"):
    return [(g, [w]) for w in ws]


def cx(pairs):
    return [("CX", list(p)) for p in pairs]


CASES = {
    "E1 1grp x 1 dense": rx([23]),
    "E2 1grp x 4 dense": rx(range(20, 24)),
    "E3 1grp x 8 dense + 2 CX": rx(range(20, 24)) + cx([(20, 21), (22, 23)]) + rx(range(20, 24), "RY"),
    "E4 1grp x 12 dense + 3 CX": rx(range(20, 24)) + cx([(20, 21), (22, 23)]) + rx(range(20, 24), "RY")
                                  + cx([(21, 22)]) + rx(range(20, 24)),
    "E5 2grp x 4 dense": rx(range(16, 24)),
    "E6 3grp x 4 dense": rx(range(12, 24)),
    "E7 1grp x 6 CX only": cx([(20, 21), (22, 23), (21, 22), (20, 21), (22, 23), (21, 22)]),
    "E8 1grp x 4 diag (RZ)": rx(range(20, 24), "RZ"),
}
for name, gates in CASES.items():
    ops, slot = [], 0
    for g, w in gates:
        if g == "CX":
            ops.append((g, w, [], -1))
        else:
            ops.append((g, w, [slot], -1)); slot += 1
    ang = torch.rand((B, max(1, slot)), device="cuda") * 6.28
    plan = N.Plan(ops, n, max(1, slot), flags=flags)
    d = plan.describe()
    ws = torch.empty(plan.workspace_bytes(B, "state"), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        N.apply_inplace(plan, ang, st, ws)
    plan.profile_begin(64)
    for _ in range(5):
        N.apply_inplace(plan, ang, st, ws)
    ms, cnt, _ = plan.profile_end()
    per = sum(ms) / 5 / B * 1e3
    print(f"{name:28s} passes={len(d['stages'])} trips={[s.get('lds_round_trips') for s in d['stages']]}"
          f"  {per:7.1f} us/state", flush=True)
