#!/usr/bin/env python3
"""K1 counter target: CX (control = target + 1) on the target wires given on the command line, n = 28,
a few launches each -- run under `rocprofv3 --pmc ...` (per-channel TCC counters)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import _native as N

n = int(os.environ.get("K1_N", "28"))
st = torch.randn((1, 1 << n, 2), device="cuda", dtype=torch.float32)
st = torch.view_as_complex(st / st.norm()).contiguous()
ang = torch.zeros((1, 1), device="cuda")
for w in (int(x) for x in sys.argv[1:]):
    plan = N.Plan([("CX", [(w + 1) % n, w], [], -1)], n, 1, flags=N.PLAN_NO_FUSION)
    ws = torch.empty(plan.workspace_bytes(1, "state"), dtype=torch.uint8, device="cuda")
    for _ in range(4):
        N.apply_inplace(plan, ang, st, ws)
    torch.cuda.synchronize()
    print("wire", w, "stage", plan.describe()["stages"][0]["kind"], flush=True)
