#!/usr/bin/env python3
"""K1 controlled gates: does the ORDER in which workgroups visit the control = 1 half matter?
CX (control = target + 1) at n = 28 on the target wires given (default: 14 16 17 18 19 20), HIP-event
time per launch for QMLE_K1_BLOCK_MUL = 0 (ascending) and a few odd multipliers (workgroup i takes block
i * mul mod grid: the workgroups in flight are spread over the whole state)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import _native as N

n = int(os.environ.get("K1_N", "28"))
wires = [int(x) for x in sys.argv[1:]] or [14, 16, 17, 18, 19, 20]
st = torch.randn((1, 1 << n, 2), device="cuda", dtype=torch.float32)
st = torch.view_as_complex(st / st.norm()).contiguous()
ang = torch.zeros((1, 1), device="cuda")
muls = [m if m == "lib" else int(m) for m in os.environ.get("K1_MULS", "0,3,17,257,4097,65537,9973,40503").split(",")]  # "lib": the library's own rule
print("target wire:", " ".join(f"{w:7d}" for w in wires))
for mul in muls:
    if mul == "lib": os.environ.pop("QMLE_K1_BLOCK_MUL", None)
    else: os.environ["QMLE_K1_BLOCK_MUL"] = str(mul)  # (0 = ascending order, also where the library would pick 4097)
    row = []
    for w in wires:
        plan = N.Plan([(os.environ.get("K1_GATE", "CX"), [(w + int(os.environ.get("K1_CTRL_DELTA", "1"))) % n, w], [] if os.environ.get("K1_GATE", "CX") in ("CX", "CY", "CZ") else [0], -1)], n, 1, flags=N.PLAN_NO_FUSION)
        ws = torch.empty(plan.workspace_bytes(1, "state"), dtype=torch.uint8, device="cuda")
        for _ in range(8):
            N.apply_inplace(plan, ang, st, ws)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(24):
            N.apply_inplace(plan, ang, st, ws)
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 24)
    print(f"mul {str(mul):>6s}:", " ".join(f"{t:7.4f}" for t in row), " ms per launch (8 D bytes = %.3f GB)" % (8 * (1 << n) / 1e9), flush=True)
