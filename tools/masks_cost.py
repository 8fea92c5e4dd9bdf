#!/usr/bin/env python3
"""Cost of the general-mask <Z..Z> epilogue of k_tile2: default-engine HE circuits (folded CX tail)
with and without the measuring epilogue (QMLE_DBG_T2=2, results wrong by design)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from qml_essentials_amd import _native as N
from dense_profile import he_ops
for n, layers in ((24, 4), (24, 3), (24, 2)):
    ops, slots = he_ops(n, layers)
    B = 64
    ang = torch.from_numpy(np.random.default_rng(1).uniform(0, 6.28, (B, slots)).astype(np.float32)).cuda()
    top = N.Plan(ops, n, slots)
    ws = torch.empty(top.workspace_bytes(B, "expval", n), dtype=torch.uint8, device="cuda")
    obs = list(range(n))
    top.run(ang, "expval", obs, workspace=ws); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): top.run(ang, "expval", obs, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    d = top.describe()
    print(f"dbg={os.environ.get('QMLE_DBG_T2','0')} n={n} layers={layers} {'folded' if 'expval_plan' in d else 'applied'}: "
          f"{e0.elapsed_time(e1)/3/B*1e3:.1f} us/state", flush=True)
