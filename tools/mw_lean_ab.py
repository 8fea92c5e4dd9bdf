#!/usr/bin/env python3
"""Meyer-Wallach at n = 28 after the circuit: resident route (three reads), fused tiled route with the round-4
split (QMLE_MW_NO_LEAN=1) and with the lean split (the producing pass leaves positions 0..3 to the first later
read).  HIP events over back-to-back calls, circuit alone subtracted."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from qml_essentials_amd import _native as N

n = int(os.environ.get("MW_N", "28"))
ops, slots = bench._he_layer_ops(n)
ang = torch.from_numpy(np.random.default_rng(6).uniform(0, 6.28, (1, slots)).astype(np.float32)).cuda()
plan = N.Plan(ops, n, slots)
st = plan.run(ang, "state")
ws_s = torch.empty(plan.workspace_bytes(1, "state"), dtype=torch.uint8, device="cuda")
ws_m = torch.empty(plan.workspace_bytes(1, "mw"), dtype=torch.uint8, device="cuda")


def timed(fn, reps=100, warm=25):
    for _ in range(warm):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


for rnd in range(2):
    circ, _ = timed(lambda: plan.run(ang, "state", out=st, workspace=ws_s))
    res, q0 = timed(lambda: N.meyer_wallach(st))
    os.environ["QMLE_MW_FUSE_TILED"] = "0"
    dflt, qd = timed(lambda: plan.run(ang, "mw", workspace=ws_m))
    del os.environ["QMLE_MW_FUSE_TILED"]
    os.environ["QMLE_MW_NO_LEAN"] = "1"
    full, qf = timed(lambda: plan.run(ang, "mw", workspace=ws_m))
    del os.environ["QMLE_MW_NO_LEAN"]
    lean, ql = timed(lambda: plan.run(ang, "mw", workspace=ws_m))
    print(f"n={n} round {rnd}: circuit {circ:.4f} ms | resident (3 reads) {res:.4f} ms | after the circuit: stand-alone reads {dflt - circ:.4f}, "
          f"fused full epilogue {full - circ:.4f}, fused lean epilogue {lean - circ:.4f} ms | "
          f"Q {float(q0[0]):.7f} {float(qd[0, 0]):.7f} {float(qf[0, 0]):.7f} {float(ql[0, 0]):.7f}", flush=True)
