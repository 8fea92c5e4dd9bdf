#!/usr/bin/env python3
"""Per-call latency of small-batch model evaluation (launch-/host-bound regime)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd.model import Model

def bench(n, L, B, reps=50):
    model = Model(n, L, "Hardware_Efficient")
    rng = np.random.default_rng(0)
    params = rng.uniform(0, 6.28, (B, *model.params.shape[1:])).astype(np.float32)
    x = np.array([0.5], dtype=np.float32)
    # host path
    model(params=params, inputs=x); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        model(params=params, inputs=x)
    torch.cuda.synchronize(); host = (time.perf_counter() - t) / reps
    pd, xd = torch.from_numpy(params).cuda(), torch.from_numpy(x).cuda()
    model(params=pd, inputs=xd); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = model(params=pd, inputs=xd)
    torch.cuda.synchronize(); dev = (time.perf_counter() - t) / reps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = model(params=pd, inputs=xd)
    e1.record(); torch.cuda.synchronize()
    print(f"n={n} L={L} B={B}: host-array call {host*1e3:.3f} ms, device-tensor call {dev*1e3:.3f} ms "
          f"(GPU busy {e0.elapsed_time(e1)/reps:.3f} ms)", flush=True)

if __name__ == "__main__":
    for n, L, B in ((4, 2, 1), (10, 6, 1), (16, 4, 1), (20, 4, 1), (20, 4, 16), (24, 1, 1)):
        bench(n, L, B)
