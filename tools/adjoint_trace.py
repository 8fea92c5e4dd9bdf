#!/usr/bin/env python3
"""Kernel launches of one fused adjoint gradient of BASELINE config 2's model (20 qubits, 4 layers, 300
parameters, one sample): run under `rocprofv3 --kernel-trace --stats` to see whether the 1 ms per
gradient is kernels or launch gaps."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd.model import Model

n = int(os.environ.get("ADJ_N", "20"))
m = Model(n, 4, "Hardware_Efficient")
p = torch.tensor(np.asarray(m.params[0]), dtype=torch.float32, device="cuda")
x = torch.tensor([[0.5]], dtype=torch.float32, device="cuda")
cot = torch.ones((1,), dtype=torch.float32, device="cuda")
for _ in range(3):
    m.vjp_device(p, x, cot, force_mean=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for _ in range(20):
    g, _ = m.vjp_device(p, x, cot, force_mean=True)
e1.record()
torch.cuda.synchronize()
print(f"n={n}: wall {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per gradient, GPU events {e0.elapsed_time(e1) / 20:.3f} ms", flush=True)
