#!/usr/bin/env python3
"""BASELINE config 2 -- Model(20, 4, Hardware_Efficient), one sample, expval: wall per call with host and with
device arguments (run under rocprofv3 --kernel-trace and feed the database to tools/rocpd_timeline.py for the
launch sequence)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd.model import Model

m = Model(20, 4, "Hardware_Efficient")
rng = np.random.default_rng(1000)
p = rng.uniform(0, 6.28, m.params.shape[1:]).astype(np.float32)
x = np.array([0.5], dtype=np.float32)
pd, xd = torch.from_numpy(p).cuda(), torch.from_numpy(x.reshape(1, 1)).cuda()
for name, fn in (("host arrays", lambda: m(params=p, inputs=x)), ("device tensors", lambda: m(params=pd, inputs=xd).cpu())):
    for _ in range(20):
        fn()
    ts = []
    for _ in range(100):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"C2 {name}: median {ts[50] * 1e3:.3f} ms (min {ts[0] * 1e3:.3f})", flush=True)
