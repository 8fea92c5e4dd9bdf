#!/usr/bin/env python3
"""Meyer-Wallach on one 2^n statevector (BASELINE config 5): HIP-event wall time per call and
the implied HBM rate against the 8 D-byte single-read roofline of SURVEY 8-d."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import _native as N
from tests.test_abi_cpu import he_layer_ops

for n in (int(x) for x in (sys.argv[1:] or ["28", "24"])):
    ops, slots = he_layer_ops(n)
    ang = torch.from_numpy(np.random.default_rng(6).uniform(0, 6.28, (1, slots)).astype(np.float32)).cuda()
    st = N.Plan(ops, n, slots).run(ang, "state")
    q = N.meyer_wallach(st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        q = N.meyer_wallach(st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    D8 = 8.0 * (1 << n)
    print(f"n={n}: Q={float(q[0]):.6f}  {ms:.4f} ms per call  = {D8/ms/1e9:.3f} TB/s of the 8D-byte "
          f"algorithm = {D8/ms/1e9/8.0:.3f} of 8 TB/s", flush=True)
    del st
