#!/usr/bin/env python3
"""Meyer-Wallach on one 2^n statevector (BASELINE config 5): HIP-event wall time per call and
the implied HBM rate against the 8 D-byte single-read roofline of SURVEY 8-d.
    python tools/mw_bench.py [--random] [n ...]      (--random: Gaussian amplitudes instead of an HE layer)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import _native as N
from tests.test_abi_cpu import he_layer_ops

args = [x for x in sys.argv[1:] if not x.startswith("--")]
rand = "--random" in sys.argv
for n in (int(x) for x in (args or ["28", "24"])):
    if rand:
        st = torch.randn((1, 1 << n, 2), device="cuda", dtype=torch.float32)
        st = torch.view_as_complex(st / st.norm()).contiguous()
    else:
        ops, slots = he_layer_ops(n)
        ang = torch.from_numpy(np.random.default_rng(6).uniform(0, 6.28, (1, slots)).astype(np.float32)).cuda()
        st = N.Plan(ops, n, slots).run(ang, "state")
    # the chip's clock settles only after ~20 ms of sustained work (the first calls after an idle
    # period run at boost clocks, the next ones over-throttled: 0.98 / 1.14 / 1.01 ms per call
    # over the first 1 / 10 / 200 calls at n = 28): warm up, then time a long run
    reps = int(os.environ.get("MW_REPS", "100"))
    for _ in range(max(1, reps // 4)):
        q = N.meyer_wallach(st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        q = N.meyer_wallach(st)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    D8 = 8.0 * (1 << n)
    print(f"n={n}{' random' if rand else ''}: Q={float(q[0]):.6f}  {ms:.4f} ms per call ({N.mw_reads(n)} reads)  = {D8/ms/1e9:.3f} TB/s of the "
          f"8D-byte algorithm = {D8/ms/1e9/8.0:.3f} of 8 TB/s", flush=True)
    del st
