#!/usr/bin/env python3
"""Kernel microbenchmarks (K1/K2 of SURVEY.md 8-d) -- prints one line per case."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import _native as N  # noqa: E402
from tests.test_abi_cpu import he_layer_ops  # noqa: E402


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2], ts[0]


def k1(n, which, flags_extra=0):
    D = 1 << n
    st = torch.randn((1, D, 2), device="cuda", dtype=torch.float32)
    st = torch.view_as_complex(st / st.norm()).contiguous()
    ang = torch.full((1, 1), 1.234, device="cuda")
    rows = []
    for gate, nbytes in which:
        for w in range(n):
            if gate in ("RX", "RZ", "H"):
                ops = [(gate, [w], [0] if gate != "H" else [], -1)]
            else:
                ops = [(gate, [w, (w + 1) % n], [0] if gate.startswith("CR") else [], -1)]
            plan = N.Plan(ops, n, 1, flags=N.plan_flags(no_fusion=True) | flags_extra)
            ws = torch.empty(plan.workspace_bytes(1, "state"), dtype=torch.uint8, device="cuda")
            med, best = timeit(lambda: N.apply_inplace(plan, ang, st, ws))
            rows.append((gate, w, med, nbytes * D / med / 1e6))
            print(f"K1 n={n} {gate:4s} wire={w:2d} bit={n-1-w:2d} {med:8.3f} ms  "
                  f"{nbytes * D / med / 1e6:8.1f} GB/s (algorithmic)", flush=True)
    return rows


def k2(n, B, fused=True, meas="expval"):
    ops, slots = he_layer_ops(n)
    rng = np.random.default_rng(1000)
    ang = torch.from_numpy(rng.uniform(0, 2 * np.pi, (B, slots)).astype(np.float32)).cuda()
    plan = N.Plan(ops, n, slots, flags=0 if fused else N.plan_flags(no_fusion=True))
    obs = list(range(n))
    ws = torch.empty(plan.workspace_bytes(B, meas, n), dtype=torch.uint8, device="cuda")
    out = None
    med, best = timeit(lambda: plan.run(ang, meas, obs, workspace=ws), reps=3, warm=1)
    st = plan.stats()
    gb = st["algo_bytes_per_state"] * B / med / 1e6
    print(f"K2 n={n} B={B} fused={fused} passes={st['n_passes']} {med:9.3f} ms  "
          f"{B / med * 1e3:9.1f} states/s  {len(ops) * B / med * 1e3:11.0f} gate-applies/s  "
          f"{gb:9.1f} GB/s (algorithmic, unfused bytes)", flush=True)


def lds(n, B, layers=3):
    ops = []
    slots = 0
    for _ in range(layers):
        o, s = he_layer_ops(n)
        ops += [(g, w, [x + slots for x in sl], m) for g, w, sl, m in o]
        slots += s
    rng = np.random.default_rng(1000)
    ang = torch.from_numpy(rng.uniform(0, 2 * np.pi, (B, slots)).astype(np.float32)).cuda()
    for nf in (False, True):
        plan = N.Plan(ops, n, slots, flags=N.plan_flags(no_fusion=nf))
        ws = torch.empty(plan.workspace_bytes(B, "state"), dtype=torch.uint8, device="cuda")
        out = torch.empty((B, 1 << n), dtype=torch.complex64, device="cuda")
        med, best = timeit(lambda: plan.run(ang, "state", out=out, workspace=ws))
        print(f"LDS n={n} B={B} ops={len(ops)} lowered={plan.stats()['n_lowered']} "
              f"merge={'off' if nf else 'on'} {med:8.3f} ms  {B / med * 1e3:10.0f} states/s  "
              f"{len(ops) * B / med * 1e3:12.0f} gate-applies/s", flush=True)


def mw(n, B=1):
    D = 1 << n
    st = torch.randn((B, D, 2), device="cuda", dtype=torch.float32)
    st = torch.view_as_complex(st / st.norm() * (B ** 0.5)).contiguous()
    med, best = timeit(lambda: N.meyer_wallach(st))
    print(f"MW n={n} B={B} {med:8.3f} ms  {8.0 * D * B / med / 1e6:8.1f} GB/s (algorithmic 8*D)", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="k1,k2,lds")
    ap.add_argument("--n", type=int, default=28)
    ap.add_argument("--k2n", type=int, default=24)
    ap.add_argument("--k2b", type=int, default=8)
    a = ap.parse_args()
    what = a.what.split(",")
    if "k1" in what:
        k1(a.n, [("RX", 16), ("RZ", 16), ("CX", 8), ("CRX", 8), ("CRZ", 8)])
    if "k1tile" in what:
        k1(a.n, [("RX", 16), ("CX", 8)], flags_extra=N.PLAN_FORCE_TILE)
    if "k2" in what:
        k2(a.k2n, a.k2b, fused=True)
        k2(a.k2n, min(a.k2b, 2), fused=False)
        k2(20, 64, fused=True)
    if "mw" in what:
        mw(28); mw(24, 8); mw(20, 64); mw(16, 256)
    if "lds" in what:
        lds(12, 2048)
        lds(10, 4096, layers=7)
        lds(14, 512)
