#!/usr/bin/env python3
"""Per-pass HIP-event timing of DENSE plans (QMLE_PLAN_NO_SPARSE: every amplitude read /
computed / stored): K2 with and without observable folding, and HE circuits of 1-4 layers.

    python tools/dense_profile.py [B]
"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import _native as N
from tests.test_abi_cpu import he_layer_ops


def he_ops(n, layers):
    ops, slots = [], 0
    for _ in range(layers):
        o, s = he_layer_ops(n)
        ops += [(g, w, [x + slots for x in sl], m) for g, w, sl, m in o]
        slots += s
    return ops, slots


def run(n, B, layers, flags, label, reps=3, absorb=True):
    ops, slots = he_ops(n, layers)
    rng = np.random.default_rng(1000)
    ang = torch.from_numpy(rng.uniform(0, 2 * np.pi, (B, slots)).astype(np.float32)).cuda()
    top = N.Plan(ops, n, slots, flags=flags)
    plan = top.executed("expval")
    d = plan.describe()
    ws = torch.empty(top.workspace_bytes(B, "expval", n), dtype=torch.uint8, device="cuda")
    obs = list(range(n))
    top.run(ang, "expval", obs, workspace=ws)
    torch.cuda.synchronize()
    plan.profile_begin(len(d["stages"]) * B * reps + 8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        top.run(ang, "expval", obs, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    ms, cnt, _ = plan.profile_end()
    tot = e0.elapsed_time(e1) / reps / B
    passes = [(s["kind"], s["T"], s["L"], s["n_lowered"], s.get("lds_round_trips"),
               round(m / max(c, 1) * 1e3 / min(B, 32 if False else B) , 1))
              for s, m, c in zip(d["stages"], ms, cnt)]
    print(f"{label}: n={n} B={B} layers={layers} total {tot*1e3:.1f} us/state; "
          f"(kind,T,L,gates,groups,us/state-ish per launch/B): {passes} launches {cnt}", flush=True)
    for s in d["stages"]:
        print("    groups:", [(g["n_ops"], g["bits"]) for g in s["groups"]], "fast:",
              [(g["n_ops"], g["relayout"]) for g in s["fast_groups"]] if s["fast"] else None, flush=True)


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    F = N.plan_flags
    NS = N.PLAN_NO_SPARSE
    run(24, B, 1, NS, "K2 dense (folded CX)")
    run(24, B, 1, NS | N.PLAN_NO_ABSORB, "K2 dense, all 96 gates applied")
    for T, L in ((12, 4), (12, 5), (13, 4), (13, 5), (13, 6)):
        run(24, B, 1, NS | N.PLAN_NO_ABSORB | F(tile_bits=T, low_bits=L), f"  T{T} L{L}")
    for layers in (2, 4):
        run(24, B, layers, NS | N.PLAN_NO_ABSORB, f"HE {layers} layers dense")
