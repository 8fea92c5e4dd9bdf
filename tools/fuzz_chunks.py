#!/usr/bin/env python3
"""Differential fuzz of the chunk pipeline (round 5): random register sizes / layer counts / plan flags / batch and
chunk sizes / measurements; the two-stream run (default) must equal the one-stream run (QMLE_NO_CHUNK_OVERLAP=1) -- to
2e-7: measurements whose final sums add in arrival order differ in the last bit --, with unrelated work queued on the caller's stream around the calls."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_essentials_amd import _native as N
from tests.test_abi_cpu import he_layer_ops

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "3")))
bad = 0
trials = int(os.environ.get("FUZZ_N", "60"))
for t in range(trials):
    n = int(rng.integers(15, 23))
    layers = int(rng.integers(1, 4))
    flags = int(rng.choice([0, 128, 128 | 32, 32]))
    ops, slots = [], 0
    for _ in range(layers):
        o, s_ = he_layer_ops(n)
        ops += [(g, w, [x + slots for x in sl], m) for g, w, sl, m in o]
        slots += s_
    if rng.random() < 0.5:
        ops += [("CRX", [0, n - 1], [0], -1), ("RZ", [n // 2], [1], -1), ("H", [1], [], -1)]
    B = int(rng.integers(5, 40))
    sif = int(rng.integers(1, max(2, B // 2)))
    ang = torch.from_numpy(rng.uniform(0, 2 * np.pi, (B, slots)).astype(np.float32)).cuda()
    plan = N.Plan(ops, n, slots, flags=flags)
    meas = str(rng.choice(["expval", "state", "probs", "mw"]))
    obs = list(range(n)) if meas == "expval" else ()
    junk = torch.zeros(1 << 18, device="cuda")
    junk += 1
    a = plan.run(ang, meas, obs, states_in_flight=sif).clone()
    junk *= 3
    os.environ["QMLE_NO_CHUNK_OVERLAP"] = "1"
    b = plan.run(ang, meas, obs, states_in_flight=sif).clone()
    del os.environ["QMLE_NO_CHUNK_OVERLAP"]
    c = plan.run(ang, meas, obs).clone()
    ok = float((a - b).abs().max()) < 2e-7 and float((a - c).abs().max()) < 1e-6 and float(junk[0]) == 3.0
    if not ok:
        bad += 1
        print(f"MISMATCH trial {t}: n={n} layers={layers} flags={flags} B={B} chunk={sif} {meas}: "
              f"|a-b| {float((a - b).abs().max()):.3e} |a-c| {float((a - c).abs().max()):.3e}", flush=True)
print(f"{trials} trials, mismatches: {bad}")
